"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (micro_raytracer_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mrt_oracle.c", "mrt_oracle.h", "oracle_math.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "mrt.h"))
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        so = build()
    L = C.CDLL(so)
    vp, u32, u64, f32p, u8p, u32p = C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    L.orc_create.restype = vp
    L.orc_create.argtypes = [vp, u64]
    L.orc_destroy.argtypes = [vp]
    L.orc_error.restype = C.c_char_p
    L.orc_dims.argtypes = [vp, u32p, u32p]
    L.orc_execute.restype = C.c_double
    L.orc_execute.argtypes = [vp, u32, u32, u32]
    L.orc_execute_rows.restype = C.c_double
    L.orc_execute_rows.argtypes = [vp, u32, u32, u32, u32, u32]
    L.orc_accum.argtypes = [vp, f32p, u32p]
    L.orc_set_accum.argtypes = [vp, f32p, u32]
    L.orc_reset.argtypes = [vp]
    L.orc_segments.restype = u64
    L.orc_segments.argtypes = [vp]
    L.orc_img.argtypes = [vp, u8p]
    L.orc_img_ss.argtypes = [vp, u8p]
    L.orc_trace_pixel.argtypes = [vp, u32, u32, u32, f32p, u32p]
    L.orc_tonemap_px.argtypes = [f32p, u32, C.c_float, C.c_float, u8p]
    L.orc_lanczos3_resize.argtypes = [u8p, u32, u32, u8p, u32, u32]
    L.orc_lanczos3_weights.argtypes = [u32, u32, u32, u32p, f32p, u32]
    L.orc_mesh_octree.argtypes = [vp, u32, f32p, u32p, u32p, u32, u32p]
    L.orc_path_key.restype = u32
    L.orc_path_key.argtypes = [u64, u32, u32]
    L.orc_draw_u32.restype = u32
    L.orc_draw_u32.argtypes = [u32, u32]
    L.orc_draw_f32.restype = C.c_float
    L.orc_draw_f32.argtypes = [u32, u32]
    L.orc_math.argtypes = [C.c_int, f32p, f32p, f32p, C.c_size_t]
    _LIB = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Oracle:
    """CPU restatement of Sampler (reference src/sampler.rs:11-100) on a flattened render description."""

    def __init__(self, desc_holder, seed=1):
        L = lib()
        self._h = desc_holder
        self._c = L.orc_create(C.cast(desc_holder.ptr(), C.c_void_p), C.c_uint64(seed))
        if not self._c:
            raise ValueError(L.orc_error().decode())
        nw, nh = C.c_uint32(), C.c_uint32()
        L.orc_dims(self._c, C.byref(nw), C.byref(nh))
        self.nw, self.nh = nw.value, nh.value
        self.res = (desc_holder.desc.frame.res_w, desc_holder.desc.frame.res_h)

    def close(self):
        if self._c:
            lib().orc_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def execute(self, n_samples=1, threads=None, n_dim=64, rows=None):
        threads = threads or os.cpu_count() or 1
        if rows is None:
            return lib().orc_execute(self._c, n_samples, threads, n_dim)
        return lib().orc_execute_rows(self._c, n_samples, threads, n_dim, rows[0], rows[1])

    def accum(self):
        out = np.empty((self.nh, self.nw, 3), np.float32)
        cnt = C.c_uint32()
        lib().orc_accum(self._c, _fp(out), C.byref(cnt))
        return out, cnt.value

    def set_accum(self, rgb, count):
        rgb = np.ascontiguousarray(rgb, np.float32)
        assert rgb.shape == (self.nh, self.nw, 3)
        lib().orc_set_accum(self._c, _fp(rgb), count)

    def reset(self):
        lib().orc_reset(self._c)

    @property
    def segments(self):
        return lib().orc_segments(self._c)

    def img(self):
        out = np.empty((self.res[1], self.res[0], 3), np.uint8)
        if lib().orc_img(self._c, _up(out)) != 0:
            raise RuntimeError("img before any sample")
        return out

    def img_ss(self):
        out = np.empty((self.nh, self.nw, 3), np.uint8)
        if lib().orc_img_ss(self._c, _up(out)) != 0:
            raise RuntimeError("img before any sample")
        return out

    def trace_pixel(self, x, y, s):
        rgb = np.zeros(3, np.float32)
        seg = C.c_uint32()
        lib().orc_trace_pixel(self._c, x, y, s, _fp(rgb), C.byref(seg))
        return rgb, seg.value

    def mesh_octree(self, renderer):
        L = lib()
        n_ids = C.c_uint32()
        nl = L.orc_mesh_octree(self._c, renderer, None, None, None, 0, C.byref(n_ids))
        if nl < 0:
            return None
        boxes = np.zeros((nl, 6), np.float32)
        counts = np.zeros(nl, np.uint32)
        ids = np.zeros(max(1, n_ids.value), np.uint32)
        L.orc_mesh_octree(self._c, renderer, _fp(boxes), counts.ctypes.data_as(C.POINTER(C.c_uint32)),
                          ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size, C.byref(n_ids))
        return boxes, counts, ids[:n_ids.value]


def math(op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        bp = _fp(b)
    lib().orc_math(op, _fp(a), bp, _fp(out), a.size)
    return out


def tonemap_px(sum3, count, gamma, exp):
    s = np.ascontiguousarray(sum3, np.float32)
    out = np.zeros(3, np.uint8)
    lib().orc_tonemap_px(_fp(s), count, C.c_float(gamma), C.c_float(exp), _up(out))
    return out


def lanczos3_resize(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape[:2]
    out = np.empty((dh, dw, 3), np.uint8)
    if lib().orc_lanczos3_resize(_up(src), sw, sh, _up(out), dw, dh) != 0:
        raise RuntimeError("resize failed")
    return out


def path_key(seed, pixel, sample):
    return lib().orc_path_key(seed, pixel, sample)


def draw_u32(pk, dim):
    return lib().orc_draw_u32(pk, dim)
