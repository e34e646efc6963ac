"""TEST INFRASTRUCTURE: ctypes binding of tests/emu/libemu.so (x86 build of the kernel headers)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        # (one builder at a time: pytest-xdist workers start together, and a second `make` would load a half-written library)
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", _HERE, "libemu.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(os.environ.get("MRT_EMU_LIB") or os.path.join(_HERE, "libemu.so"))      # MRT_EMU_LIB: experiment builds (tests/mesh_probe.py with scaled margins)
        L.emu_error.restype = C.c_char_p
        L.emu_pack.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
        L.emu_features.argtypes = [C.c_void_p]
        L.emu_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        L.emu_img.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.emu_math.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t]
        L.emu_octree.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def pack(holder):
    L = lib()
    bw, nw, nh = C.c_uint32(), C.c_uint32(), C.c_uint32()
    info = (C.c_uint32 * 8)()
    rc = L.emu_pack(C.cast(holder.ptr(), C.c_void_p), C.byref(bw), C.byref(nw), C.byref(nh), info)
    if rc:
        raise ValueError((rc, L.emu_error().decode()))
    keys = ("n_tex_u8", "n_tex_f32", "n_nodes", "n_leaf_ids", "n_tris", "n_xf", "n_inst", "n_rend")
    d = dict(zip(keys, list(info)))
    d.update(blob_words=bw.value, nw=nw.value, nh=nh.value)
    return d


LAYOUT_KEYS = ("rend", "cam", "lin", "bvh", "bvhinst", "inst", "instx", "xf", "mat", "light", "tex", "lut", "mesh", "node", "parent", "tbvh",
               "tri", "memb", "membe", "leaf", "lds_words", "blob_words", "n_tbvh_nodes", "n_nodes", "n_tris", "n_leaf_ids", "n_bvh_nodes", "features",
               "lds_words_hot")


def layout(holder):
    """Word offsets of the packed tables in blob order + sizes (csrc/mrt_scene.h)."""
    out = (C.c_uint32 * 32)()
    rc = lib().emu_layout(C.cast(holder.ptr(), C.c_void_p), out)
    if rc:
        raise ValueError((rc, lib().emu_error().decode()))
    return dict(zip(LAYOUT_KEYS, list(out)))


def render(holder, seed, n_samples, sample_base=0, rows=None, threads=8, accum=None, deep_nodes=0):
    L = lib()
    info = pack(holder)
    nw, nh = info["nw"], info["nh"]
    if accum is None:
        accum = np.zeros((nh, nw, 3), np.float32)
    seg = C.c_uint64()
    r0, r1 = rows if rows else (0, nh)
    rc = L.emu_render_deep(C.cast(holder.ptr(), C.c_void_p), C.c_uint64(seed), sample_base, n_samples, r0, r1, threads, _fp(accum), C.byref(seg), deep_nodes)
    if rc:
        raise ValueError((rc, L.emu_error().decode()))
    return accum, seg.value


def img(holder, accum, count):
    L = lib()
    d = holder.desc
    info = pack(holder)
    ss = np.empty((info["nh"], info["nw"], 3), np.uint8)
    out = np.empty((d.frame.res_h, d.frame.res_w, 3), np.uint8)
    accum = np.ascontiguousarray(accum, np.float32)
    rc = L.emu_img(C.cast(holder.ptr(), C.c_void_p), _fp(accum), count, ss.ctypes.data_as(C.POINTER(C.c_uint8)),
                   out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise ValueError((rc, L.emu_error().decode()))
    return ss, out


def math(op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        bp = _fp(b)
    lib().emu_math(op, _fp(a), bp, _fp(out), a.size)
    return out


def octree(tris):
    L = lib()
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    n_ids = C.c_uint32()
    nl = L.emu_octree(_fp(tris), tris.shape[0], None, None, None, 0, C.byref(n_ids))
    if nl < 0:
        return None
    boxes = np.zeros((nl, 6), np.float32)
    counts = np.zeros(nl, np.uint32)
    ids = np.zeros(max(1, n_ids.value), np.uint32)
    L.emu_octree(_fp(tris), tris.shape[0], _fp(boxes), counts.ctypes.data_as(C.POINTER(C.c_uint32)),
                 ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size, C.byref(n_ids))
    return boxes, counts, ids[:n_ids.value]
