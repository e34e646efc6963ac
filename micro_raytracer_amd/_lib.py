"""Loader of libmrt_hip.so (the C ABI of include/mrt.h).  Fails loudly: there is no fallback path."""
import ctypes as C
import os
import subprocess

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRT_LIB") or os.path.join(_PKG, "libmrt_hip.so")   # MRT_LIB: experiment builds only
_LIB = None

# every symbol include/mrt.h declares
SYMBOLS = (
    "mrt_create", "mrt_destroy", "mrt_execute", "mrt_dims", "mrt_accum", "mrt_accum_local", "mrt_accum_device_ptr",
    "mrt_set_accum", "mrt_img", "mrt_img_ss", "mrt_reset", "mrt_get_stats", "mrt_last_error", "mrt_last_status",
    "mrt_abi_version", "mrt_device_count", "mrt_selftest_math", "mrt_padded_rows", "mrt_bind_accum",
    "mrt_set_accum_device", "mrt_save_image", "mrt_selftest_sweep", "mrt_plan_launch",
)


class MrtError(RuntimeError):
    """Err(String) of the reference's Result<_, String> (src/sampler.rs:80, src/cli.rs:155) + the ABI status code."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


def build(force=False):
    """Compile libmrt_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    args = ["make", "-C", csrc]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(micro_raytracer_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u32, u32p, f32p, u8p = C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    L.mrt_create.restype = vp
    L.mrt_create.argtypes = [vp, C.POINTER(_abi.Opts)]
    L.mrt_destroy.restype = None
    L.mrt_destroy.argtypes = [vp]
    L.mrt_execute.argtypes = [vp, u32, C.POINTER(C.c_double)]
    L.mrt_dims.argtypes = [vp, u32p, u32p, u32p]
    L.mrt_accum.argtypes = [vp, f32p, u32p]
    L.mrt_accum_local.argtypes = [vp, f32p, u32p]
    L.mrt_accum_device_ptr.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.mrt_set_accum.argtypes = [vp, f32p, u32]
    L.mrt_img.argtypes = [vp, u8p]
    L.mrt_img_ss.argtypes = [vp, u8p]
    L.mrt_reset.argtypes = [vp]
    L.mrt_get_stats.argtypes = [vp, C.POINTER(_abi.Stats)]
    L.mrt_last_error.restype = C.c_char_p
    L.mrt_last_status.restype = C.c_int
    L.mrt_abi_version.restype = u32
    L.mrt_device_count.restype = C.c_int
    L.mrt_padded_rows.argtypes = [vp, u32p]
    L.mrt_bind_accum.argtypes = [vp, vp, C.c_size_t]
    L.mrt_set_accum_device.argtypes = [vp, vp, u32]
    L.mrt_save_image.argtypes = [C.c_char_p, u8p, u32, u32]
    L.mrt_plan_launch.argtypes = [vp, C.POINTER(_abi.Plan)]
    L.mrt_selftest_math.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, C.c_size_t]
    L.mrt_selftest_sweep.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, u32, C.POINTER(C.c_uint64), f32p]
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        L = lib()
        raise MrtError(rc, L.mrt_last_error().decode())


def plan_launch(render_or_holder):
    """What mrt_create would stage in LDS for this scene and the launch shape it would take (host only, no device needed)."""
    h = render_or_holder if hasattr(render_or_holder, "ptr") else _abi.build_desc(render_or_holder)
    pl = _abi.Plan()
    check(lib().mrt_plan_launch(C.cast(h.ptr(), C.c_void_p), C.byref(pl)))
    d = {k: getattr(pl, k) for k, _ in pl._fields_ if k != "reserved"}
    d["staging"] = _abi.STAGING[pl.staging]
    return d


def selftest_math(op, a, b=None, device=0):
    import numpy as np
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        bp = b.ctypes.data_as(C.POINTER(C.c_float))
    check(lib().mrt_selftest_math(device, op, a.ctypes.data_as(C.POINTER(C.c_float)), bp,
                                  out.ctypes.data_as(C.POINTER(C.c_float)), a.size))
    return out


def selftest_sweep(op, first, count, seed=1, device=0):
    """(mismatches, example[4]) of mrt_selftest_sweep: fast IEEE cores vs the compiler's expansions, on the device."""
    import numpy as np
    mis = C.c_uint64()
    ex = np.zeros(4, np.float32)
    check(lib().mrt_selftest_sweep(device, op, first, count, seed, C.byref(mis), ex.ctypes.data_as(C.POINTER(C.c_float))))
    return mis.value, ex


def save_image(path, rgb8):
    """img.save(filename) (src/cli.rs:168,174) for .ppm / .png."""
    import numpy as np
    a = np.ascontiguousarray(rgb8, np.uint8)
    check(lib().mrt_save_image(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0]))
