"""Python mirror of the reference's `Sampler` (src/sampler.rs:11-100) over the C ABI.

Same three entry points, same meaning:
    Sampler(workers, n_dim)              Sampler::new      (src/sampler.rs:19)
    execute(scene, frame, rt) -> secs    Sampler::execute  (src/sampler.rs:28)   one sample pass
    img(frame) -> uint8 [h][w][3]        Sampler::img      (src/sampler.rs:80)
`workers` / `n_dim` (the thread-pool size and tile grid of the CPU implementation) have no
meaning on the GPU and are accepted for signature compatibility only.  The context is created
lazily on the first execute, because the scene and frame only arrive then, exactly as the
Rust shim in INTEGRATION.md does.
"""
from __future__ import annotations

import ctypes as C
import zlib

import numpy as np

from . import _abi, _lib
from .scene import Render


def _fingerprint(render: Render):
    """What the device context was built from, cheap enough to recompute on every execute (the reference's
    Sampler::execute receives scene, frame and rt on every call, src/sampler.rs:28): every scalar and small vector by
    value; bulk arrays (mesh triangles, texels, long instance lists) by shape, dtype and a CRC of a strided sample of
    at most ~4 K elements -- an in-place edit of a few elements of a large array between two sample elements is NOT
    seen (call Sampler.invalidate() after such an edit); anything that replaces, resizes or rewrites the array is.
    Bulk data that is not an ndarray (a Python list assigned after load_render) is converted and hashed whole on every
    call: correct, but slow -- keep bulk data as ndarrays, as load_render leaves it."""
    def arr(a):
        if a is None:
            return None
        if not isinstance(a, np.ndarray):
            a = np.asarray(a)
            return (a.shape, a.dtype.str, zlib.crc32(a.tobytes()))
        if a.size <= 64:
            return a.tobytes()
        flat = a.reshape(-1)
        step = max(1, flat.size // 4096)
        return (a.shape, a.dtype.str, zlib.crc32(np.ascontiguousarray(flat[::step]).tobytes()), zlib.crc32(flat[-16:].tobytes()))

    def tex(t):
        return None if t is None else (t.w, t.h, arr(t.dat))

    def insts(lst):
        if len(lst) <= 16:
            return tuple((arr(p), arr(d)) for p, d in lst)
        step = max(1, len(lst) // 64)
        return (len(lst), tuple((arr(p), arr(d)) for p, d in lst[::step]), arr(lst[-1][0]), arr(lst[-1][1]))

    cam = render.frame.cam
    out = [render.rt.bounce, render.rt.loss, tuple(render.frame.res), render.frame.ssaa,
           arr(cam.pos), arr(cam.dir), cam.fov, cam.gamma, cam.exp, cam.aprt, cam.foc,
           arr(render.scene.sky.color), render.scene.sky.pwr]
    for l in render.scene.light:
        out.append((l.kind, arr(l.v), l.pwr, arr(l.color)))
    for o in render.scene.renderer:
        m = o.mat
        out.append((o.kind, o.r, arr(o.n), arr(o.sizes), arr(o.vtx), arr(o.mesh), arr(m.albedo), m.rough, m.metal, m.glass,
                    m.opacity, m.emit, tuple(tex(getattr(m, k)) for k in _abi.MAP_SLOTS), insts(o.inst)))
    return tuple(out)


class Sampler:
    def __init__(self, workers: int = 24, n_dim: int = 64, *, seed: int = 1, device: int = -1,
                 shard_index: int = 0, shard_count: int = 1, shard_rows: int = 8, n_devices: int = 0, flags: int = 0):
        self.workers, self.n_dim = workers, n_dim
        self.seed, self.device = seed, device
        self.shard_index, self.shard_count, self.shard_rows = shard_index, shard_count, shard_rows
        self.n_devices = n_devices      # > 1: one process drives that many GPUs (RCCL gather inside mrt_execute)
        self.flags = flags              # _abi.FLAG_COUNT_SEGMENTS: keep the path-segment counter (stats()["segments"])
        self._ctx = None
        self._holder = None
        self._render = None             # strong reference: the context belongs to THIS description (id() of a dead
        self._print = None              # temporary is reused by CPython), and to its contents at creation time
        self._bound = None              # (device pointer, bytes) of a caller-owned accumulator (bind_accum), re-applied after a rebuild
        self.nw = self.nh = self.local_rows = 0
        self.res = (0, 0)

    # -- context -----------------------------------------------------------------------------
    def _ensure(self, render: Render):
        fp = _fingerprint(render)
        if self._ctx is not None and render is self._render and fp == self._print:
            return
        # Another description, or this one edited in place: the device context is rebuilt from what is passed NOW.
        # Like the reference's Sampler, what has been accumulated so far is kept when the supersampled frame keeps
        # its size (src/sampler.rs:60-70 adds into the same map whatever the scene); a different size starts afresh.
        # A context that owns only some rows (shard_count > 1) or several devices has no entry point that restores
        # its rows and their sample count, so rebuilding it with samples on board would silently desynchronise the
        # caller's view (the bound buffer, ShardedSampler.count): that is an error, not a quiet restart.
        carry = None
        bound = self._bound
        if self._ctx is not None:
            same_size = (self.nw, self.nh) == (render.frame.nw, render.frame.nh)
            whole = self.shard_count == 1 and self.n_devices <= 1
            if not whole and self._count() > 0:
                raise _lib.MrtError(_abi.MRT_ERR_STATE, "the render description changed under a sharded / multi-device context that "
                                    "already holds samples: create a new Sampler (or reset() first)")
            if whole and same_size:
                carry = self.accum()
            if not same_size:
                bound = None                 # the caller's buffer was sized for the old frame
        self.close()
        L = _lib.lib()
        self._holder = _abi.build_desc(render)
        opts = _abi.Opts()
        opts.abi_version = _abi.ABI_VERSION
        opts.seed = self.seed
        opts.device = self.device
        opts.shard_index, opts.shard_count, opts.shard_rows = self.shard_index, self.shard_count, self.shard_rows
        opts.n_devices = self.n_devices
        opts.flags = self.flags
        ctx = L.mrt_create(C.cast(self._holder.ptr(), C.c_void_p), C.byref(opts))
        if not ctx:
            raise _lib.MrtError(L.mrt_last_status(), L.mrt_last_error().decode())
        self._ctx, self._render, self._print = ctx, render, fp
        nw, nh, lr = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(L.mrt_dims(ctx, C.byref(nw), C.byref(nh), C.byref(lr)))
        self.nw, self.nh, self.local_rows = nw.value, nh.value, lr.value
        self.res = tuple(render.frame.res)
        if bound is not None:
            self.bind_accum(*bound)          # the new context renders into the caller's buffer again (zeroed by the bind)
        if carry is not None and carry[1] > 0:
            self.set_accum(*carry)

    def _count(self) -> int:
        cnt = C.c_uint32()
        _lib.check(_lib.lib().mrt_accum(self._ctx, None, C.byref(cnt)))
        return cnt.value

    def invalidate(self):
        """Force the next execute to rebuild the device context from the description it is given (after an in-place
        edit of bulk data the fingerprint cannot see)."""
        self._print = None

    def close(self):
        if self._ctx is not None:
            _lib.lib().mrt_destroy(self._ctx)
            self._ctx = None
        self._render = self._print = None
        self._bound = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the reference's API -----------------------------------------------------------------
    def execute(self, scene_or_render, frame=None, rt=None, n_samples: int = 1) -> float:
        """One Sampler::execute pass (n_samples > 1: that many consecutive passes in one launch).

        Accepts either a whole Render, or (scene, frame, rt) like the reference."""
        render = scene_or_render if isinstance(scene_or_render, Render) else Render(rt=rt, frame=frame, scene=scene_or_render)
        if not isinstance(scene_or_render, Render):
            # keep one Render object per (scene, frame, rt) triple so the context is reused across passes
            k = (id(scene_or_render), id(frame), id(rt))
            if getattr(self, "_triple_key", None) == k:
                render = self._triple_render
            else:
                self._triple_key, self._triple_render = k, render
        self._ensure(render)
        secs = C.c_double()
        _lib.check(_lib.lib().mrt_execute(self._ctx, n_samples, C.byref(secs)))
        return secs.value

    def img(self, frame=None) -> np.ndarray:
        self._need()
        out = np.empty((self.res[1], self.res[0], 3), np.uint8)
        _lib.check(_lib.lib().mrt_img(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    # -- extras of the C ABI -----------------------------------------------------------------
    def _need(self):
        if self._ctx is None:
            raise _lib.MrtError(_abi.MRT_ERR_STATE, "no context: call execute() first")

    def img_ss(self) -> np.ndarray:
        self._need()
        out = np.empty((self.nh, self.nw, 3), np.uint8)
        _lib.check(_lib.lib().mrt_img_ss(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def accum(self):
        """(colors, last_count): full-frame f32 sums [nh][nw][3] (rows of other shards are zero)."""
        self._need()
        out = np.zeros((self.nh, self.nw, 3), np.float32)
        cnt = C.c_uint32()
        _lib.check(_lib.lib().mrt_accum(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(cnt)))
        return out, cnt.value

    def accum_local(self):
        self._need()
        out = np.zeros((self.local_rows, self.nw, 3), np.float32)
        rows = np.zeros(self.local_rows, np.uint32)
        _lib.check(_lib.lib().mrt_accum_local(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)),
                                              rows.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out, rows

    def accum_device_ptr(self):
        self._need()
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().mrt_accum_device_ptr(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def padded_rows(self) -> int:
        self._need()
        n = C.c_uint32()
        _lib.check(_lib.lib().mrt_padded_rows(self._ctx, C.byref(n)))
        return n.value

    def bind_accum(self, dev_ptr, nbytes):
        """Render into caller-owned device memory (a torch tensor's data_ptr()) from now on."""
        self._need()
        _lib.check(_lib.lib().mrt_bind_accum(self._ctx, C.c_void_p(dev_ptr), nbytes))
        self._bound = (dev_ptr, nbytes) if dev_ptr else None

    def set_accum_device(self, dev_ptr, count):
        self._need()
        _lib.check(_lib.lib().mrt_set_accum_device(self._ctx, C.c_void_p(dev_ptr), int(count)))

    def create(self, render):
        """Create the context without rendering (Sampler::new + scene upload)."""
        self._ensure(render)
        return self

    def set_accum(self, rgb, count):
        self._need()
        rgb = np.ascontiguousarray(rgb, np.float32)
        if rgb.shape != (self.nh, self.nw, 3):
            raise ValueError(f"expected {(self.nh, self.nw, 3)}, got {rgb.shape}")
        _lib.check(_lib.lib().mrt_set_accum(self._ctx, rgb.ctypes.data_as(C.POINTER(C.c_float)), int(count)))

    def reset(self):
        self._need()
        _lib.check(_lib.lib().mrt_reset(self._ctx))

    def stats(self) -> dict:
        self._need()
        st = _abi.Stats()
        _lib.check(_lib.lib().mrt_get_stats(self._ctx, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}


def raytrace(render: Render, *, seed=1, update=None, batch=None, **kw):
    """CLI::raytrace / HttpServer::raytrace (src/cli.rs:155-177, src/http.rs:136-148): the per-sample loop
    followed by img().  `update(sample_index, image)` mirrors --update; `batch` fuses that many passes
    into one launch when no per-sample image is needed."""
    s = Sampler(kw.pop("workers", 24), kw.pop("n_dim", 64), seed=seed, **kw)
    n = render.rt.sample
    if update is None:
        step = batch or n
        done = 0
        while done < n:
            k = min(step, n - done)
            s.execute(render, n_samples=k)
            done += k
    else:
        for i in range(n):
            s.execute(render)
            update(i, s.img())
    out = s.img()
    s.close()
    return out
