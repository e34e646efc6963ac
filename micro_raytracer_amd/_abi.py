"""ctypes mirror of include/mrt.h (the C-ABI boundary) and the flattening of a loaded
render description into the POD `mrt_render_desc`.

The same descriptor feeds libmrt_hip.so (product) and, in tests only, the CPU oracle.
Reference types mirrored: rt::Render and everything under it, /root/reference src/rt.rs:10-190.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 3

KIND_SPHERE, KIND_PLANE, KIND_BOX, KIND_TRIANGLE, KIND_MESH = range(5)
LIGHT_POINT, LIGHT_DIR = 0, 1
KIND_IDS = {"sphere": KIND_SPHERE, "plane": KIND_PLANE, "box": KIND_BOX, "triangle": KIND_TRIANGLE, "mesh": KIND_MESH}

MRT_OK, MRT_ERR_ARG, MRT_ERR_SCENE, MRT_ERR_DEVICE, MRT_ERR_LIMIT, MRT_ERR_STATE = 0, -1, -2, -3, -4, -5
FLAG_COUNT_SEGMENTS, FLAG_NO_EVENT_TIMING, FLAG_DEFER, FLAG_NO_LOOKAHEAD = 1, 2, 4, 8


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 4), ("fov", C.c_float), ("gamma", C.c_float),
                ("exp", C.c_float), ("aprt", C.c_float), ("foc", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("res_w", C.c_uint16), ("res_h", C.c_uint16), ("ssaa", C.c_float), ("cam", Camera)]


class Rt(C.Structure):
    _fields_ = [("bounce", C.c_uint32), ("sample", C.c_uint32), ("loss", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("dat", C.POINTER(C.c_float))]


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("rough", C.c_float), ("metal", C.c_float), ("glass", C.c_float),
                ("opacity", C.c_float), ("emit", C.c_float), ("tex", C.c_int32), ("rmap", C.c_int32),
                ("mmap", C.c_int32), ("gmap", C.c_int32), ("omap", C.c_int32), ("emap", C.c_int32)]


class Instance(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 4)]


class Renderer(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("param", C.c_float * 9), ("tris", C.POINTER(C.c_float)),
                ("n_tris", C.c_uint32), ("mat", Material), ("inst", C.POINTER(Instance)), ("n_inst", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("v", C.c_float * 3), ("pwr", C.c_float), ("color", C.c_float * 3)]


class Sky(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("pwr", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("renderer", C.POINTER(Renderer)), ("n_renderer", C.c_uint32), ("light", C.POINTER(Light)),
                ("n_light", C.c_uint32), ("sky", Sky), ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint32)]


class RenderDesc(C.Structure):
    _fields_ = [("rt", Rt), ("frame", Frame), ("scene", Scene)]


class Opts(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("seed", C.c_uint64), ("device", C.c_int32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("shard_rows", C.c_uint32),
                ("n_devices", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("gather_ms", C.c_double), ("samples", C.c_uint64),
                ("segments", C.c_uint64), ("launches", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("block_threads", C.c_uint32), ("scene_bytes", C.c_uint32), ("k_split", C.c_uint32), ("deferred", C.c_uint32), ("img_ms", C.c_double),
                ("reduce_ms", C.c_double), ("kernel_features", C.c_uint32), ("scene_in_lds", C.c_uint32)]


class Plan(C.Structure):
    _fields_ = [("staging", C.c_uint32), ("block_threads", C.c_uint32), ("lds_bytes", C.c_uint32), ("staged_bytes", C.c_uint32),
                ("scene_bytes", C.c_uint32), ("kernel_features", C.c_uint32), ("tbvh_nodes", C.c_uint32), ("tbvh_hot_nodes", C.c_uint32),
                ("small_plain_grid", C.c_uint32), ("walk_cap", C.c_uint32), ("reserved", C.c_uint32 * 2)]


STAGING = ("all", "warm", "deep", "none")
MAP_SLOTS = ("tex", "rmap", "mmap", "gmap", "omap", "emap")


def _f3(dst, src):
    for i in range(3):
        dst[i] = float(np.float32(src[i]))


def _f4(dst, src):
    for i in range(4):
        dst[i] = float(np.float32(src[i]))


class DescHolder:
    """Owns a RenderDesc together with every array it points into."""

    def __init__(self):
        self.desc = RenderDesc()
        self.keep = []

    def ptr(self):
        return C.byref(self.desc)


def build_desc(render) -> DescHolder:
    """Flatten a loaded render (micro_raytracer_amd.scene.Render) into mrt_render_desc.

    Renderer order and per-renderer instance order are preserved: ties in the reference's
    closest-hit `min_by` resolve to the first candidate (src/rt.rs:872).
    """
    h = DescHolder()
    d = h.desc
    d.rt.bounce = int(render.rt.bounce)
    d.rt.sample = int(render.rt.sample)
    d.rt.loss = float(np.float32(render.rt.loss))
    fr = render.frame
    d.frame.res_w, d.frame.res_h = int(fr.res[0]), int(fr.res[1])
    d.frame.ssaa = float(np.float32(fr.ssaa))
    cam = fr.cam
    _f3(d.frame.cam.pos, cam.pos)
    _f4(d.frame.cam.dir, cam.dir)
    for k in ("fov", "gamma", "exp", "aprt", "foc"):
        setattr(d.frame.cam, k, float(np.float32(getattr(cam, k))))

    sc = render.scene
    # textures: de-duplicated by object identity
    tex_index = {}
    tex_list = []

    def tex_id(t):
        if t is None:
            return -1
        key = id(t)
        if key not in tex_index:
            tex_index[key] = len(tex_list)
            tex_list.append(t)
        return tex_index[key]

    rends = (Renderer * max(1, len(sc.renderer)))()
    for i, r in enumerate(sc.renderer):
        o = rends[i]
        o.kind = KIND_IDS[r.kind]
        params = np.zeros(9, np.float32)
        if r.kind == "sphere":
            params[0] = r.r
        elif r.kind == "plane":
            params[:3] = r.n
        elif r.kind == "box":
            params[:3] = r.sizes
        elif r.kind == "triangle":
            params[:] = np.asarray(r.vtx, np.float32).reshape(9)
        for k in range(9):
            o.param[k] = float(params[k])
        if r.kind == "mesh":
            tris = np.ascontiguousarray(np.asarray(r.mesh, np.float32).reshape(-1, 9))
            h.keep.append(tris)
            o.tris = tris.ctypes.data_as(C.POINTER(C.c_float))
            o.n_tris = tris.shape[0]
        m = r.mat
        _f3(o.mat.albedo, m.albedo)
        for k in ("rough", "metal", "glass", "opacity", "emit"):
            setattr(o.mat, k, float(np.float32(getattr(m, k))))
        for k in MAP_SLOTS:
            setattr(o.mat, k, tex_id(getattr(m, k)))
        insts = (Instance * max(1, len(r.inst)))()
        for j, (pos, direc) in enumerate(r.inst):
            _f3(insts[j].pos, pos)
            _f4(insts[j].dir, direc)
        h.keep.append(insts)
        o.inst = C.cast(insts, C.POINTER(Instance))
        o.n_inst = len(r.inst)
    h.keep.append(rends)
    d.scene.renderer = C.cast(rends, C.POINTER(Renderer))
    d.scene.n_renderer = len(sc.renderer)

    lights = (Light * max(1, len(sc.light)))()
    for i, l in enumerate(sc.light):
        lights[i].kind = LIGHT_POINT if l.kind == "point" else LIGHT_DIR
        _f3(lights[i].v, l.v)
        lights[i].pwr = float(np.float32(l.pwr))
        _f3(lights[i].color, l.color)
    h.keep.append(lights)
    d.scene.light = C.cast(lights, C.POINTER(Light))
    d.scene.n_light = len(sc.light)

    _f3(d.scene.sky.color, sc.sky.color)
    d.scene.sky.pwr = float(np.float32(sc.sky.pwr))

    texs = (Texture * max(1, len(tex_list)))()
    for i, t in enumerate(tex_list):
        texs[i].w, texs[i].h = int(t.w), int(t.h)
        if t.dat is not None:
            arr = np.ascontiguousarray(np.asarray(t.dat, np.float32).reshape(-1, 3))
            h.keep.append(arr)
            texs[i].dat = arr.ctypes.data_as(C.POINTER(C.c_float))
    h.keep.append(texs)
    d.scene.textures = C.cast(texs, C.POINTER(Texture))
    d.scene.n_textures = len(tex_list)
    return h
