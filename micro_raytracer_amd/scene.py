"""Loader for the reference's render-description JSON (host-side plumbing for the harness).

Mirrors the serde schema, its defaults and `Wrapper<T>::unwrap` of the reference
(/root/reference src/parser.rs:16-166 types, :188-271 defaults, :713-733 hex colours,
:620-628 / :674-682 gzip+base64 inline assets, :838-864 instance list).  The reference's
front-end stays Rust (BASELINE.json north_star); this module only exists so that tests,
bench.py and the Python `Sampler` can be driven by the same JSON files.

Numbers follow serde_json -> f32: parsed as f64, rounded once to f32.
"""
from __future__ import annotations

import base64
import gzip
import json
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

f32 = np.float32


def _v3(v):
    a = np.asarray(v, dtype=np.float64).astype(np.float32)
    if a.shape != (3,):
        raise ValueError(f"expected [f32;3], got {v!r}")
    return a


def _v4(v):
    a = np.asarray(v, dtype=np.float64).astype(np.float32)
    if a.shape != (4,):
        raise ValueError(f"expected [f32;4] (w,x,y,z), got {v!r}")
    return a


def parse_color(c):
    """ColorWrapper::unwrap, src/parser.rs:713-733: '#rrggbb' -> byte/255, or [r,g,b]."""
    if isinstance(c, str):
        if not c.startswith("#"):
            raise ValueError(f"{c} is not a hex color!")
        v = int(c[1:7], 16)
        return np.array([f32((v >> 16) & 255) / f32(255.0), f32((v >> 8) & 255) / f32(255.0),
                         f32(v & 255) / f32(255.0)], dtype=np.float32)
    return _v3(c)


@dataclass
class Texture:
    """rt::Texture, src/rt.rs:82-86."""
    w: int
    h: int
    dat: Optional[np.ndarray]  # (w*h, 3) f32, row-major x + y*w

    @staticmethod
    def from_json(obj, base_dir="."):
        """TextureWrapper -> Texture, src/parser.rs:659-711, 770-784."""
        if isinstance(obj, dict):
            dat = obj.get("dat")
            arr = None if dat is None else np.asarray(dat, dtype=np.float64).astype(np.float32).reshape(-1, 3)
            return Texture(int(obj.get("w", 0)), int(obj.get("h", 0)), arr)
        if isinstance(obj, str):
            if "." in obj:  # file name (src/parser.rs:688-689)
                from PIL import Image
                im = Image.open(os.path.join(base_dir, obj))
                if im.mode != "RGB":
                    raise ValueError("is not rgb888 image!")
                a = np.asarray(im, dtype=np.uint8)
                dat = (a.astype(np.float32) / f32(255.0)).reshape(-1, 3)
                return Texture(a.shape[1], a.shape[0], dat)
            text = gzip.decompress(base64.b64decode(obj)).decode("utf-8")
            return Texture.from_json(json.loads(text), base_dir)
        raise ValueError(f"bad texture: {type(obj)}")

    def to_inline(self) -> str:
        """TextureWrapper::to_inline, src/parser.rs:698-710."""
        obj = {"w": self.w, "h": self.h, "dat": None if self.dat is None else [[float(x) for x in t] for t in self.dat]}
        return base64.b64encode(gzip.compress(json.dumps(obj).encode(), 9)).decode()


@dataclass
class Material:
    """rt::Material with MaterialWrapper defaults, src/rt.rs:89-103, src/parser.rs:242-259."""
    albedo: np.ndarray = field(default_factory=lambda: np.ones(3, np.float32))
    rough: float = 0.0
    metal: float = 0.0
    glass: float = 0.0
    opacity: float = 1.0
    emit: float = 0.0
    tex: Optional[Texture] = None
    rmap: Optional[Texture] = None
    mmap: Optional[Texture] = None
    gmap: Optional[Texture] = None
    omap: Optional[Texture] = None
    emap: Optional[Texture] = None

    @staticmethod
    def from_json(obj, base_dir=".", tex_cache=None):
        m = Material()
        if obj is None:
            return m
        if "albedo" in obj:
            m.albedo = parse_color(obj["albedo"])
        for k in ("rough", "metal", "glass", "opacity", "emit"):
            if k in obj:
                setattr(m, k, float(f32(obj[k])))
        for k in ("tex", "rmap", "mmap", "gmap", "omap", "emap"):
            if obj.get(k) is not None:
                src = obj[k]
                key = src if isinstance(src, str) else None
                if tex_cache is not None and key is not None and key in tex_cache:
                    setattr(m, k, tex_cache[key])
                else:
                    t = src if isinstance(src, Texture) else Texture.from_json(src, base_dir)
                    if tex_cache is not None and key is not None:
                        tex_cache[key] = t
                    setattr(m, k, t)
        return m


BACKWARD = np.array([-0.0, -0.0, -1.0, -0.0], np.float32)  # Vec4f::backward(), src/lin.rs:143-145


def load_obj(path) -> np.ndarray:
    """MeshWrapper::load, src/parser.rs:602-618: first object / first group, first three vertices of each face."""
    pos, tris, groups_seen = [], [], 0
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                pos.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] in ("o", "g"):
                if tris:
                    groups_seen += 1
                if groups_seen:
                    break
            elif p[0] == "f":
                idx = []
                for tok in p[1:4]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(pos) + i)
                tris.append([pos[idx[0]], pos[idx[1]], pos[idx[2]]])
    return np.asarray(tris, dtype=np.float64).astype(np.float32).reshape(-1, 3, 3)


def mesh_from_json(obj, base_dir=".") -> np.ndarray:
    """MeshWrapper -> triangles (n,3,3), src/parser.rs:601-657."""
    if isinstance(obj, str):
        if "." in obj:
            return load_obj(os.path.join(base_dir, obj))
        text = gzip.decompress(base64.b64decode(obj)).decode("utf-8")
        return mesh_from_json(json.loads(text), base_dir)
    return np.asarray(obj, dtype=np.float64).astype(np.float32).reshape(-1, 3, 3)


def mesh_to_inline(tris: np.ndarray) -> str:
    obj = [[[float(c) for c in v] for v in t] for t in np.asarray(tris, np.float32).reshape(-1, 3, 3)]
    return base64.b64encode(gzip.compress(json.dumps(obj).encode(), 9)).decode()


@dataclass
class Renderer:
    """rt::Renderer, src/rt.rs:153-158 (aabb is never read by the reference)."""
    kind: str
    r: float = 0.0
    n: Optional[np.ndarray] = None
    sizes: Optional[np.ndarray] = None
    vtx: Optional[np.ndarray] = None
    mesh: Optional[np.ndarray] = None
    mat: Material = field(default_factory=Material)
    inst: List[Tuple[np.ndarray, np.ndarray]] = field(default_factory=list)
    name: Optional[str] = None

    @staticmethod
    def from_json(obj, base_dir=".", tex_cache=None):
        kind = obj["type"]
        r = Renderer(kind=kind)
        if kind == "sphere":
            r.r = float(f32(obj["r"]))
        elif kind == "plane":
            r.n = _v3(obj["n"])
        elif kind == "box":
            r.sizes = _v3(obj["sizes"])
        elif kind == "triangle":
            r.vtx = np.asarray(obj["vtx"], dtype=np.float64).astype(np.float32).reshape(3, 3)
        elif kind == "mesh":
            r.mesh = mesh_from_json(obj["mesh"], base_dir)
        else:
            raise ValueError(f"`{kind}` type is unxpected!")
        r.mat = Material.from_json(obj.get("mat"), base_dir, tex_cache)
        r.name = obj.get("name")
        # Wrapper<Renderer>::unwrap, src/parser.rs:838-853
        pos, direc, inst = obj.get("pos"), obj.get("dir"), obj.get("inst")
        if inst is not None:
            lst = [(_v3(p), _v4(d)) for p, d in inst]
            if pos is not None or direc is not None:
                lst.insert(0, (_v3(pos) if pos is not None else np.zeros(3, np.float32),
                               _v4(direc) if direc is not None else BACKWARD.copy()))
            r.inst = lst
        else:
            r.inst = [(_v3(pos) if pos is not None else np.zeros(3, np.float32),
                       _v4(direc) if direc is not None else BACKWARD.copy())]
        return r


@dataclass
class Light:
    """rt::Light, src/rt.rs:161-175; defaults src/parser.rs:261-271."""
    kind: str = "point"
    v: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    pwr: float = 0.5
    color: np.ndarray = field(default_factory=lambda: np.ones(3, np.float32))

    @staticmethod
    def from_json(obj):
        l = Light()
        l.kind = obj.get("type", "point")
        if l.kind == "point":
            l.v = _v3(obj["pos"])
        elif l.kind == "dir":
            l.v = _v3(obj["dir"])
        else:
            raise ValueError(f"`{l.kind}` type is unxpected!")
        if "pwr" in obj:
            l.pwr = float(f32(obj["pwr"]))
        if "color" in obj:
            l.color = parse_color(obj["color"])
        return l


@dataclass
class Sky:
    color: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    pwr: float = 0.5


@dataclass
class Camera:
    """rt::Camera, defaults src/parser.rs:198-210 (pos = -forward = (-0,-1,-0), dir = (w,x,y,z) = (0,0,1,0))."""
    pos: np.ndarray = field(default_factory=lambda: np.array([-0.0, -1.0, -0.0], np.float32))
    dir: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 1.0, 0.0], np.float32))
    fov: float = 70.0
    gamma: float = 0.8
    exp: float = 0.2
    aprt: float = 0.001
    foc: float = 100.0


@dataclass
class Frame:
    res: Tuple[int, int] = (1280, 720)
    ssaa: float = 1.0
    cam: Camera = field(default_factory=Camera)

    @property
    def nw(self):
        return int(f32(self.res[0]) * f32(self.ssaa))  # src/sampler.rs:29

    @property
    def nh(self):
        return int(f32(self.res[1]) * f32(self.ssaa))


@dataclass
class RayTracer:
    bounce: int = 8
    sample: int = 16
    loss: float = 0.15


@dataclass
class Scene:
    renderer: List[Renderer] = field(default_factory=list)
    light: List[Light] = field(default_factory=list)
    sky: Sky = field(default_factory=Sky)


@dataclass
class Render:
    """rt::Render, src/rt.rs:10-14."""
    rt: RayTracer = field(default_factory=RayTracer)
    frame: Frame = field(default_factory=Frame)
    scene: Scene = field(default_factory=Scene)


def load_render(src, base_dir=".") -> Render:
    """RenderWrapper (serde) + unwrap, src/parser.rs:160-166, 929-937.  `src`: dict, JSON text or a file path."""
    if isinstance(src, (str, os.PathLike)) and os.path.exists(str(src)):
        base_dir = os.path.dirname(os.path.abspath(str(src)))
        with open(src) as f:
            src = json.load(f)
    elif isinstance(src, str):
        src = json.loads(src)
    out = Render()
    rt = src.get("rt", {})
    out.rt = RayTracer(int(rt.get("bounce", 8)), int(rt.get("sample", 16)), float(f32(rt.get("loss", 0.15))))
    fr = src.get("frame", {})
    cam = fr.get("cam", {})
    c = Camera()
    if "pos" in cam:
        c.pos = _v3(cam["pos"])
    if "dir" in cam:
        c.dir = _v4(cam["dir"])
    for k in ("fov", "gamma", "exp", "aprt", "foc"):
        if k in cam:
            setattr(c, k, float(f32(cam[k])))
    res = fr.get("res", (1280, 720))
    if not (0 <= int(res[0]) < 65536 and 0 <= int(res[1]) < 65536):
        raise ValueError("res must fit u16")
    out.frame = Frame((int(res[0]), int(res[1])), float(f32(fr.get("ssaa", 1.0))), c)
    sc = src.get("scene", {})
    tex_cache = {}
    out.scene.renderer = [Renderer.from_json(o, base_dir, tex_cache) for o in (sc.get("renderer") or [])]
    out.scene.light = [Light.from_json(o) for o in (sc.get("light") or [])]
    sky = sc.get("sky", {})
    out.scene.sky = Sky(parse_color(sky["color"]) if "color" in sky else np.zeros(3, np.float32),
                        float(f32(sky.get("pwr", 0.5))))
    return out


def dump_render(r: Render) -> dict:
    """The `-v -d` JSON dump (src/bin/raytrace.rs:36-40) of the wrapper form; scalars as Python floats."""
    def fl(a):
        return [float(x) for x in a]

    def tex(t):
        if t is None:
            return None
        return {"w": t.w, "h": t.h, "dat": None if t.dat is None else [fl(x) for x in t.dat]}

    rend = []
    for o in r.scene.renderer:
        e = {"type": o.kind}
        if o.kind == "sphere":
            e["r"] = float(o.r)
        elif o.kind == "plane":
            e["n"] = fl(o.n)
        elif o.kind == "box":
            e["sizes"] = fl(o.sizes)
        elif o.kind == "triangle":
            e["vtx"] = [fl(v) for v in o.vtx]
        else:
            e["mesh"] = [[fl(v) for v in t] for t in o.mesh]
        m = o.mat
        e["mat"] = {"albedo": fl(m.albedo), "rough": m.rough, "metal": m.metal, "glass": m.glass,
                    "opacity": m.opacity, "emit": m.emit, "tex": tex(m.tex), "rmap": tex(m.rmap),
                    "mmap": tex(m.mmap), "gmap": tex(m.gmap), "omap": tex(m.omap), "emap": tex(m.emap)}
        e["inst"] = [[fl(p), fl(d)] for p, d in o.inst]
        e["name"] = o.name
        rend.append(e)
    lights = []
    for l in r.scene.light:
        lights.append({"type": l.kind, ("pos" if l.kind == "point" else "dir"): fl(l.v), "pwr": l.pwr, "color": fl(l.color)})
    cam = r.frame.cam
    return {
        "rt": {"bounce": r.rt.bounce, "sample": r.rt.sample, "loss": r.rt.loss},
        "frame": {"res": list(r.frame.res), "ssaa": r.frame.ssaa,
                  "cam": {"pos": fl(cam.pos), "dir": fl(cam.dir), "fov": cam.fov, "gamma": cam.gamma,
                          "exp": cam.exp, "aprt": cam.aprt, "foc": cam.foc}},
        "scene": {"renderer": rend or None, "light": lights or None,
                  "sky": {"color": fl(r.scene.sky.color), "pwr": r.scene.sky.pwr}},
    }
