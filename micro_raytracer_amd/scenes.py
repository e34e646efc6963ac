"""Programmatic render descriptions shaped like the reference's example scenes.

The reference's example/*.json files do not travel with this repository, so every scene
is regenerated here from its parameters; binary assets (mesh, textures)
are replaced by procedural ones of the same size and kind.  All builders return the
reference's JSON schema as a dict: feed it to scene.load_render().
"""
from __future__ import annotations

import math

import numpy as np


def _frame(res, ssaa, cam):
    return {"res": [int(res[0]), int(res[1])], "ssaa": ssaa, "cam": cam}


def default_scene(res=(1280, 720), ssaa=1, sample=16, bounce=8):
    """example/Default.json: 1 sphere + 1 point light (BASELINE.json configs[0])."""
    return {
        "rt": {"bounce": bounce, "sample": sample, "loss": 0.15},
        "frame": _frame(res, ssaa, {"pos": [0, -1, 0], "dir": [0, 0, 1, 0], "fov": 70, "gamma": 0.8, "exp": 0.2}),
        "scene": {
            "renderer": [{"type": "sphere", "r": 0.5}],
            "light": [{"type": "point", "pos": [-0.5, -1, 0.5], "pwr": 0.5, "color": "#ffffff"}],
            "sky": {"color": "#000000", "pwr": 0.5},
        },
    }


def cornell_box(res=(512, 512), ssaa=1, sample=64, bounce=8, floor_z=-0.2):
    """example/CornellBox.json: 5 planes + 5 spheres, no lights (BASELINE.json configs[1])."""
    return {
        "rt": {"sample": sample, "bounce": bounce},
        "frame": _frame(res, ssaa, {"exp": 0.75, "fov": 60, "gamma": 0.5, "pos": [0, -1.2, 0.1]}),
        "scene": {"renderer": [
            {"type": "plane", "n": [0, -1, 0], "pos": [0, 1, 0], "mat": {"rough": 1}},
            {"type": "plane", "n": [1, 0, 0], "pos": [-1, 0, 0], "mat": {"albedo": "#ff0000", "rough": 1}},
            {"type": "plane", "n": [-1, 0, 0], "pos": [1, 0, 0], "mat": {"albedo": "#00ff00", "rough": 1}},
            {"type": "plane", "n": [0, 0, -1], "pos": [0, 0, 1], "mat": {"rough": 1}},
            {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, floor_z], "mat": {"rough": 1}},
            {"type": "sphere", "r": 0.2, "pos": [-0.15, -0.5, 0], "mat": {"glass": 0.08, "opacity": 0}},
            {"type": "sphere", "r": 0.2, "pos": [0.5, 0, 0], "mat": {"metal": 1}},
            {"type": "sphere", "r": 0.2, "pos": [0, 0.5, 0], "mat": {"albedo": "#ff0000"}},
            {"type": "sphere", "r": 0.2, "pos": [-0.5, 0, 0], "mat": {"rough": 1}},
            {"type": "sphere", "r": 0.2, "pos": [0.5, 0.5, 0], "mat": {"albedo": "#ffc177", "emit": 1.0}},
        ]},
    }


def cornell_box2(res=(1080, 1080), ssaa=2, sample=512, bounce=8):
    """example/CornellBox2.json: 7 boxes (one rotated, one emissive) + 1 sphere (configs[2], [3])."""
    return {
        "rt": {"sample": sample, "bounce": bounce},
        "frame": _frame(res, ssaa, {"pos": [0, -1.25, 0], "exp": 0.8, "fov": 60, "gamma": 0.6}),
        "scene": {"renderer": [
            {"type": "box", "sizes": [0.01, 1, 1], "pos": [0.5, 0, 0], "mat": {"albedo": "#00ff00"}},
            {"type": "box", "sizes": [0.01, 1, 1], "pos": [-0.5, 0, 0], "mat": {"albedo": "#ff0000"}},
            {"type": "box", "sizes": [1, 1, 0.01], "pos": [0, 0, -0.5]},
            {"type": "box", "sizes": [1, 1, 0.01], "pos": [0, 0, 0.5]},
            {"type": "box", "sizes": [1, 0.01, 1], "pos": [0, 0.5, 0]},
            {"type": "box", "sizes": [0.3, 0.3, 0.01], "pos": [0, 0, 0.499], "mat": {"emit": 1}},
            {"type": "box", "sizes": [0.25, 0.25, 0.25], "pos": [0, 0, -0.375], "dir": [0, 0.5, 0.5, 0]},
            {"type": "sphere", "r": 0.15, "pos": [0, 0, -0.1]},
        ]},
    }


def checker_texture(w=64, h=64, cell=8, a=(1.0, 1.0, 1.0), b=(40 / 255.0, 40 / 255.0, 40 / 255.0)):
    """A w x h checker as the reference's texture buffer JSON; values are exact k/255."""
    dat = []
    for y in range(h):
        for x in range(w):
            dat.append(list(a if ((x // cell) + (y // cell)) % 2 == 0 else b))
    return {"w": w, "h": h, "dat": dat}


def floor_checker():
    """The 64x64 floor texture of example/dof.json and example/Mesh.json as parameters: an 8-texel checker of
    64/255 (top-left cell) and 150/255 greys (the asset carries +-3/255 of compression noise on top)."""
    return checker_texture(64, 64, 8, a=(64 / 255.0,) * 3, b=(150 / 255.0,) * 3)


def icosphere(subdiv=2, radius=0.45, squash=(1.5, 0.93, 1.08)):
    """Closed triangle mesh, 20 * 4^subdiv triangles (subdiv 2 -> 320, 3 -> 1280)."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    sq = np.array(squash, float)
    tris = np.array([[v[a] * radius * sq, v[b] * radius * sq, v[c] * radius * sq] for a, b, c in f], np.float32)
    return tris


def bumpy_mesh(n_target=967, seed=7):
    """~n_target triangles: an icosphere(3) with lobes, truncated to exactly n_target triangles
    (so the mesh is open, like a scanned asset; the reference's Mesh.json has 967)."""
    tris = icosphere(3, 0.42, (1.6, 1.0, 1.15)).astype(np.float64)
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * math.pi, 3)
    p = tris.reshape(-1, 3)
    r = 1.0 + 0.12 * np.sin(7 * p[:, 0] + ph[0]) * np.sin(6 * p[:, 1] + ph[1]) + 0.08 * np.sin(9 * p[:, 2] + ph[2])
    p = p * r[:, None]
    tris = p.reshape(-1, 3, 3)
    return tris[:n_target].astype(np.float32)


def big_mesh(n_tris):
    """A closed icosphere-based mesh of the first n_tris triangles of the smallest subdivision that has them (5120 -> 4,
    20480 -> 5): the shape of the reference's Mesh.json scene with a mesh the LDS cannot hold."""
    sub = 0
    while 20 * 4 ** sub < n_tris:
        sub += 1
    return icosphere(sub, 0.45, (1.3, 1.0, 1.1))[:n_tris]


def mesh_scene(res=(1920, 1080), ssaa=1, sample=256, bounce=8, n_tris=967, inline=False):
    """example/Mesh.json shape: ~1k-triangle mesh (octree depth 3) + textured plane + point light (configs[4]).
    n_tris > 1280: the same scene around big_mesh(n_tris) (any mesh size is legal input, src/parser.rs:805-824)."""
    from .scene import mesh_to_inline
    tris = bumpy_mesh(n_tris) if n_tris <= 1280 else big_mesh(n_tris)
    mesh = mesh_to_inline(tris) if inline else (np.asarray(tris, np.float32) if n_tris > 1280 else [[[float(c) for c in vv] for vv in t] for t in tris])
    return {
        "rt": {"sample": sample, "bounce": bounce},
        "frame": _frame(res, ssaa, {"aprt": 0.008, "foc": 0.725, "fov": 60}),
        "scene": {
            "renderer": [
                {"type": "mesh", "mesh": mesh, "pos": [0, 0.5, 0], "mat": {"rough": 1}},
                {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.5], "mat": {"rough": 1, "tex": floor_checker()}},
            ],
            "light": [{"type": "point", "pos": [-0.5, -1, 0.5], "pwr": 0.5, "color": "#ffffff"}],
        },
    }


def _atlas_texture(w=64, h=48, seed=0, base=(0.6, 0.45, 0.3)):
    """Procedural 4x3 box cross atlas with k/255 texels."""
    rng = np.random.default_rng(seed)
    noise = rng.integers(-30, 31, size=(h, w))
    dat = []
    for y in range(h):
        for x in range(w):
            face = (x // (w // 4)) + 4 * (y // (h // 3))
            k = [int(min(255, max(0, round(255 * base[c]) + noise[y, x] + 6 * face))) for c in range(3)]
            dat.append([k[0] / 255.0, k[1] / 255.0, k[2] / 255.0])
    return {"w": w, "h": h, "dat": dat}


def _scalar_map(w, h, fn):
    dat = []
    for y in range(h):
        for x in range(w):
            k = fn(x, y)
            dat.append([k / 255.0, k / 255.0, k / 255.0])
    return {"w": w, "h": h, "dat": dat}


def minecraft_like(res=(1920, 1080), ssaa=2, sample=512, bounce=8):
    """example/Minecraft.json shape (configs[4]): 9 renderers, 85 instances (84 boxes + ground plane),
    64x48 cross-atlas textures, one opacity-mapped door, one emit-mapped rotated torch, dir light, sky."""
    bdir = [0, 0, -1, 0]
    oak, cobble, mossy, slab, stair = [], [], [], [], []
    # a small hut: cobble walls (36), oak frame / roof (20), mossy (3), slabs (11), stairs (11)
    for x in range(-2, 3):
        for z in range(0, 3):
            if len(cobble) < 36 and not (x == 0 and z < 2):
                cobble.append([[float(x), 2.0, float(z)], bdir])
    for x in (-2, 2):
        for y in (0, 1):
            for z in range(0, 3):
                if len(cobble) < 36:
                    cobble.append([[float(x), float(y), float(z)], bdir])
    for x in range(-2, 3):
        for y in (0, 1):
            if len(cobble) < 36:
                cobble.append([[float(x), float(y), -1.0], bdir])
    k = 0
    while len(cobble) < 36:
        cobble.append([[4.0 + k, 3.0, 0.0], bdir])
        k += 1
    for x in range(-2, 3):
        for y in range(0, 3):
            if len(oak) < 20:
                oak.append([[float(x), float(y), 3.0], bdir])
    k = 0
    while len(oak) < 20:
        oak.append([[-4.0, 1.0 + k, 0.0], bdir])
        k += 1
    mossy = [[[-3.0, 3.0, 0.0], bdir], [[3.0, 4.0, 0.0], bdir], [[1.0, 5.0, 0.0], bdir]]
    for i in range(11):
        slab.append([[-3.0 + 0.6 * i, -1.0 - 0.1 * i, -0.25 + 0.5 * (i % 2)], bdir])
        stair.append([[-2.5 + 0.5 * i, 4.0, 3.25 + 0.5 * (i % 3)], bdir])
    doors_lo = [[[0.0, 2.4, 0.0], bdir]]
    doors_hi = [[[0.0, 2.4, 1.0], bdir]]
    torch = [[[1.0, 1.45, 1.5], [0, 0, 1, -0.8]]]
    omap = _scalar_map(64, 48, lambda x, y: 0 if (x // 4 + y // 4) % 3 == 0 else 255)
    emap = _scalar_map(8, 30, lambda x, y: 255 if y < 8 else 0)
    rend = [
        {"type": "box", "sizes": [1, 1, 1], "mat": {"rough": 1, "tex": _atlas_texture(seed=1, base=(0.62, 0.5, 0.3))}, "inst": oak},
        {"type": "box", "sizes": [1, 1, 1], "mat": {"rough": 1, "tex": _atlas_texture(seed=2, base=(0.5, 0.5, 0.5))}, "inst": cobble},
        {"type": "box", "sizes": [1, 1, 1], "mat": {"rough": 1, "tex": _atlas_texture(seed=3, base=(0.4, 0.5, 0.4))}, "inst": mossy},
        {"type": "box", "sizes": [1, 1, 0.5], "mat": {"rough": 1, "tex": _atlas_texture(seed=4, base=(0.55, 0.55, 0.55))}, "inst": slab},
        {"type": "box", "sizes": [1, 0.5, 0.5], "mat": {"rough": 1, "tex": _atlas_texture(seed=5, base=(0.6, 0.48, 0.3))}, "inst": stair},
        {"type": "box", "sizes": [1, 0.2, 1], "mat": {"rough": 1, "tex": _atlas_texture(seed=6, base=(0.5, 0.35, 0.2))}, "inst": doors_lo},
        {"type": "box", "sizes": [1, 0.2, 1], "mat": {"rough": 1, "tex": _atlas_texture(seed=7, base=(0.5, 0.35, 0.2)), "omap": omap}, "inst": doors_hi},
        {"type": "box", "sizes": [0.125, 0.125, 0.625],
         "mat": {"rough": 1, "tex": _atlas_texture(8, 30, seed=8, base=(0.8, 0.6, 0.2)), "emap": emap}, "inst": torch},
        {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.5], "mat": {"rough": 1, "tex": _atlas_texture(16, 16, seed=9, base=(0.3, 0.55, 0.25))}},
    ]
    return {
        "rt": {"sample": sample, "bounce": bounce},
        "frame": _frame(res, ssaa, {"dir": [0, 0.5, 0.5, 0], "fov": 70, "pos": [-3, -1, 1]}),
        "scene": {
            "renderer": rend,
            "light": [{"type": "dir", "dir": [-0.1, 1, -0.5], "color": "#ffe594", "pwr": 0.7}],
            "sky": {"color": "#517eb2", "pwr": 0.3},
        },
    }


def instance_grid(res=(1280, 720), ssaa=1, sample=16, bounce=8, n=10):
    """example/Instance.json: one sphere renderer with n^3 instances on a lattice of step 0.5."""
    inst = [[[0.5 * x, 0.5 * y, 0.5 * z], [0, 0, -1, 0]] for x in range(n) for y in range(n) for z in range(n)]
    return {
        "rt": {"sample": sample, "bounce": bounce},
        "frame": _frame(res, ssaa, {"pos": [2.25, -4, 2.25]}),
        "scene": {
            "renderer": [{"type": "sphere", "r": 0.2, "inst": inst}],
            "light": [{"type": "point", "pos": [-0.5, -1, 0.5], "pwr": 0.5, "color": "#ffffff"}],
            "sky": {"color": "#000000", "pwr": 0.5},
        },
    }


def dof_scene(res=(1280, 720), ssaa=1, sample=256, bounce=8):
    """example/dof.json: 2 spheres, 1 box, textured plane, strong DoF, rolled camera (rotate_y path)."""
    return {
        "rt": {"sample": sample, "bounce": bounce},
        "frame": _frame(res, ssaa, {"aprt": 0.008, "foc": 0.525, "fov": 60, "pos": [0, -1, 0.25], "dir": [0, 0, 1, -0.25]}),
        "scene": {
            "renderer": [
                {"type": "sphere", "r": 0.3, "mat": {"albedo": "#ee8c57"}},
                {"type": "sphere", "r": 0.3, "pos": [-1.125, 1.25, 0], "mat": {"metal": 1}},
                {"type": "box", "sizes": [0.6, 0.6, 0.6], "pos": [1.125, 1.25, 0], "mat": {"rough": 1, "albedo": "#f7a3d7"}},
                {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.3], "mat": {"rough": 1, "tex": floor_checker()}},
            ],
            "light": [{"type": "point", "pos": [-0.5, -1, 0.5], "pwr": 0.5, "color": "#ffffff"}],
        },
    }


def kitchen_sink(res=(96, 64), ssaa=1, sample=4, bounce=6):
    """Every primitive kind, rotated instances, all six texture maps, both light kinds: parity stress."""
    tri = [[0.6, 0.3, -0.25], [0.1, 0.35, 0.55], [-0.4, 0.25, -0.2]]
    small_mesh = icosphere(1, 0.22, (1.0, 1.0, 1.3))
    rmap = _scalar_map(8, 8, lambda x, y: 32 * ((x + y) % 8))
    mmap = _scalar_map(4, 4, lambda x, y: 255 if (x + y) % 2 else 0)
    gmap = _scalar_map(4, 4, lambda x, y: 20 * (x % 4))
    omap = _scalar_map(8, 8, lambda x, y: 0 if (x // 2 + y // 2) % 2 else 255)
    emap = _scalar_map(8, 8, lambda x, y: 255 if (x, y) in ((1, 1), (5, 6)) else (128 if x == 3 else 0))
    return {
        "rt": {"sample": sample, "bounce": bounce, "loss": 0.1},
        "frame": _frame(res, ssaa, {"pos": [0.1, -1.6, 0.35], "dir": [0.05, 0.1, 1, -0.15], "fov": 65, "gamma": 0.7, "exp": 0.5,
                                    "aprt": 0.01, "foc": 1.5}),
        "scene": {
            "renderer": [
                {"type": "plane", "n": [0, 0, 2], "pos": [0, 0, -0.4], "mat": {"rough": 1, "tex": checker_texture(16, 16, 2)}},
                {"type": "sphere", "r": 0.25, "pos": [-0.6, 0.2, -0.1], "dir": [0.3, 0.2, 1, 0.1],
                 "mat": {"tex": _atlas_texture(16, 12, seed=11), "rmap": rmap, "mmap": mmap}},
                {"type": "sphere", "r": 0.2, "mat": {"glass": 0.3, "opacity": 0.2, "gmap": gmap},
                 "inst": [[[0.0, -0.3, -0.15], [0, 0, -1, 0]], [[0.55, -0.2, -0.2], [0.2, 0.4, -1, 0.3]]]},
                {"type": "box", "sizes": [0.4, 0.3, 0.5], "pos": [0.6, 0.5, -0.1], "dir": [0.25, 0.6, 1, -0.2],
                 "mat": {"tex": _atlas_texture(32, 24, seed=12), "omap": omap, "emap": emap}},
                {"type": "box", "sizes": [0.3, 0.3, 0.3], "pos": [-0.15, 0.7, -0.25], "mat": {"metal": 0.6, "rough": 0.3, "albedo": "#80c0ff"}},
                {"type": "triangle", "vtx": tri, "pos": [0.0, 0.4, 0.2], "mat": {"albedo": "#ffd060", "rough": 0.5}},
                {"type": "mesh", "mesh": [[[float(c) for c in v] for v in t] for t in small_mesh], "pos": [-0.25, -0.1, 0.35],
                 "dir": [0.1, 0.3, -1, 0.2], "mat": {"albedo": "#c0ffc0", "rough": 0.8}},
                {"type": "sphere", "r": 0.08, "pos": [0.3, 0.0, 0.6], "mat": {"emit": 0.7, "albedo": "#fff0c0"}},
            ],
            "light": [
                {"type": "point", "pos": [-0.8, -1.0, 0.9], "pwr": 0.4, "color": "#ffe0c0"},
                {"type": "dir", "dir": [0.3, 0.5, -1.0], "pwr": 0.3, "color": "#c0d0ff"},
            ],
            "sky": {"color": [0.3, 0.4, 0.6], "pwr": 0.4},
        },
    }
