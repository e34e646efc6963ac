"""Offline check of the mesh triangle-BVH route against the reference's octree walk (both from the kernel headers, x86 build).

Usage: python tests/mesh_probe.py [n_meshes] [rays_per_mesh]
Rays: random, aimed at vertices / edge midpoints / centroids (ties between neighbouring triangles), axis-parallel,
starting inside the mesh, and grazing the octree cell boundaries.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_raytracer_amd import scenes  # noqa: E402
from micro_raytracer_amd._abi import build_desc  # noqa: E402
from micro_raytracer_amd.scene import load_render  # noqa: E402
from tests.emu import emu  # noqa: E402


def random_mesh(rng, kind):
    if kind == 0:
        return scenes.bumpy_mesh(int(rng.integers(40, 1400)))
    if kind == 1:
        return scenes.icosphere(int(rng.integers(0, 3)), float(rng.uniform(0.2, 1.0)), tuple(rng.uniform(0.5, 1.5, 3)))
    if kind == 2:      # soup of large triangles spanning many octree cells
        n = int(rng.integers(1, 60))
        return rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    if kind == 3:      # small triangles + a few huge ones + duplicates + degenerate ones
        n = int(rng.integers(8, 300))
        c = rng.uniform(-1, 1, (n, 1, 3))
        t = (c + rng.uniform(-0.08, 0.08, (n, 3, 3))).astype(np.float32)
        t[:: 17] = rng.uniform(-1.2, 1.2, t[:: 17].shape)
        t[3] = t[2]
        t[5, 1] = t[5, 0]
        return t
    # grid-aligned vertices (exactly on octree cell boundaries)
    n = int(rng.integers(8, 200))
    return (rng.integers(-4, 5, (n, 3, 3)) / 4.0).astype(np.float32)


def rays_for(rng, tris, n):
    v = tris.reshape(-1, 3).astype(np.float64)
    ext = np.abs(v).max() + 1e-3
    o = rng.uniform(-3 * ext, 3 * ext, (n, 3))
    k = rng.integers(0, 6, n)
    tgt = rng.uniform(-ext, ext, (n, 3))
    ti = rng.integers(0, len(tris), n)
    a, b, c = tris[ti, 0].astype(np.float64), tris[ti, 1].astype(np.float64), tris[ti, 2].astype(np.float64)
    tgt = np.where((k == 1)[:, None], a, tgt)
    tgt = np.where((k == 2)[:, None], 0.5 * (a + b), tgt)
    tgt = np.where((k == 3)[:, None], (a + b + c) / 3, tgt)
    inside = k == 4
    o = np.where(inside[:, None], rng.uniform(-0.5 * ext, 0.5 * ext, (n, 3)), o)
    d = tgt - o
    axis = k == 5
    ax = np.eye(3)[rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], n)[:, None]
    d = np.where(axis[:, None], ax, d)
    # some axis rays start exactly on cell boundaries
    snap = axis & (rng.random(n) < 0.5)
    o = np.where(snap[:, None], np.round(o / (ext / 4)) * (ext / 4), o)
    # grazing rays: start on the line through a triangle edge, far outside, run along the edge with a tiny tilt
    graz = rng.random(n) < 0.15
    e = b - a
    e /= np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-30)
    tilt = rng.normal(size=(n, 3)) * (10.0 ** rng.uniform(-7, -2, (n, 1)))
    o = np.where(graz[:, None], a - e * rng.uniform(0.5, 30, (n, 1)) * ext, o)
    d = np.where(graz[:, None], e + tilt, d)
    # far origins aimed at vertices / edges: rounding of the exact test grows with the distance
    far = (~graz) & (rng.random(n) < 0.25)
    fo = tgt + (o - tgt) / np.maximum(np.linalg.norm(o - tgt, axis=1, keepdims=True), 1e-30) * (10.0 ** rng.uniform(1, 5.5, (n, 1))) * ext
    o = np.where(far[:, None], fo, o)
    d = np.where(far[:, None], tgt - fo, d)
    d = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    return o.astype(np.float32), d.astype(np.float32)


def main():
    n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    L = emu.lib()
    L.emu_mesh_probe.restype = C.c_int
    total_bad = 0
    for seed in range(n_meshes):
        rng = np.random.default_rng(seed)
        tris = random_mesh(rng, seed % 5)
        pos = [float(x) for x in rng.uniform(-0.5, 0.5, 3)] if seed % 3 else [0.0, 0.0, 0.0]
        desc = {"frame": {"res": [8, 8]}, "scene": {"renderer": [{"type": "mesh", "mesh": [[[float(c) for c in vv] for vv in t] for t in tris], "pos": pos}]}}
        h = build_desc(load_render(desc))
        o, d = rays_for(rng, tris, n_rays)
        o = np.ascontiguousarray(o + np.asarray(pos, np.float32))
        out = np.zeros((n_rays, 10), np.uint32)
        stats = (C.c_uint32 * 2)()
        bad = L.emu_mesh_probe(C.cast(h.ptr(), C.c_void_p), n_rays, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p), stats)
        ties = int(((out[:, 0] == 1) & (out[:, 1] == out[:, 3]) & (out[:, 2] != out[:, 4])).sum())
        print(f"mesh {seed:3d} kind {seed % 5} tris {len(tris):5d} tbvh {stats[1]} hits {stats[0]:6d} equal-t ties {ties:5d} mismatches {bad}", flush=True)
        if bad < 0:
            raise SystemExit(f"probe failed: {bad}")
        if bad:
            idx = np.nonzero((out[:, :5] != out[:, 5:]).any(axis=1))[0][:5]
            for i in idx:
                print("   ray", i, o[i], d[i], out[i])
        total_bad += bad
    print("total mismatches", total_bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
