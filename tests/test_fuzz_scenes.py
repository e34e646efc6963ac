"""Seeded random scenes (every primitive kind, random materials / maps / rotations / lights / cameras) through the
parity bars: the kernel's lane code on x86 against the oracle for many seeds, a few of them on the GPU."""
import numpy as np
import pytest

from conftest import make_holder


def random_scene(seed):
    from micro_raytracer_amd import scenes
    rng = np.random.default_rng(seed)
    u = lambda a, b, n=None: rng.uniform(a, b, n)
    def vec(a, b): return [float(x) for x in u(a, b, 3)]
    def quat(): return [float(x) for x in np.r_[u(-0.9, 0.9), rng.normal(size=3)]] if rng.random() < 0.4 else [0, 0, -1, 0]
    def smap(w, h): return {"w": w, "h": h, "dat": [[float(k) / 255] * 3 for k in rng.integers(0, 256, w * h)]}
    def cmap(w, h): return {"w": w, "h": h, "dat": [[float(c) / 255 for c in rng.integers(0, 256, 3)] for _ in range(w * h)]}
    def mat(textured_ok):
        m = {"albedo": [float(x) for x in u(0.1, 1.0, 3)], "rough": float(rng.choice([0, 0.3, 1])), "metal": float(rng.choice([0, 0, 0.7, 1])),
             "glass": float(rng.choice([0, 0.1, 0.8])), "opacity": float(rng.choice([1, 1, 0.5, 0])), "emit": float(rng.choice([0, 0, 0, 0.4, 1]))}
        if textured_ok and rng.random() < 0.4:
            for k in ("tex", "rmap", "mmap", "gmap", "omap", "emap"):
                if rng.random() < 0.35:
                    m[k] = cmap(int(rng.integers(1, 9)), int(rng.integers(1, 9))) if k == "tex" else smap(int(rng.integers(1, 6)), int(rng.integers(1, 6)))
        return m
    rend = []
    for _ in range(int(rng.integers(1, 9))):
        kind = rng.choice(["sphere", "plane", "box", "triangle", "mesh"], p=[0.3, 0.2, 0.25, 0.15, 0.1])
        o = {"type": str(kind), "mat": mat(kind in ("sphere", "plane", "box"))}
        if kind == "sphere": o["r"] = float(u(0.1, 0.6))
        elif kind == "plane": o["n"] = vec(-1, 1)
        elif kind == "box": o["sizes"] = [float(x) for x in u(0.1, 1.2, 3)]
        elif kind == "triangle": o["vtx"] = [vec(-0.8, 0.8) for _ in range(3)]
        else: o["mesh"] = [[[float(c) for c in v] for v in t] for t in scenes.icosphere(int(rng.integers(0, 2)), float(u(0.2, 0.5)))]
        if rng.random() < 0.3:
            o["inst"] = [[vec(-1.5, 1.5), quat()] for _ in range(int(rng.integers(1, 4)))]
        else:
            o["pos"] = vec(-1.2, 1.2)
            o["dir"] = quat()
        rend.append(o)
    lights = []
    for _ in range(int(rng.integers(0, 3))):
        lights.append({"type": "point", "pos": vec(-2, 2), "pwr": float(u(0.1, 0.8)), "color": [float(x) for x in u(0.3, 1, 3)]} if rng.random() < 0.6
                      else {"type": "dir", "dir": vec(-1, 1), "pwr": float(u(0.1, 0.8)), "color": [float(x) for x in u(0.3, 1, 3)]})
    return {
        "rt": {"sample": int(rng.integers(1, 5)), "bounce": int(rng.integers(0, 7)), "loss": float(u(0, 0.5))},
        "frame": {"res": [int(rng.integers(3, 29)), int(rng.integers(3, 21))], "ssaa": float(rng.choice([1, 1, 2, 1.5, 0.75])),
                  "cam": {"pos": vec(-0.5, 0.5)[:1] + [float(u(-3, -1.5))] + [float(u(-0.3, 0.6))], "dir": [float(u(-0.3, 0.3)), float(u(-0.3, 0.3)), 1.0, float(u(-0.3, 0.3))],
                          "fov": float(u(40, 90)), "gamma": float(u(0.4, 1.0)), "exp": float(u(0.1, 0.85)), "aprt": float(rng.choice([0.0, 0.001, 0.02])), "foc": float(u(0.5, 50))}},
        "scene": {"renderer": rend, "light": lights, "sky": {"color": [float(x) for x in u(0, 0.6, 3)], "pwr": float(u(0, 1))}},
    }


def crowd_scene(seed):
    """random_scene with 4-40 extra instances per non-plane renderer: enough to switch on the instance BVH."""
    d = random_scene(seed)
    rng = np.random.default_rng(seed + 77777)
    for o in d["scene"]["renderer"]:
        if o["type"] == "plane":
            continue
        base = o.pop("inst", None) or [[o.pop("pos", [0, 0, 0]), o.pop("dir", [0, 0, -1, 0])]]
        o.pop("pos", None); o.pop("dir", None)
        inst = list(base)
        for _ in range(int(rng.integers(4, 40))):
            q = [float(x) for x in np.r_[rng.uniform(-0.9, 0.9), rng.normal(size=3)]] if rng.random() < 0.3 else [0, 0, -1, 0]
            inst.append([[float(x) for x in rng.uniform(-3, 3, 3)], q])
        o["inst"] = inst
    d["frame"]["res"] = [int(rng.integers(6, 20)), int(rng.integers(6, 14))]
    return d


def ident_scene(seed):
    """random_scene with every instance untransformed, on a half-unit lattice, no maps; half of the cameras are axis-aligned
    pinholes on a lattice point: rays and shifted origins with zero components everywhere (the F_IDENT kernels, mrt_trace.h)."""
    d = random_scene(seed)
    rng = np.random.default_rng(seed + 4242)
    keep = []
    for o in d["scene"]["renderer"]:
        if o["type"] in ("mesh", "triangle"):
            continue                                      # (plain kernels only: planes, spheres, boxes)
        o.pop("dir", None)
        if "inst" in o:
            o["inst"] = [[[float(round(c * 2) / 2) for c in i[0]], [0, 0, -1, 0]] for i in o["inst"]]
        elif "pos" in o:
            o["pos"] = [float(round(c * 2) / 2) for c in o["pos"]]
        o.get("mat", {}).pop("tex", None)
        for k in ("rmap", "mmap", "gmap", "omap", "emap"):
            o.get("mat", {}).pop(k, None)
        keep.append(o)
    d["scene"]["renderer"] = keep or [{"type": "sphere", "r": 0.5}]
    if rng.random() < 0.5:                                # camera on the lattice, looking straight down +y, pinhole
        cam = d["frame"]["cam"]
        cam["pos"] = [0.0, -2.0, 0.0]; cam["dir"] = [0, 0, 1, 0]; cam["aprt"] = 0.0
    return d


def _check(got, ref, spp):
    assert (np.isnan(got) == np.isnan(ref)).all()
    fin = np.isfinite(ref)
    if fin.any():
        scale = max(1.0, float(np.abs(ref[fin]).max()) / spp)
        assert np.abs(got[fin] - ref[fin]).max() / spp <= 1e-4 * scale


@pytest.mark.parametrize("seed", list(range(40)) + [1000 + k for k in range(8)] + [2000 + k for k in range(4)])
def test_fuzz_kernel_headers_on_x86(seed, oracle_mod, emu_mod):
    render, h = make_holder(ident_scene(seed - 2000) if seed >= 2000 else (crowd_scene(seed - 1000) if seed >= 1000 else random_scene(seed)))
    spp = render.rt.sample
    o = oracle_mod.Oracle(h, seed=seed)
    o.execute(spp)
    ref, _ = o.accum()
    got, _ = emu_mod.render(h, seed, spp)
    _check(got, ref, spp)
    o.set_accum(got, spp)
    ss, out = emu_mod.img(h, got, spp)
    assert np.array_equal(ss, o.img_ss()) and np.array_equal(out, o.img())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(0, 40, 3)) + [1000, 1001, 1002, 1004] + [2000 + k for k in range(10)])
def test_fuzz_gpu(seed, oracle_mod):
    from micro_raytracer_amd import Sampler
    render, h = make_holder(ident_scene(seed - 2000) if seed >= 2000 else (crowd_scene(seed - 1000) if seed >= 1000 else random_scene(seed)))
    spp = render.rt.sample
    o = oracle_mod.Oracle(h, seed=seed)
    o.execute(spp)
    ref, _ = o.accum()
    s = Sampler(seed=seed)
    s.execute(render, n_samples=spp)
    got, _ = s.accum()
    _check(got, ref, spp)
    o.set_accum(got, spp)
    assert np.array_equal(s.img_ss(), o.img_ss()) and np.array_equal(s.img(), o.img())
    s.close()
