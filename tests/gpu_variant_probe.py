"""Ad-hoc GPU probe (not a test): Minecraft-shaped scene with parts of its work removed, to see where the time goes."""
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_raytracer_amd import Sampler, scenes  # noqa: E402
from micro_raytracer_amd.scene import load_render  # noqa: E402


def run(name, desc, spp):
    render = load_render(desc)
    s = Sampler(seed=1, device=0, flags=1).create(render)
    s.execute(render, n_samples=spp)
    t = time.perf_counter()
    s.execute(render, n_samples=spp)
    dt = time.perf_counter() - t
    st = s.stats()
    print(json.dumps({"name": name, "ms": round(dt * 1e3, 1), "Msamples_s": round(st["samples"] / dt / 1e6, 1), "seg_per_sample": round(st["segments"] / st["samples"], 2),
                      "Gseg_s": round(st["segments"] / dt / 1e9, 2), "block": st["block_threads"], "lds": st["lds_bytes"]}), flush=True)
    s.close()


base = scenes.minecraft_like(res=(1920, 1080), ssaa=2, sample=32)
run("minecraft", base, 32)
d = copy.deepcopy(base)
for r in d["scene"]["renderer"]:
    for k in ("tex", "rmap", "mmap", "gmap", "omap", "emap"):
        r.get("mat", {}).pop(k, None)
run("minecraft without maps", d, 32)
d2 = copy.deepcopy(base)
d2["scene"]["light"] = []
run("minecraft without the light (no shadow rays)", d2, 32)
d3 = copy.deepcopy(d)
d3["scene"]["light"] = []
run("minecraft without maps and light", d3, 32)
m = scenes.mesh_scene(res=(1920, 1080), sample=64, n_tris=1280)
run("mesh", m, 64)
m2 = copy.deepcopy(m)
m2["scene"]["light"] = []
run("mesh without the light", m2, 64)
if os.environ.get("MID"):
    for n in (300, 500, 700):
        run(f"mesh {n} tris", scenes.mesh_scene(res=(1920, 1080), sample=64, n_tris=n), 64)
if os.environ.get("BIG"):
    for n in (1600, 2500, 5120):
        d = scenes.mesh_scene(res=(1920, 1080), sample=16)
        d["scene"]["renderer"][0]["mesh"] = [[[float(c) for c in v] for v in t] for t in scenes.icosphere(4, 0.45, (1.3, 1.0, 1.1))[:n]]
        run(f"mesh {n} tris (icosphere 4)", d, 16)
