"""Ad-hoc GPU probe used while developing (not a test): timing of the headline configs."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from micro_raytracer_amd import Sampler, load_render, scenes


def run(name, desc, spp, reps=3):
    r = load_render(desc)
    s = Sampler(seed=1, flags=1)
    s.execute(r, n_samples=1)
    best = None
    for _ in range(reps):
        s.reset()
        t = s.execute(r, n_samples=spp)
        st = s.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = dict(st, wall=t)
    n = s.nw * s.nh * spp
    print(json.dumps({"name": name, "nw": s.nw, "nh": s.nh, "spp": spp, "kernel_ms": best["kernel_ms"], "wall_s": best["wall"],
                      "Msamples_s": n / best["kernel_ms"] / 1e3, "seg_per_sample": best["segments"] / n,
                      "lds": best["lds_bytes"], "block": best["block_threads"]}), flush=True)
    s.close()


if __name__ == "__main__":
    run("C2 cornell 512x512x64 b8", scenes.cornell_box(res=(512, 512), sample=64), 64)
    run("cornell 1920x1080x16 b8", scenes.cornell_box(res=(1920, 1080), sample=16), 16)
    run("cornell2 1080x1080 ssaa1 x16 b8", scenes.cornell_box2(res=(1080, 1080), ssaa=1, sample=16), 16)
    run("default 1280x720x16", scenes.default_scene(sample=16), 16)
    run("mesh 480x270x4", scenes.mesh_scene(res=(480, 270), sample=4), 4)
    run("minecraft 480x270 ssaa2 x4", scenes.minecraft_like(res=(480, 270), ssaa=2, sample=4), 4)
    run("instance 640x360x4", scenes.instance_grid(res=(640, 360), sample=4), 4)
    big = scenes.mesh_scene(res=(960, 540), sample=8)
    big["scene"]["renderer"][0]["mesh"] = [[[float(c) for c in v] for v in t] for t in scenes.icosphere(5, 0.45, (1.3, 1.0, 1.1))]
    run("mesh 20480 tris 960x540x8 (scene in L2)", big, 8)
    run("instance 1000 spheres 1920x1080x8", scenes.instance_grid(res=(1920, 1080), sample=8), 8)
