"""Ad-hoc GPU probe (not a test): throughput of one shard of an N-GPU frame for several sample-split factors."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_raytracer_amd import Sampler, scenes  # noqa: E402
from micro_raytracer_amd.scene import load_render  # noqa: E402

render = load_render(scenes.cornell_box(res=(1920, 1080), sample=1024, bounce=8))
for world in [int(x) for x in os.environ.get("WORLDS", "8,4,2").split(",")]:
    for ks in (os.environ.get("KS", "0,2,4,8,16")).split(","):
        if ks != "0":
            os.environ["MRT_K_SPLIT"] = ks
        else:
            os.environ.pop("MRT_K_SPLIT", None)
        s = Sampler(seed=1, device=0, shard_index=0, shard_count=world).create(render)
        s.execute(render, n_samples=64)
        t = time.perf_counter()
        s.execute(render, n_samples=1024)
        dt = time.perf_counter() - t
        st = s.stats()
        print(json.dumps({"world": world, "k_split_env": ks, "k_split": st["k_split"], "kernel_ms": round(st["kernel_ms"], 2), "wall_ms": round(dt * 1e3, 2),
                          "shard_Gsamples_s": round(st["samples"] / dt / 1e9, 3), "equiv_whole_frame_Gs": round(st["samples"] * world / dt / 1e9, 3)}), flush=True)
        s.close()
