import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from micro_raytracer_amd import Sampler, load_render, scenes
for name, d, spp in (("instance_grid_1000", scenes.instance_grid(res=(1920,1080), sample=64), 64),):
    r = load_render(d); s = Sampler(seed=1, flags=1)
    s.execute(r, n_samples=spp); t0=time.perf_counter(); s.execute(r, n_samples=spp); dt=time.perf_counter()-t0
    st=s.stats(); print(json.dumps({"name": name, "Msamples_s": round(st["samples"]/dt/1e6,1), "block": st["block_threads"], "lds": st["lds_bytes"]}))
