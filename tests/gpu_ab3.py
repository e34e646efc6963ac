"""Ad-hoc: large scenes at 1080p (not a test)."""
import sys
sys.path.insert(0, ".")
from micro_raytracer_amd import Sampler, load_render, scenes
def run(name, desc, spp, reps=2, **kw):
    r = load_render(desc); s = Sampler(seed=1, **kw); s.execute(r, n_samples=1); best = 1e9
    for _ in range(reps):
        s.reset(); s.execute(r, n_samples=spp); st = s.stats(); best = min(best, st["kernel_ms"])
    print(f"  {name}: {best:.3f} ms  {s.nw*s.local_rows*spp/best/1e3:.0f} Msamples/s  block {st['block_threads']} lds {st['lds_bytes']} k {st['k_split']} seg/sample {st['segments']/(s.nw*s.local_rows*spp):.2f}", flush=True)
run("minecraft 1080p ssaa2 x2", scenes.minecraft_like(res=(1920,1080), ssaa=2, sample=2), 2)
run("mesh 1080p x4", scenes.mesh_scene(res=(1920,1080), sample=4), 4)
run("instance 1080p x4", scenes.instance_grid(res=(1920,1080), sample=4), 4)
run("sink 1080p x8", scenes.kitchen_sink(res=(1920,1080), sample=8), 8)
run("dof 1080p x16", scenes.dof_scene(res=(1920,1080), sample=16), 16)
run("default 1080p x16", scenes.default_scene(res=(1920,1080), sample=16), 16)
