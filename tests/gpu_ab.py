"""Ad-hoc A/B timing of experiment builds (not a test): MRT_LIB selects the library."""
import json, os, subprocess, sys
libs = sys.argv[1:]
code = r'''
import sys, json
sys.path.insert(0, ".")
from micro_raytracer_amd import Sampler, load_render, scenes
def run(name, desc, spp, reps=3):
    r = load_render(desc); s = Sampler(seed=1); s.execute(r, n_samples=1); best = 1e9
    for _ in range(reps):
        s.reset(); s.execute(r, n_samples=spp); best = min(best, s.stats()["kernel_ms"])
    print(f"  {name}: {best:.3f} ms  {s.nw*s.nh*spp/best/1e3:.0f} Msamples/s", flush=True)
run("cornell 1080p x32", scenes.cornell_box(res=(1920,1080), sample=32), 32)
run("cornell 512 x64", scenes.cornell_box(res=(512,512), sample=64), 64)
run("cornell2 1080sq x16", scenes.cornell_box2(res=(1080,1080), ssaa=1, sample=16), 16)
run("default 720p x16", scenes.default_scene(sample=16), 16)
run("minecraft 480x270 ssaa2 x4", scenes.minecraft_like(res=(480,270), ssaa=2, sample=4), 4)
run("mesh 480x270 x4", scenes.mesh_scene(res=(480,270), sample=4), 4)
run("instance 640x360 x4", scenes.instance_grid(res=(640,360), sample=4), 4)
run("sink 640x400 x16", scenes.kitchen_sink(res=(640,400), sample=16), 16)
'''
for lib in libs:
    print(lib, flush=True)
    env = dict(os.environ); 
    if lib != "default": env["MRT_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code], env=env, timeout=300)
