"""Ad-hoc A/B timing of experiment builds on the round-4 workloads (not a test).
usage: gpu_ab4.py [lib.so ...]     env: AB_SET=bvh|headline|all, any MRT_* knob
Sustained figures: 2 warm-up launches, then the median kernel time of `reps` launches."""
import os, subprocess, sys
libs = sys.argv[1:] or ["default"]
code = r'''
import sys, os, statistics
sys.path.insert(0, ".")
from micro_raytracer_amd import Sampler, load_render, scenes
def run(name, desc, spp, reps=5, **kw):
    r = load_render(desc); s = Sampler(seed=1, **kw); ts = []
    for i in range(reps + 2):
        s.execute(r, n_samples=spp); st = s.stats(); s.reset()
        if i >= 2: ts.append(st["kernel_ms"])
    t = statistics.median(ts)
    print(f"  {name}: {t:.2f} ms (min {min(ts):.2f})  {s.nw*s.local_rows*spp/t/1e3:.0f} Msamples/s  block {st['block_threads']} lds {st['lds_bytes']} feat {st['kernel_features']} k_split {st['k_split']}", flush=True)
which = os.environ.get("AB_SET", "bvh")
if which in ("bvh", "all"):
    run("mesh 1080p x512", scenes.mesh_scene(res=(1920,1080), sample=512), 512, reps=3)
    run("minecraft 1080p ssaa2 x64", scenes.minecraft_like(res=(1920,1080), ssaa=2, sample=64), 64, reps=3)
    run("mesh5k 1080p x64", scenes.mesh_scene(res=(1920,1080), sample=64, n_tris=5120), 64, reps=8)
    run("mesh20k 540p x64", scenes.mesh_scene(res=(960,540), sample=64, n_tris=20480), 64, reps=8)
if which in ("headline", "all"):
    run("cornell 1080p x1024", scenes.cornell_box(res=(1920,1080), sample=1024), 1024, reps=3)
    run("cornell2 4k x64", scenes.cornell_box2(res=(1920,1080), ssaa=2, sample=64, bounce=16), 64)
if which in ("misc", "all"):
    run("instance 1080p x64", scenes.instance_grid(res=(1920,1080), sample=64), 64)
    run("sink 1080p x64", scenes.kitchen_sink(res=(1920,1080), sample=64), 64)
    run("default 1080p x256", scenes.default_scene(res=(1920,1080), sample=256), 256)
'''
for lib in libs:
    print(lib, flush=True)
    env = dict(os.environ)
    if lib != "default": env["MRT_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code], env=env, timeout=900)
