"""Ad-hoc A/B timing of launch shapes / experiment builds (not a test).  usage: gpu_ab.py [lib ...] ; env MRT_BLOCK_THREADS"""
import os, subprocess, sys
libs = sys.argv[1:] or ["default"]
code = r'''
import sys, os
sys.path.insert(0, ".")
from micro_raytracer_amd import Sampler, load_render, scenes
def run(name, desc, spp, reps=3, **kw):
    r = load_render(desc); s = Sampler(seed=1, **kw); s.execute(r, n_samples=1); best = 1e9
    for _ in range(reps):
        s.reset(); s.execute(r, n_samples=spp); st = s.stats(); best = min(best, st["kernel_ms"])
    print(f"  {name}: {best:.3f} ms  {s.nw*s.local_rows*spp/best/1e3:.0f} Msamples/s  block {st['block_threads']} lds {st['lds_bytes']}", flush=True)
run("cornell 1080p x32", scenes.cornell_box(res=(1920,1080), sample=32), 32)
run("cornell 1080p x256 shard 1/8", scenes.cornell_box(res=(1920,1080), sample=256), 256, shard_index=3, shard_count=8)
run("cornell 512 x64", scenes.cornell_box(res=(512,512), sample=64), 64)
run("cornell2 1080sq x16", scenes.cornell_box2(res=(1080,1080), ssaa=1, sample=16), 16)
run("default 720p x16", scenes.default_scene(sample=16), 16)
run("minecraft 480x270 ssaa2 x4", scenes.minecraft_like(res=(480,270), ssaa=2, sample=4), 4)
run("mesh 480x270 x4", scenes.mesh_scene(res=(480,270), sample=4), 4)
run("instance 640x360 x4", scenes.instance_grid(res=(640,360), sample=4), 4)
run("sink 640x400 x16", scenes.kitchen_sink(res=(640,400), sample=16), 16)
'''
for lib in libs:
    for bt in os.environ.get("AB_BLOCKS", "0").split(","):
        print(lib, "MRT_BLOCK_THREADS", bt, flush=True)
        env = dict(os.environ)
        if lib != "default": env["MRT_LIB"] = os.path.abspath(lib)
        if bt != "0": env["MRT_BLOCK_THREADS"] = bt
        subprocess.run([sys.executable, "-c", code], env=env, timeout=300)
