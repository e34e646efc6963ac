"""Ad-hoc check of the workgroup-cooperative kernels against the per-lane ones (not a test): bits and speed.  Run under `timeout`."""
import os, subprocess, sys
code = r'''
import sys, os, hashlib, statistics
sys.path.insert(0, ".")
import numpy as np
from micro_raytracer_amd import Sampler, load_render, scenes
which = sys.argv[1]
out = {}
cases = [("mesh", scenes.mesh_scene(res=(256, 144), sample=20), 20), ("mesh5k", scenes.mesh_scene(res=(200, 120), sample=18, n_tris=5120), 18),
         ("sink", scenes.kitchen_sink(res=(160, 100), sample=17), 17)]
for name, desc, spp in cases:
    r = load_render(desc); s = Sampler(seed=3); s.execute(r, n_samples=spp); a, c = s.accum(); st = s.stats()
    print(which, name, hashlib.sha1(np.nan_to_num(a).tobytes()).hexdigest()[:16], "feat", st["kernel_features"], "block", st["block_threads"], "lds", st["lds_bytes"], flush=True)
if len(sys.argv) > 2:
    for name, desc, spp, reps in (("mesh 1080p x512", scenes.mesh_scene(res=(1920,1080), sample=512), 512, 3), ("mesh5k 1080p x64", scenes.mesh_scene(res=(1920,1080), sample=64, n_tris=5120), 64, 6), ("mesh20k 540p x64", scenes.mesh_scene(res=(960,540), sample=64, n_tris=20480), 64, 6)):
        r = load_render(desc); s = Sampler(seed=1); ts = []
        for i in range(reps + 2):
            s.execute(r, n_samples=spp); st = s.stats(); s.reset()
            if i >= 2: ts.append(st["kernel_ms"])
        t = statistics.median(ts)
        print(f"  {which} {name}: {t:.2f} ms  {s.nw*s.local_rows*spp/t/1e3:.0f} Msamples/s feat {st['kernel_features']} lds {st['lds_bytes']}", flush=True)
'''
for which, env in (("percall-lane", {"MRT_COOP": "0"}), ("coop", {})) + tuple(("coop:" + os.path.basename(l), {"MRT_LIB": os.path.abspath(l)}) for l in os.environ.get("COOP_LIBS", "").split()):
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, "-c", code, which] + sys.argv[1:], env=e, timeout=600)
