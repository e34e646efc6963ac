"""Ad-hoc ablation timing (not a test)."""
import os, subprocess, sys
code = r'''
import sys
sys.path.insert(0, ".")
from micro_raytracer_amd import Sampler, load_render, scenes
r = load_render(scenes.cornell_box(res=(1920,1080), sample=32)); s = Sampler(seed=1); s.execute(r, n_samples=1); best = 1e9
for _ in range(3):
    s.reset(); s.execute(r, n_samples=32); st = s.stats(); best = min(best, st["kernel_ms"])
print(f"  cornell 1080p x32: {best:.3f} ms  seg/sample {st['segments']/(s.nw*s.nh*32):.2f}", flush=True)
'''
for lib in sys.argv[1:]:
    print(lib, flush=True)
    env = dict(os.environ); env["MRT_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code], env=env, timeout=300)
