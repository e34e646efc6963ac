import sys, json, os
sys.path.insert(0, ".")
from micro_raytracer_amd import Sampler, load_render, scenes, _lib
os.makedirs("gpurun_out/renders", exist_ok=True)
jobs = {"cornell": scenes.cornell_box(res=(640,360), sample=512, bounce=8),
        "cornell2": scenes.cornell_box2(res=(400,400), ssaa=2, sample=256),
        "minecraft": scenes.minecraft_like(res=(640,360), ssaa=2, sample=64),
        "mesh": scenes.mesh_scene(res=(640,360), sample=64),
        "dof": scenes.dof_scene(res=(640,360), sample=128),
        "sink": scenes.kitchen_sink(res=(480,320), sample=128)}
for name, d in jobs.items():
    r = load_render(d); s = Sampler(seed=1); s.execute(r, n_samples=r.rt.sample)
    _lib.save_image(f"gpurun_out/renders/{name}.png", s.img()); print(name, s.stats()["kernel_ms"]); s.close()
