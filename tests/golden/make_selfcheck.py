"""Self-consistency vectors: accumulators and images of the CPU oracle at a fixed seed for small instances of every
scene family.  These are NOT reference outputs (the reference is unseeded); they freeze the math / RNG contract so
that an accidental change of either shows up as a diff, and give the GPU tests committed expected values.

    python tests/golden/make_selfcheck.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SEED = 20240611
CASES = {
    "default": ("default_scene", dict(res=(40, 24), sample=2)),
    "cornell": ("cornell_box", dict(res=(32, 32), sample=3)),
    "cornell2": ("cornell_box2", dict(res=(20, 20), ssaa=2, sample=2)),
    "dof": ("dof_scene", dict(res=(40, 24), sample=2)),
    "instance": ("instance_grid", dict(res=(32, 18), sample=2, n=4)),
    "mesh": ("mesh_scene", dict(res=(32, 18), sample=2)),
    "minecraft": ("minecraft_like", dict(res=(32, 18), ssaa=1, sample=2)),
    "sink": ("kitchen_sink", dict(res=(36, 24), sample=3)),
}


def build(name):
    from micro_raytracer_amd import _abi, load_render, scenes
    fn, kw = CASES[name]
    render = load_render(getattr(scenes, fn)(**kw))
    return render, _abi.build_desc(render)


def main():
    from oracle import oracle
    out = {}
    for name in CASES:
        render, h = build(name)
        o = oracle.Oracle(h, seed=SEED)
        o.execute(render.rt.sample)
        acc, cnt = o.accum()
        out[f"{name}_acc"] = acc
        out[f"{name}_img"] = o.img()
        out[f"{name}_count"] = np.array(cnt)
    np.savez_compressed(os.path.join(HERE, "selfcheck.npz"), **out)


if __name__ == "__main__":
    main()
