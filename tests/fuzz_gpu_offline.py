"""Offline GPU fuzz (not collected by pytest): libmrt_hip.so against the oracle on many random and crowd scenes.
Usage (on the GPU box): python tests/fuzz_gpu_offline.py FIRST LAST [crowd|ident]
`ident`: random scenes with every instance untransformed and an axis-aligned pinhole camera on a lattice point half of the time
(the F_IDENT kernels; rays and shifted origins with zero components take the reference's mat-vecs)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import make_holder  # noqa: E402
from test_fuzz_scenes import random_scene, crowd_scene, ident_scene, _check  # noqa: E402


def main():
    import importlib
    from micro_raytracer_amd import Sampler
    first, last = int(sys.argv[1]), int(sys.argv[2])
    crowd = len(sys.argv) > 3 and sys.argv[3] == "crowd"
    ident = len(sys.argv) > 3 and sys.argv[3] == "ident"

    oracle_mod = importlib.import_module("oracle.oracle")
    bad = 0
    worst = 0.0
    shapes = {}
    for seed in range(first, last):
        render, h = make_holder(ident_scene(seed) if ident else (crowd_scene(seed) if crowd else random_scene(seed)))
        spp = render.rt.sample
        o = oracle_mod.Oracle(h, seed=seed)
        o.execute(spp)
        ref, _ = o.accum()
        s = Sampler(seed=seed).create(render)
        s.execute(render, n_samples=spp)
        got, cnt = s.accum()
        st = s.stats()
        key = (st["block_threads"], st["kernel_features"]) if ident else st["block_threads"]
        shapes[key] = shapes.get(key, 0) + 1
        try:
            _check(got, ref, spp)
            fin = np.isfinite(ref)
            if fin.any():
                worst = max(worst, float(np.abs(got[fin] - ref[fin]).max()) / spp)
            o.set_accum(got, cnt)
            assert np.array_equal(s.img_ss(), o.img_ss()) and np.array_equal(s.img(), o.img())
        except AssertionError as e:
            bad += 1
            print("MISMATCH seed", seed, e, flush=True)
        s.close()
        o.close()
    print(f"GPU fuzz seeds {first}..{last} crowd={crowd} ident={ident}: {bad} mismatches, worst L-inf on mean radiance {worst:.3e}, launch shapes {shapes}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
