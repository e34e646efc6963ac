"""Offline fuzz (not collected by pytest): the kernel headers on x86 against the oracle for many seeds, including
"crowd" scenes whose instance counts switch on the instance BVH.  Usage: python tests/fuzz_offline.py FIRST LAST [crowd]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import make_holder  # noqa: E402
from test_fuzz_scenes import random_scene, crowd_scene, _check  # noqa: E402


def main():
    import importlib
    first, last = int(sys.argv[1]), int(sys.argv[2])
    crowd = len(sys.argv) > 3 and sys.argv[3] == "crowd"
    oracle_mod = importlib.import_module("oracle.oracle")
    from tests.emu import emu as emu_mod
    bad = 0
    n_bvh = 0
    for seed in range(first, last):
        desc = crowd_scene(seed) if crowd else random_scene(seed)
        render, h = make_holder(desc)
        spp = render.rt.sample
        o = oracle_mod.Oracle(h, seed=seed)
        o.execute(spp)
        ref, _ = o.accum()
        got, _ = emu_mod.render(h, seed, spp)
        import ctypes as C
        n_bvh += bool(emu_mod.lib().emu_features(C.cast(h.ptr(), C.c_void_p)) & 16)
        try:
            _check(got, ref, spp)
            o.set_accum(got, spp)
            ss, out = emu_mod.img(h, got, spp)
            assert np.array_equal(ss, o.img_ss()) and np.array_equal(out, o.img())
        except AssertionError as e:
            bad += 1
            print("MISMATCH seed", seed, e, flush=True)
    print(f"seeds {first}..{last} crowd={crowd}: {bad} mismatches, {n_bvh} scenes with an instance BVH", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
