"""Edge cases of the reference's semantics (empty and ragged frames, bounce 0, camera inside a box, ssaa < 1, lone
triangles, total internal reflection, signed-zero rays, many lights, legal out-of-range materials) through the same
parity bars as the main scenes: CPU (x86 build of the kernel headers) and GPU."""
import numpy as np
import pytest

from conftest import make_holder
from edge_cases import cases

CASES = cases()


def _check(name, got, seg, o, spp):
    ref, _ = o.accum()
    assert (np.isnan(got) == np.isnan(ref)).all(), name
    fin = np.isfinite(ref)
    if fin.any():
        assert np.abs(got[fin] - ref[fin]).max() / spp <= 1e-4, name
    assert (got[~fin & ~np.isnan(ref)] == ref[~fin & ~np.isnan(ref)]).all()


@pytest.mark.parametrize("name", list(CASES))
def test_edge_case_kernel_headers_on_x86(name, oracle_mod, emu_mod):
    render, h = make_holder(CASES[name])
    spp = render.rt.sample
    o = oracle_mod.Oracle(h, seed=17)
    o.execute(spp)
    got, seg = emu_mod.render(h, 17, spp)
    _check(name, got, seg, o, spp)
    o.set_accum(got, spp)
    ss, out = emu_mod.img(h, got, spp)
    assert np.array_equal(ss, o.img_ss()) and np.array_equal(out, o.img())


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_edge_case_gpu(name, oracle_mod):
    from micro_raytracer_amd import Sampler
    render, h = make_holder(CASES[name])
    spp = render.rt.sample
    o = oracle_mod.Oracle(h, seed=17)
    o.execute(spp)
    s = Sampler(seed=17, flags=1)          # MRT_FLAG_COUNT_SEGMENTS
    s.execute(render, n_samples=spp)
    got, cnt = s.accum()
    assert cnt == spp
    _check(name, got, s.stats()["segments"], o, spp)
    o.set_accum(got, cnt)
    assert np.array_equal(s.img_ss(), o.img_ss()) and np.array_equal(s.img(), o.img())
    s.close()
