"""The fast correctly rounded cores of the math contract (csrc/mrt_math.h: sqrt_, recip_, div_, recip_sqrt_) against
IEEE arithmetic.

The reference's sqrt / recip / divide are Rust f32 operations, i.e. IEEE correctly rounded (src/lin.rs:60-66,
src/rt.rs:335-359, 400-412); the oracle computes them with the host FPU.  On the device a wavefront whose operands all
lie inside the exponent window [2^-40, 2^40] runs the bare refinement sequences instead of the compiler's full
expansions.  Checked here:
  * on the device (mrt_selftest_sweep): every one of the 2^32 f32 bit patterns for sqrt and recip, 10^10 operand pairs
    for divide and 2^32 vectors for the norm scale -- fast core == compiler expansion, bit for bit;
  * against the host FPU (numpy, the arithmetic the oracle uses): every 2^32 pattern for sqrt and recip and 2^28 pairs
    for divide through mrt_selftest_math.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op,name,count", [(0, "sqrt", 1 << 32), (1, "recip", 1 << 32), (2, "divide", 10_000_000_000),
                                            (3, "norm scale", 1 << 32)])
def test_fast_cores_equal_the_compiler_expansions_on_device(op, name, count):
    from micro_raytracer_amd import _lib
    mis, ex = _lib.selftest_sweep(op, 0, count, seed=12345)
    print(f"{name}: {count} inputs, {mis} mismatches")
    assert mis == 0, f"{name}: {mis} mismatches, e.g. a={ex[0]!r} b={ex[1]!r} fast={ex[2]!r} ieee={ex[3]!r}"


def _same(g, o):
    return (g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))


def test_sqrt_and_recip_equal_the_host_fpu_on_every_f32():
    """All 2^32 bit patterns, 2^26 per call, against numpy's sqrt and 1/x (correctly rounded, the oracle's arithmetic)."""
    from micro_raytracer_amd import _lib
    step = 1 << 26
    one = np.float32(1.0)
    with np.errstate(all="ignore"):
        for first in range(0, 1 << 32, step):
            x = np.arange(first, first + step, dtype=np.uint64).astype(np.uint32).view(np.float32)
            g = _lib.selftest_math(6, x)
            bad = ~_same(g, np.sqrt(x))
            assert not bad.any(), f"sqrt: {np.count_nonzero(bad)} mismatches from pattern {first:#x}, x={x[bad][:3]} gpu={g[bad][:3]}"
            g = _lib.selftest_math(5, x)
            bad = ~_same(g, one / x)
            assert not bad.any(), f"recip: {np.count_nonzero(bad)} mismatches from pattern {first:#x}, x={x[bad][:3]} gpu={g[bad][:3]}"


def test_divide_equals_the_host_fpu_on_random_pairs():
    """2^28 pairs: half with exponents inside the fast window (whole wavefronts take the core), half raw bit patterns."""
    from micro_raytracer_amd import _lib
    rng = np.random.default_rng(7)
    n = 1 << 24
    with np.errstate(all="ignore"):
        for rep in range(16):
            if rep % 2 == 0:
                def draw():
                    e = rng.integers(127 - 40, 127 + 40, n, dtype=np.uint32)
                    return ((rng.integers(0, 2, n, dtype=np.uint32) << 31) | (e << 23) | rng.integers(0, 1 << 23, n, dtype=np.uint32)).view(np.float32)
                a, b = draw(), draw()
            else:
                a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
                b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
            g = _lib.selftest_math(7, a, b)
            bad = ~_same(g, a / b)
            assert not bad.any(), f"divide: {np.count_nonzero(bad)} mismatches, a={a[bad][:3]} b={b[bad][:3]} gpu={g[bad][:3]}"
