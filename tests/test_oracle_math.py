"""The math / RNG contract of the oracle: accuracy against libm, and agreement with the x86 build of the
device header (tests/emu) bit for bit.  The GPU side of the same check is tests/test_gpu_parity.py."""
import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_sin_cos_accuracy(oracle_mod):
    x = np.linspace(0, 2 * np.pi, 400001).astype(np.float32)
    for op, f in ((0, np.sin), (1, np.cos)):
        got = oracle_mod.math(op, x)
        ref = f(x.astype(np.float64))
        err = np.abs(got - ref)
        # absolute error relative to 1 ulp of the unit-magnitude result range
        assert err.max() < 1.5e-7, (op, err.max())


def test_acos_accuracy(oracle_mod):
    x = np.linspace(-1, 1, 400001).astype(np.float32)
    got = oracle_mod.math(2, x)
    ref = np.arccos(x.astype(np.float64)).astype(np.float32)
    assert ulp_diff(got, ref).max() <= 4
    assert oracle_mod.math(2, np.array([1.0], np.float32))[0] == 0.0
    assert np.isnan(oracle_mod.math(2, np.array([1.5, np.nan], np.float32))).all()


def test_atan2_accuracy(oracle_mod):
    rng = np.random.default_rng(1)
    y = rng.normal(size=200000).astype(np.float32)
    x = rng.normal(size=200000).astype(np.float32)
    got = oracle_mod.math(3, y, x)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(got - ref).max() < 6e-7
    assert oracle_mod.math(3, np.array([0.0], np.float32), np.array([0.0], np.float32))[0] == 0.0


def test_powf_is_correctly_rounded_almost_everywhere(oracle_mod):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(0, 3, 300000), rng.uniform(0, 1e-3, 50000), rng.uniform(0, 1e4, 50000)]).astype(np.float32)
    y = np.concatenate([rng.uniform(0.2, 2.5, 300000), np.full(50000, 0.5), np.full(50000, 0.8)]).astype(np.float32)
    got = oracle_mod.math(4, x, y)
    ref = np.power(x.astype(np.float64), y.astype(np.float64)).astype(np.float32)
    d = ulp_diff(got, ref)
    assert d.max() <= 1
    assert (d != 0).mean() < 1e-4
    edge_x = np.array([0.0, 0.0, 1.0, np.inf, -1.0, np.nan, 2.0], np.float32)
    edge_y = np.array([0.8, 0.0, 5.0, 0.5, 0.5, 1.0, np.inf], np.float32)
    e = oracle_mod.math(4, edge_x, edge_y)
    assert e[0] == 0 and e[1] == 1 and e[2] == 1 and np.isinf(e[3]) and np.isnan(e[4]) and np.isnan(e[5]) and np.isinf(e[6])


def test_device_header_matches_oracle_bit_for_bit_on_x86(oracle_mod, emu_mod):
    rng = np.random.default_rng(3)
    n = 100000
    cases = {
        0: (rng.uniform(0, 2 * np.pi, n), None), 1: (rng.uniform(0, 2 * np.pi, n), None), 2: (rng.uniform(-1, 1, n), None),
        3: (rng.normal(size=n), rng.normal(size=n)), 4: (rng.uniform(0, 4, n), rng.uniform(0.2, 2.5, n)),
    }
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-42, 3.4e38, 65536.0, 7e4], np.float32)
    for op, (a, b) in cases.items():
        a = np.concatenate([a.astype(np.float32), edge])
        b = None if b is None else np.concatenate([b.astype(np.float32), edge[::-1]])
        g, o = emu_mod.math(op, a, b), oracle_mod.math(op, a, b)
        assert ((g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))).all(), op


def test_rng_contract(oracle_mod):
    """Counter RNG: pure function of (seed, pixel, sample, dim); uniform on the 23-bit lattice; decorrelated."""
    L = oracle_mod.lib()
    pk = oracle_mod.path_key(1, 12345, 7)
    assert pk == oracle_mod.path_key(1, 12345, 7)
    assert len({oracle_mod.path_key(1, 12345, s) for s in range(4096)}) == 4096      # bijective in the sample index
    u = np.array([L.orc_draw_f32(oracle_mod.path_key(9, p, s), d) for p in range(40) for s in range(40) for d in range(10)], np.float64)
    assert 0.0 <= u.min() and u.max() < 1.0
    assert np.all(u * 2**23 == np.round(u * 2**23))
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    a = u.reshape(-1, 10)
    c = np.corrcoef(a.T)
    assert np.abs(c - np.eye(10)).max() < 0.08


def test_contract_v3_polar_angle(oracle_mod):
    """Math contract v3 (oracle D8): RayTracer::rand's polar angle (reference src/rt.rs:997-1003) as
    cos th = 1 - 2 u1, sin th = sqrt(4 u1 (1 - u1)), measured over EVERY value u1 can take (the 2^23 lattice of
    rand's Uniform<f32>) against (a) the literal reading -- th = acos(1 - 2 u1), then sin th, cos th through the contract's
    own functions -- and (b) the f64 truth.  This is the committed bound the oracle's restatement may deviate by; a
    rewrite of rt_rand that is less innocent fails here."""
    u = (np.arange(1 << 23, dtype=np.float64) * 2.0 ** -23).astype(np.float32)
    s3, c3 = oracle_mod.math(8, u), oracle_mod.math(9, u)
    sl, cl = oracle_mod.math(10, u), oracle_mod.math(11, u)
    u64 = u.astype(np.float64)
    ct = 1.0 - 2.0 * u64
    st = np.sqrt(4.0 * u64 * (1.0 - u64))
    d_lit = max(np.abs(s3.astype(np.float64) - sl).max(), np.abs(c3.astype(np.float64) - cl).max())
    e_sin, e_cos = np.abs(s3 - st).max(), np.abs(c3 - ct).max()
    e_sin_lit, e_cos_lit = np.abs(sl - st).max(), np.abs(cl - ct).max()
    print(f"v3 vs literal {d_lit:.3e}; vs f64: sin {e_sin:.3e} cos {e_cos:.3e} (literal: {e_sin_lit:.3e} / {e_cos_lit:.3e})")
    assert d_lit <= 3e-7                       # measured 2.98e-7: inside the <= 2 ulp libm-vs-contract divergence D2
    assert e_cos == 0.0 and e_sin <= 6e-8      # cos exact on the lattice, sin one rounding + a correctly rounded root (5.05e-8)
    assert e_sin <= e_sin_lit and e_cos <= e_cos_lit      # never further from the truth than the literal composition
    assert np.all(s3 >= 0.0) and np.all(s3 <= 1.0) and np.all(np.abs(c3) <= 1.0)
    # unit length to f32 rounding, as sin^2 + cos^2 of one angle would be
    assert np.abs(s3.astype(np.float64) ** 2 + ct ** 2 - 1.0).max() <= 2e-7


def test_oracle_build_takes_mfma_only_from_the_host(oracle_mod):
    """oracle/Makefile passes -mfma iff the build host's CPU has FMA (without it __builtin_fmaf goes through libm's
    correctly rounded fmaf: same bits, no SIGILL); the contract's fused steps are exact either way."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(oracle_mod.__file__))
    cmd = subprocess.run(["make", "-C", here, "-n", "-B", "liboracle.so"], capture_output=True, text=True, check=True).stdout
    flags = open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
    assert ("-mfma" in cmd.split()) == ("fma" in flags)
    assert "-ffp-contract=off" in cmd
    a = np.array([1.0 + 2.0 ** -12], np.float32)
    assert oracle_mod.math(0, a)[0] == np.float32(np.sin(np.float64(a[0])))      # a fused-step function still rounds as agreed
