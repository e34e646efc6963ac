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
