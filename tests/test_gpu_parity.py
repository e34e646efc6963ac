"""GPU parity tests: the HIP path (through the C ABI, libmrt_hip.so) against the CPU oracle.

Bars (BASELINE.json north_star): mean radiance within 1e-4 per channel (L-inf) on identical
seeds; integer / byte outputs (tone-mapped and resampled images) bit-exact.
"""
import numpy as np
import pytest

from conftest import make_holder

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: <= 1e-4 per-channel L-inf on mean radiance vs the CPU reference

SCENES = {
    "default": lambda S: S.default_scene(res=(160, 90), sample=4),
    "cornell": lambda S: S.cornell_box(res=(96, 96), sample=8),
    "cornell_b16": lambda S: S.cornell_box(res=(64, 64), sample=8, bounce=16, floor_z=-0.201),
    "cornell2": lambda S: S.cornell_box2(res=(64, 64), ssaa=2, sample=4),
    "dof": lambda S: S.dof_scene(res=(128, 72), sample=4),
    "instance": lambda S: S.instance_grid(res=(96, 54), sample=2, n=5),
    "mesh": lambda S: S.mesh_scene(res=(96, 54), sample=2),
    "minecraft": lambda S: S.minecraft_like(res=(96, 54), ssaa=1, sample=2),
    "sink": lambda S: S.kitchen_sink(res=(96, 64), sample=8),
    "ragged": lambda S: S.cornell_box(res=(37, 23), ssaa=1.5, sample=3),
    "mesh1280": lambda S: S.mesh_scene(res=(64, 36), sample=2, n_tris=1280),   # 131 KB scene: 1024 threads, lane state in registers
    "instance_1728": lambda S: S.instance_grid(res=(64, 36), sample=2, n=12),    # ~150 KB scene with the instance BVH, same shape
    "bigmesh": lambda S: _big_mesh_scene(S),          # 20480 triangles: scene read through L2, triangle BVH route
    "mesh_glass_inst": lambda S: _glass_mesh_instances(S),   # t1 / i1 (exit hit) of meshes, rotated + translated instances
}


def _big_mesh_scene(S):
    d = S.mesh_scene(res=(64, 36), sample=2)
    d["scene"]["renderer"][0]["mesh"] = [[[float(c) for c in v] for v in t] for t in S.icosphere(5, 0.45, (1.3, 1.0, 1.1))]
    return d


def _glass_mesh_instances(S):
    d = S.mesh_scene(res=(80, 45), sample=4)
    m = d["scene"]["renderer"][0]
    m["mesh"] = [[[float(c) for c in v] for v in t] for t in S.icosphere(2, 0.3, (1.0, 1.2, 0.9))]
    m["mat"] = {"rough": 0.1, "glass": 0.4, "opacity": 0.2, "albedo": [0.8, 0.9, 1.0]}
    m.pop("pos", None)
    m["inst"] = [[[-0.5, 0.6, 0.0], [0, 0, -1, 0]], [[0.3, 0.5, 0.1], [0.4, 0.2, -1, 0.3]], [[0.1, 1.4, -0.1], [-0.7, 1, 0.2, 0]]]
    return d


def _gpu_render(render, spp, seed=5, **kw):
    from micro_raytracer_amd import Sampler
    s = Sampler(seed=seed, **kw)
    s.execute(render, n_samples=spp)
    return s


def test_device_math_contract_bit_exact(oracle_mod):
    """Every contract function gives the same bits on gfx950 as in the oracle."""
    from micro_raytracer_amd import _lib
    rng = np.random.default_rng(0)
    n = 200000
    cases = {
        0: (rng.uniform(0, 2 * np.pi, n).astype(np.float32), None),
        1: (rng.uniform(0, 2 * np.pi, n).astype(np.float32), None),
        2: (rng.uniform(-1, 1, n).astype(np.float32), None),
        3: (rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)),
        4: (rng.uniform(0, 4, n).astype(np.float32), rng.uniform(0.2, 2.5, n).astype(np.float32)),
        5: (rng.normal(size=n).astype(np.float32) * np.float32(1e3), None),
        6: (rng.uniform(0, 1e6, n).astype(np.float32), None),
        7: (rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)),
    }
    # edge values
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-42, -1e-42, 3.4e38, 1e-38, 2.0, 65536.0, 7e4], np.float32)
    for op, (a, b) in cases.items():
        a = np.concatenate([a, edge])
        if b is not None:
            b = np.concatenate([b, edge[::-1]])
        g = _lib.selftest_math(op, a, b)
        o = oracle_mod.math(op, a, b)
        same = (g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))
        assert same.all(), f"op {op}: {np.count_nonzero(~same)} mismatches, first a={a[~same][:4]} gpu={g[~same][:4]} cpu={o[~same][:4]}"


def test_minmax_zero_sign_and_total_key():
    """fmax/fmin on (+0,-0) and NaN, and the total_cmp key, match the contract."""
    from micro_raytracer_amd import _lib
    a = np.array([0.0, -0.0, np.nan, 1.0, -0.0, 0.0], np.float32)
    b = np.array([-0.0, 0.0, 2.0, np.nan, -0.0, 0.0], np.float32)
    mx = _lib.selftest_math(8, a, b)
    mn = _lib.selftest_math(9, a, b)
    assert list(np.signbit(mx)) == [False, False, False, False, True, False]
    assert list(np.signbit(mn)) == [True, True, False, False, True, False]
    assert mx[2] == 2.0 and mx[3] == 1.0 and mn[2] == 2.0 and mn[3] == 1.0
    t = np.array([-np.inf, -1.0, -0.0, 0.0, 1e-45, 1.0, np.inf, np.nan], np.float32)
    key = _lib.selftest_math(10, t).view(np.int32)
    assert (np.diff(key[:7].astype(np.int64)) > 0).all()
    assert key[7] == np.int32(-2**31)


@pytest.mark.parametrize("name", list(SCENES))
def test_scene_parity(name, oracle_mod):
    from micro_raytracer_amd import scenes
    render, holder = make_holder(SCENES[name](scenes))
    spp = render.rt.sample
    o = oracle_mod.Oracle(holder, seed=5)
    o.execute(spp)
    ref, cnt = o.accum()
    s = _gpu_render(render, spp)
    got, gcnt = s.accum()
    assert gcnt == cnt == spp
    assert (np.isnan(got) == np.isnan(ref)).all()
    err = np.nanmax(np.abs(got - ref)) / spp if np.isfinite(ref).any() else 0.0
    print(f"{name}: L-inf on mean radiance {err:.3e}, bit-identical pixels {np.mean(got == ref):.4f}")
    assert err <= TOL
    # Sampler::img: bytes must be identical when fed identical accumulators
    o.set_accum(got, gcnt)
    assert np.array_equal(s.img_ss(), o.img_ss())
    assert np.array_equal(s.img(), o.img())
    s.close()


def test_incremental_execute_equals_batched():
    """The accumulator is a pure function of (scene, seed, set of sample indices): batches aligned to the 16-sample
    accumulation chunks give identical bits however they are cut; unaligned cuts only re-associate f32 additions."""
    from micro_raytracer_amd import Sampler, scenes
    render, _ = make_holder(scenes.cornell_box(res=(64, 48), sample=48))
    a = _gpu_render(render, 48)
    b = Sampler(seed=5)
    for n in (16, 32):
        b.execute(render, n_samples=n)
    c = Sampler(seed=5)
    for n in (1, 3, 12, 20, 12):
        c.execute(render, n_samples=n)
    ra, ca = a.accum()
    rb, cb = b.accum()
    rc, cc = c.accum()
    assert ca == cb == cc == 48
    assert np.array_equal(ra, rb)
    assert np.abs(ra - rc).max() / 48 <= 1e-6


def test_sample_split_launch_is_bit_identical(monkeypatch):
    """Spreading a small frame over k lanes per pixel (chunk sums reduced in chunk order) changes no bit."""
    from micro_raytracer_amd import scenes
    render, _ = make_holder(scenes.cornell_box2(res=(96, 64), ssaa=1, sample=64))
    monkeypatch.setenv("MRT_K_SPLIT", "1")
    s1 = _gpu_render(render, 64)
    r1, _ = s1.accum()
    assert s1.stats()["k_split"] == 1
    for k in ("2", "4"):
        monkeypatch.setenv("MRT_K_SPLIT", k)
        sk = _gpu_render(render, 64)
        assert sk.stats()["k_split"] == int(k)
        assert np.array_equal(sk.accum()[0], r1)
    monkeypatch.delenv("MRT_K_SPLIT")
    sa = _gpu_render(render, 64)            # the automatic choice splits this small frame
    assert sa.stats()["k_split"] > 1
    assert np.array_equal(sa.accum()[0], r1)


def test_persistent_workgroups_change_no_bit(monkeypatch):
    """Multi-wavefront workgroups draw their tiles from a counter (csrc/mrt_kernels.hip); which wavefront renders a tile
    must not matter: the fixed blockIdx -> tile mapping (MRT_NO_PERSIST) gives the same accumulator bits."""
    from micro_raytracer_amd import scenes
    for desc in (scenes.mesh_scene(res=(200, 120), sample=32), scenes.minecraft_like(res=(96, 54), ssaa=2, sample=16)):
        render, _ = make_holder(desc)
        spp = render.rt.sample
        a = _gpu_render(render, spp)
        assert a.stats()["block_threads"] > 64
        ra, _ = a.accum()
        monkeypatch.setenv("MRT_NO_PERSIST", "1")
        b = _gpu_render(render, spp)
        rb, _ = b.accum()
        monkeypatch.delenv("MRT_NO_PERSIST")
        assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32))
        a.close(); b.close()


def test_every_launch_shape_gives_the_same_bits(monkeypatch):
    """The launch shape (64 / 256 / 512 / 1024 threads, scene in LDS or read through L2, persistent or not) is a pure
    scheduling choice: the same scene must give the same accumulator bits through every kernel instantiation."""
    from micro_raytracer_amd import scenes
    for desc in (scenes.cornell_box(res=(96, 64), sample=32), scenes.kitchen_sink(res=(80, 48), sample=16),
                 scenes.minecraft_like(res=(64, 40), ssaa=1, sample=16)):
        render, _ = make_holder(desc)
        spp = render.rt.sample
        ref = None
        seen = set()
        for threads, l2 in ((None, False), ("64", False), ("256", False), ("512", False), ("1024", False), (None, True)):
            if threads:
                monkeypatch.setenv("MRT_BLOCK_THREADS", threads)
            if l2:
                monkeypatch.setenv("MRT_SCENE_IN_L2", "1")
            s = _gpu_render(render, spp)
            got, _ = s.accum()
            st = s.stats()
            seen.add((st["block_threads"], st["lds_bytes"] > 0))
            s.close()
            monkeypatch.delenv("MRT_BLOCK_THREADS", raising=False)
            monkeypatch.delenv("MRT_SCENE_IN_L2", raising=False)
            if ref is None:
                ref = got
            assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (threads, l2)
        assert len(seen) >= 3          # the shapes really differed (big scenes refuse the small workgroups)


def test_every_staging_level_gives_the_same_bits(monkeypatch):
    """What a workgroup stages in LDS is a layout choice (csrc/mrt_api.cpp): the whole scene, the warm prefix (membership
    tables and texels read from global memory; the mesh kernels queue every leaf of a closest-hit walk), the deep level (the
    triangle-BVH table in level order, only its first nodes staged, triangles in global memory), nothing (all through L2).
    Every level must give the same accumulator bits, for every workgroup size it admits."""
    from micro_raytracer_amd import scenes
    for desc in (scenes.kitchen_sink(res=(80, 48), sample=16), scenes.mesh_scene(res=(96, 54), sample=8, n_tris=600),
                 scenes.minecraft_like(res=(64, 40), ssaa=1, sample=8)):
        render, _ = make_holder(desc)
        spp = render.rt.sample
        has_mesh = any(o.kind == "mesh" for o in render.scene.renderer)
        monkeypatch.setenv("MRT_COLD", "0")
        base = _gpu_render(render, spp)
        ref = base.accum()[0]
        assert base.stats()["kernel_features"] & 64 == 0
        monkeypatch.delenv("MRT_COLD")
        seen = set()
        levels = [{"MRT_COLD": "1"}] + ([{"MRT_DEEP_NODES": "3"}, {"MRT_DEEP_NODES": "40"}, {"MRT_DEEP_NODES": "100000"}] if has_mesh else [])
        for env in levels:
            for threads in (None, "256", "512", "1024"):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                if threads:
                    monkeypatch.setenv("MRT_BLOCK_THREADS", threads)
                s = _gpu_render(render, spp)
                got = s.accum()[0]
                st = s.stats()
                seen.add((st["kernel_features"] & 192, st["block_threads"]))
                s.close()
                for k in list(env) + ["MRT_BLOCK_THREADS"]:
                    monkeypatch.delenv(k, raising=False)
                assert st["kernel_features"] & 64, (env, threads, st)
                assert bool(st["kernel_features"] & 128) == ("MRT_DEEP_NODES" in env), (env, st)
                assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (env, threads)
        assert len(seen) >= (6 if has_mesh else 3), seen


def test_shards_reassemble_to_whole_frame():
    """Row shards (block-cyclic, 8-row blocks) of 3 contexts tile the single-context frame bit for bit."""
    from micro_raytracer_amd import scenes
    render, _ = make_holder(scenes.cornell_box2(res=(48, 52), ssaa=1, sample=3))
    whole, _ = _gpu_render(render, 3).accum()
    out = np.zeros_like(whole)
    seen = np.zeros(whole.shape[0], int)
    for r in range(3):
        s = _gpu_render(render, 3, shard_index=r, shard_count=3)
        loc, rows = s.accum_local()
        out[rows] = loc
        seen[rows] += 1
    assert (seen == 1).all()
    assert np.array_equal(out, whole)


def test_reset_and_set_accum_roundtrip():
    from micro_raytracer_amd import scenes
    render, _ = make_holder(scenes.default_scene(res=(64, 36), sample=2))
    s = _gpu_render(render, 2)
    a, c = s.accum()
    img = s.img()
    s.reset()
    z, c0 = s.accum()
    assert c0 == 0 and not z.any()
    s.set_accum(a, c)
    assert np.array_equal(s.img(), img)


def test_img_before_samples_is_an_error():
    from micro_raytracer_amd import MrtError, scenes, Sampler
    render, _ = make_holder(scenes.default_scene(res=(32, 18), sample=1))
    s = Sampler()
    s.execute(render, n_samples=0)
    with pytest.raises(MrtError):
        s.img()


def test_more_shards_than_row_blocks_and_context_churn():
    """Shards that own no rows are legal (a tiny frame on many GPUs); contexts can be created and destroyed freely."""
    from micro_raytracer_amd import Sampler, scenes
    render, _ = make_holder(scenes.cornell_box(res=(24, 8), sample=2))
    whole, _ = _gpu_render(render, 2).accum()
    out = np.zeros_like(whole)
    for r in range(4):
        s = _gpu_render(render, 2, shard_index=r, shard_count=4)
        loc, rows = s.accum_local()
        assert len(rows) == (8 if r == 0 else 0)
        out[rows] = loc
        assert s.accum()[1] == 2
        s.close()
    assert np.array_equal(out, whole)
    for _ in range(60):
        s = Sampler(seed=1)
        s.execute(render, n_samples=1)
        s.close()


def test_one_execute_is_cut_into_bounded_launches_without_changing_a_bit(monkeypatch):
    """mrt_execute bounds the sample-split buffer by cutting a large n_samples into launches of at most 64 chunks
    (1024 samples): 4096 samples in one call == 4 x 1024 == k_split 1, bit for bit, and the launch count shows the cut."""
    from micro_raytracer_amd import Sampler, scenes
    render, _ = make_holder(scenes.cornell_box(res=(256, 256), sample=4096, bounce=3))
    a = Sampler(seed=5)
    a.execute(render, n_samples=4096)
    sa = a.stats()
    assert sa["k_split"] > 1 and sa["launches"] == 4, sa
    ra, ca = a.accum()
    b = Sampler(seed=5)
    for _ in range(4):
        b.execute(render, n_samples=1024)
        assert b.stats()["launches"] == 1
    rb, cb = b.accum()
    assert ca == cb == 4096 and np.array_equal(ra.view(np.uint32), rb.view(np.uint32))
    a.close(); b.close()
    # a small cap (tests only) and an unaligned start: launches end on chunk boundaries of the GLOBAL sample index
    small, _ = make_holder(scenes.cornell_box2(res=(96, 64), ssaa=1, sample=200))
    monkeypatch.setenv("MRT_K_SPLIT", "1")
    ref = Sampler(seed=5)
    ref.execute(small, n_samples=5)
    ref.execute(small, n_samples=195)
    assert ref.stats()["launches"] == 1
    monkeypatch.setenv("MRT_K_SPLIT", "4")
    monkeypatch.setenv("MRT_MAX_CHUNKS", "4")
    c = Sampler(seed=5)
    c.execute(small, n_samples=5)
    c.execute(small, n_samples=195)          # chunks 0..12, four per launch -> 4 launches
    assert c.stats()["launches"] == 4 and c.stats()["k_split"] == 4
    assert np.array_equal(ref.accum()[0].view(np.uint32), c.accum()[0].view(np.uint32))


def test_sample_split_falls_back_to_one_lane_per_pixel_when_the_buffer_cannot_be_had(monkeypatch):
    """No room for the chunk planes (forced here through the test-only byte limit): the launch runs with k_split 1 and
    the same bits; a failed allocation must not poison the launch that follows (HIP 7 keeps the last real error)."""
    from micro_raytracer_amd import scenes
    render, _ = make_holder(scenes.cornell_box2(res=(96, 64), ssaa=1, sample=64))
    ref = _gpu_render(render, 64)
    assert ref.stats()["k_split"] > 1
    monkeypatch.setenv("MRT_PARTIAL_LIMIT_BYTES", "1000")
    s = _gpu_render(render, 64)
    assert s.stats()["k_split"] == 1
    assert np.array_equal(ref.accum()[0].view(np.uint32), s.accum()[0].view(np.uint32))
    monkeypatch.delenv("MRT_PARTIAL_LIMIT_BYTES")
    monkeypatch.setenv("MRT_PARTIAL_FAIL_ALLOC", "1")         # the hipMalloc route: an impossible size, tolerated
    f = _gpu_render(render, 64)
    assert f.stats()["k_split"] == 1
    assert np.array_equal(ref.accum()[0].view(np.uint32), f.accum()[0].view(np.uint32))
    f.execute(render, n_samples=16)                           # and the context keeps working
    assert f.accum()[1] == 80


def test_segment_counter_only_with_the_flag():
    """MRT_FLAG_COUNT_SEGMENTS (include/mrt.h): without it the per-call path has no counter reduction, no atomic and no
    blocking read-back, and mrt_stats.segments stays 0; with it the counter counts."""
    from micro_raytracer_amd import Sampler, _abi, scenes
    render, _ = make_holder(scenes.cornell_box(res=(64, 48), sample=4))
    plain = Sampler(seed=5)
    plain.execute(render, n_samples=4)
    st = plain.stats()
    assert st["segments"] == 0 and st["samples"] == 64 * 48 * 4 and st["kernel_ms"] > 0
    counted = Sampler(seed=5, flags=_abi.FLAG_COUNT_SEGMENTS)
    counted.execute(render, n_samples=4)
    sc = counted.stats()
    assert 64 * 48 * 4 <= sc["segments"] <= 64 * 48 * 4 * 9
    assert np.array_equal(plain.accum()[0], counted.accum()[0])


def test_per_call_loop_equals_one_batched_call():
    """The reference's callers run one Sampler::execute per sample (src/cli.rs:162-170, src/http.rs:141-144): 32 x
    mrt_execute(ctx, 1) accumulates the same samples as mrt_execute(ctx, 32); chunk sums are re-associated (<= 1e-6)."""
    from micro_raytracer_amd import Sampler, scenes
    render, _ = make_holder(scenes.cornell_box(res=(96, 64), sample=32))
    a = _gpu_render(render, 32)
    b = Sampler(seed=5)
    for _ in range(32):
        b.execute(render)
    ra, ca = a.accum()
    rb, cb = b.accum()
    assert ca == cb == 32
    assert np.abs(ra - rb).max() / 32 <= 1e-6
    assert np.array_equal(a.img(), b.img()) or np.abs(a.img().astype(int) - b.img().astype(int)).max() <= 1


def test_look_ahead_serves_the_per_call_loop_bit_for_bit():
    """The eager per-call loop (one mrt_execute per sample, src/cli.rs:162-170) is served from look-ahead launches from the
    third consecutive one-sample call on (csrc/mrt_api.cpp run_lookahead): the accumulator holds exactly the requested
    samples at EVERY return, bit for bit what the plain one-sample launches leave there; a batched call, a reset or a restored
    accumulator in the middle drops what was traced ahead and changes nothing."""
    from micro_raytracer_amd import Sampler, _abi, scenes
    for desc in (scenes.cornell_box(res=(96, 64), sample=45), scenes.kitchen_sink(res=(64, 40), sample=45)):
        render, _ = make_holder(desc)
        plain, ahead = Sampler(seed=5, flags=_abi.FLAG_NO_LOOKAHEAD), Sampler(seed=5)
        for i in range(45):
            plain.execute(render)
            ahead.execute(render)
            if i in (0, 1, 2, 3, 4, 7, 8, 20, 44):       # direct calls, first look-ahead sets, set boundaries, the 16-sample sets
                a, ca = plain.accum()
                b, cb = ahead.accum()
                assert ca == cb == i + 1 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
        assert ahead.stats()["samples"] == plain.stats()["samples"] and ahead.stats()["kernel_ms"] > 0
        assert np.array_equal(plain.img(), ahead.img())
        # a batched call in the middle: what was ahead is dropped, the samples are the same ones
        for s in (plain, ahead):
            s.execute(render, n_samples=19)
            for _ in range(5):
                s.execute(render)
        a, ca = plain.accum()
        b, cb = ahead.accum()
        assert ca == cb == 69 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # reset, then a restored accumulator: the loop picks up at the restored count
        keep, cnt = ahead.accum()
        for s in (plain, ahead):
            s.reset()
            for _ in range(4):
                s.execute(render)
            s.set_accum(keep, cnt)
            for _ in range(6):
                s.execute(render)
        a, ca = plain.accum()
        b, cb = ahead.accum()
        assert ca == cb == 75 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        plain.close(); ahead.close()


def test_look_ahead_on_a_row_shard_and_with_a_bound_accumulator():
    """Row shards run their own look-ahead; the accumulator may live in caller memory (mrt_bind_accum is what a rank of the
    multi-GPU path does): it still holds exactly the requested samples at every return."""
    from micro_raytracer_amd import Sampler, _abi, scenes
    render, _ = make_holder(scenes.cornell_box2(res=(80, 56), ssaa=1, sample=12))
    whole = Sampler(seed=6, flags=_abi.FLAG_NO_LOOKAHEAD)
    for _ in range(12):
        whole.execute(render)
    ref, _ = whole.accum()
    got = np.zeros_like(ref)
    for r in range(3):
        s = Sampler(seed=6, shard_index=r, shard_count=3)
        for _ in range(12):
            s.execute(render)
        loc, rows = s.accum_local()
        got[rows] = loc
        assert s.stats()["launches"] == 1
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))


def test_deferred_execution_books_calls_and_traces_them_batched():
    """MRT_FLAG_DEFER: the per-call loop of the reference's callers (src/cli.rs:162-170) only books samples; observing the
    accumulator traces them in one batch -- bit-identical to one batched call; with an observation after every call
    (--update, src/cli.rs:166-169) it equals the eager per-call loop; stats, reset and resume behave like eager."""
    from micro_raytracer_amd import Sampler, _abi, scenes
    render, _ = make_holder(scenes.cornell_box(res=(96, 64), sample=40))
    batched = _gpu_render(render, 40)
    want, _ = batched.accum()
    d = Sampler(seed=5, flags=_abi.FLAG_DEFER)
    for _ in range(40):
        secs = d.execute(render)
        assert secs < 0.05
    got, cnt = d.accum()                               # the observation: one launch of 40 samples
    assert cnt == 40 and np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert d.stats()["samples"] == 96 * 64 * 40 and d.stats()["launches"] == 1
    assert np.array_equal(d.img(), batched.img())
    # --update: an image after every sample
    eager, upd = Sampler(seed=5), Sampler(seed=5, flags=_abi.FLAG_DEFER)
    for _ in range(6):
        eager.execute(render)
        upd.execute(render)
        assert np.array_equal(eager.img(), upd.img())
    assert np.array_equal(eager.accum()[0].view(np.uint32), upd.accum()[0].view(np.uint32))
    # more booked samples than one batch: 1024 are traced as soon as they are booked, the rest at the observation
    small, _ = make_holder(scenes.default_scene(res=(32, 24), sample=1100))
    big = Sampler(seed=5, flags=_abi.FLAG_DEFER)
    for _ in range(1100):
        big.execute(small)
    ref = Sampler(seed=5)
    ref.execute(small, n_samples=1024)
    ref.execute(small, n_samples=76)
    assert np.array_equal(ref.accum()[0].view(np.uint32), big.accum()[0].view(np.uint32)) and big.accum()[1] == 1100
    # reset drops what is booked; set_accum settles first
    big.execute(small)
    big.reset()
    assert big.accum()[1] == 0 and not big.accum()[0].any()


def test_deferral_stays_off_once_the_device_pointer_was_handed_out(monkeypatch):
    """A caller that took the raw device pointer (mrt_accum_device_ptr) reads the sums behind the library's back: deferral
    must stay off for good, also after a bind / un-bind and after a bind that fails; stats.deferred tells which way a call
    went; MRT_DEFER=0 and an empty MRT_DEFER do not switch deferral on."""
    from micro_raytracer_amd import Sampler, _abi, scenes
    from micro_raytracer_amd._lib import MrtError
    render, _ = make_holder(scenes.cornell_box(res=(64, 48), sample=4))
    d = Sampler(seed=5, flags=_abi.FLAG_DEFER).create(render)
    d.execute(render)
    assert d.stats()["deferred"] == 1
    ptr, nbytes = d.accum_device_ptr()                   # settles what is booked, then marks the memory as visible
    with pytest.raises(MrtError):
        d.bind_accum(ptr, 16)                            # too small: refused, nothing changes
    d.bind_accum(0, 0)                                   # un-bind: back to library memory, the pointer is still out there
    d.execute(render, n_samples=3)
    assert d.stats()["deferred"] == 0
    # read the device memory directly, as a caller holding the pointer would: a second context copies it device-to-device
    reader = Sampler(seed=99).create(render)
    reader.set_accum_device(ptr, 4)
    seen = reader.accum()[0]
    eager = Sampler(seed=5)
    eager.execute(render)
    eager.execute(render, n_samples=3)                   # the same two launches, run at once
    assert np.array_equal(seen.view(np.uint32), eager.accum()[0].view(np.uint32))
    for val in ("0", ""):
        monkeypatch.setenv("MRT_DEFER", val)
        e = Sampler(seed=5)
        e.execute(render)
        assert e.stats()["deferred"] == 0
    monkeypatch.setenv("MRT_DEFER", "1")
    e = Sampler(seed=5)
    e.execute(render)
    assert e.stats()["deferred"] == 1


def test_no_event_timing_flag_leaves_kernel_ms_zero():
    from micro_raytracer_amd import Sampler, _abi, scenes
    render, _ = make_holder(scenes.cornell_box(res=(64, 48), sample=4))
    s = Sampler(seed=5, flags=_abi.FLAG_NO_EVENT_TIMING)
    s.execute(render, n_samples=4)
    st = s.stats()
    assert st["kernel_ms"] == 0 and st["samples"] == 64 * 48 * 4 and st["launches"] == 1
    assert np.array_equal(s.accum()[0], _gpu_render(render, 4).accum()[0])
