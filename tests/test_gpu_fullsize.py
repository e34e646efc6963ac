"""Properties at BASELINE.json's full sizes, where the oracle is too slow for a whole-frame comparison:
determinism, independence of batching and of the shard count, and oracle parity on a band of rows."""
import numpy as np
import pytest

from conftest import make_holder

pytestmark = pytest.mark.gpu


def _render(render, spp, seed=9, chunks=None, **kw):
    from micro_raytracer_amd import Sampler
    s = Sampler(seed=seed, **kw)
    for n in (chunks or [spp]):
        s.execute(render, n_samples=n)
    return s


def test_c2_cornell_512x512x64_determinism_batching_shards_and_band_parity(oracle_mod):
    """BASELINE.json configs[1]: CornellBox 512x512, 64 spp, 8 bounces."""
    from micro_raytracer_amd import scenes
    render, holder = make_holder(scenes.cornell_box(res=(512, 512), sample=64, bounce=8))
    a, ca = _render(render, 64).accum()
    b, cb = _render(render, 64).accum()
    assert ca == cb == 64 and np.array_equal(a, b)                         # deterministic
    c, _ = _render(render, 64, chunks=[16, 16, 32]).accum()
    assert np.array_equal(a, c)                                            # independent of (chunk-aligned) batching
    c, _ = _render(render, 64, chunks=[1, 7, 24, 32]).accum()
    assert np.abs(a - c).max() / 64 <= 1e-6                                # unaligned: f32 re-association only
    whole = np.zeros_like(a)
    for r in range(4):                                                     # 4 row shards == 1 context
        loc, rows = _render(render, 64, shard_index=r, shard_count=4).accum_local()
        whole[rows] = loc
    assert np.array_equal(a, whole)
    assert np.isfinite(a).all() and a.min() >= 0.0
    mean = a / 64
    assert 0.005 < mean.mean() < 0.2                                       # closed box lit by one emissive sphere
    o = oracle_mod.Oracle(holder, seed=9)                                  # oracle on an 8-row band through the spheres
    o.execute(64, rows=(296, 304))
    ref, _ = o.accum()
    err = np.abs(a[296:304] - ref[296:304]).max() / 64
    print(f"C2 band L-inf {err:.3e}")
    assert err <= 1e-4


def test_c3_cornellbox2_3840x2160_ssaa2_shards_and_band_parity(oracle_mod):
    """BASELINE.json configs[2]/[3] geometry at full supersampled size (1920x1080 ssaa 2 = 3840x2160), bounce 16,
    few samples: 8-way row shards reassemble the single-context frame; oracle parity on a band; img runs."""
    from micro_raytracer_amd import scenes
    render, holder = make_holder(scenes.cornell_box2(res=(1920, 1080), ssaa=2, sample=2, bounce=16))
    s = _render(render, 2)
    a, _ = s.accum()
    assert a.shape == (2160, 3840, 3)
    whole = np.zeros_like(a)
    for r in range(8):
        loc, rows = _render(render, 2, shard_index=r, shard_count=8).accum_local()
        whole[rows] = loc
    assert np.array_equal(a, whole)
    o = oracle_mod.Oracle(holder, seed=9)
    o.execute(2, rows=(1400, 1404))
    ref, _ = o.accum()
    err = np.abs(a[1400:1404] - ref[1400:1404]).max() / 2
    print(f"C3 band L-inf {err:.3e}")
    assert err <= 1e-4
    img = s.img()
    assert img.shape == (1080, 1920, 3) and img.dtype == np.uint8
    # resize of the GPU image == oracle's Lanczos3 of the GPU's tone-mapped frame (bytes)
    assert np.array_equal(img, oracle_mod.lanczos3_resize(s.img_ss(), 1920, 1080))


def test_c4_cornellbox2_3840x2160_ssaa1_shards_and_band_parity(oracle_mod):
    """BASELINE.json configs[3] as described: CornellBox2 at res 3840x2160, ssaa 1 (no resize in img), bounce 16,
    row-sharded 8 ways (block-cyclic, the partition the 8-GPU run uses): shards reassemble the single-context frame
    bit for bit, oracle parity on a band, and img == the tone-mapped frame (image 0.24 copies when sizes match)."""
    from micro_raytracer_amd import scenes
    render, holder = make_holder(scenes.cornell_box2(res=(3840, 2160), ssaa=1, sample=2, bounce=16))
    s = _render(render, 2)
    a, _ = s.accum()
    assert a.shape == (2160, 3840, 3)
    whole = np.zeros_like(a)
    for r in range(8):
        loc, rows = _render(render, 2, shard_index=r, shard_count=8).accum_local()
        whole[rows] = loc
    assert np.array_equal(a, whole)
    o = oracle_mod.Oracle(holder, seed=9)
    o.execute(2, rows=(1080, 1084))
    ref, _ = o.accum()
    err = np.abs(a[1080:1084] - ref[1080:1084]).max() / 2
    print(f"C4 band L-inf {err:.3e}")
    assert err <= 1e-4
    img = s.img()
    assert img.shape == (2160, 3840, 3) and np.array_equal(img, s.img_ss())
    o.set_accum(a, 2)
    assert np.array_equal(img[1080:1084], o.img()[1080:1084])


def test_c5_mesh_and_minecraft_shapes_band_parity(oracle_mod):
    """BASELINE.json configs[4]: triangle-heavy / textured scenes at 1920x1080 (1 spp), parity on a band."""
    from micro_raytracer_amd import scenes
    for name, desc, band in (("mesh", scenes.mesh_scene(res=(1920, 1080), sample=1), (500, 504)),
                             ("minecraft", scenes.minecraft_like(res=(1920, 1080), ssaa=2, sample=1), (1300, 1302))):
        render, holder = make_holder(desc)
        a, _ = _render(render, 1).accum()
        o = oracle_mod.Oracle(holder, seed=9)
        o.execute(1, rows=band)
        ref, _ = o.accum()
        err = np.nanmax(np.abs(a[band[0]:band[1]] - ref[band[0]:band[1]]))
        print(f"C5 {name} band L-inf {err:.3e}")
        assert err <= 1e-4


def test_c1_default_256x256_1spp_bounce1_whole_frame_parity(oracle_mod):
    """BASELINE.json configs[0] literally: example/Default.json geometry, 256x256, sample = 1, bounce = 1 -- the whole
    frame against the oracle (65 536 paths), image bytes included."""
    from micro_raytracer_amd import scenes
    render, holder = make_holder(scenes.default_scene(res=(256, 256), ssaa=1, sample=1, bounce=1))
    s = _render(render, 1)
    a, cnt = s.accum()
    assert cnt == 1 and a.shape == (256, 256, 3)
    o = oracle_mod.Oracle(holder, seed=9)
    o.execute(1)
    ref, _ = o.accum()
    err = np.abs(a - ref).max()
    print(f"C1 L-inf {err:.3e}")
    assert err <= 1e-4
    o.set_accum(a, 1)
    assert np.array_equal(s.img(), o.img())


def test_headline_launch_1080p_1024spp_matches_the_oracle_at_its_own_size(oracle_mod):
    """The configuration bench.py reports (`cornell_1080p_1024spp_b8`: Cornell box, 1920x1080, 1024 spp, 8 bounces) in ONE
    mrt_execute -- the very launch the headline number comes from: 256-thread persistent workgroups, 8 lanes per pixel
    (k_split 8: 64 chunk planes + reduce_chunks) -- against the oracle's 1024 sequential sample passes
    (reference src/cli.rs:162-170, fold src/rt.rs:956-994) on an 8-row band through the spheres."""
    from micro_raytracer_amd import scenes
    render, holder = make_holder(scenes.cornell_box(res=(1920, 1080), sample=1024, bounce=8))
    s = _render(render, 1024, seed=1)
    st = s.stats()
    assert st["k_split"] == 8 and st["block_threads"] == 256 and st["launches"] == 1, st
    assert st["kernel_features"] == 256 and st["scene_in_lds"] == 1                   # planes and spheres, every instance untransformed (F_IDENT)
    a, cnt = s.accum()
    assert cnt == 1024 and a.shape == (1080, 1920, 3)
    band = (640, 648)
    o = oracle_mod.Oracle(holder, seed=1)
    o.execute(1024, rows=band)
    ref, _ = o.accum()
    err = np.abs(a[band[0]:band[1]] - ref[band[0]:band[1]]).max() / 1024
    print(f"headline band L-inf {err:.3e} (kernel {st['kernel_ms']:.1f} ms)")
    assert err <= 1e-4
    o.set_accum(a, 1024)
    assert np.array_equal(s.img()[band[0]:band[1]], o.img()[band[0]:band[1]])


def test_c5_full_width_bands_with_the_sample_split(oracle_mod):
    """BASELINE.json configs[4] at full width with more than one sample chunk per launch (48 spp = three chunks), in the launch
    shapes the benchmarked 512-spp renders use: the 1080p mesh frame with the sample split (k_split > 1: chunk planes +
    reduce_chunks), the Minecraft-shaped frame at ssaa 2 (129 600 wave tiles: not split, chunk sums added in place, three
    chunks per lane).  2-row bands against the oracle."""
    from micro_raytracer_amd import scenes
    for name, desc, band, split in (("mesh", scenes.mesh_scene(res=(1920, 1080), sample=48), (520, 522), True),
                                    ("minecraft", scenes.minecraft_like(res=(1920, 1080), ssaa=2, sample=48), (1300, 1302), False)):
        render, holder = make_holder(desc)
        s = _render(render, 48)
        st = s.stats()
        assert (st["k_split"] > 1) == split, st
        a, _ = s.accum()
        o = oracle_mod.Oracle(holder, seed=9)
        o.execute(48, rows=band)
        ref, _ = o.accum()
        err = np.nanmax(np.abs(a[band[0]:band[1]] - ref[band[0]:band[1]])) / 48
        print(f"C5 {name} 48 spp band L-inf {err:.3e}, k_split {st['k_split']}, block {st['block_threads']}, features {st['kernel_features']}")
        assert err <= 1e-4
