"""Host-side logic without a GPU: the C-ABI library loads and exports its whole header, scene
validation mirrors the reference's panics, the JSON loader applies the reference's defaults, the packed
scene / octree / Lanczos taps agree with the oracle, and the kernel's per-lane code (compiled for x86,
tests/emu) reproduces the oracle."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, make_holder


def test_library_exports_every_declared_symbol():
    from micro_raytracer_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "mrt.h")).read()
    declared = set(re.findall(r"\b(mrt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.mrt_abi_version() == 3


def test_no_device_fails_loudly_not_silently():
    """Without a GPU mrt_create must fail with MRT_ERR_DEVICE: there is no CPU fallback."""
    from micro_raytracer_amd import MrtError, Sampler, _lib, load_render, scenes
    if _lib.lib().mrt_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(MrtError) as e:
        Sampler().execute(load_render(scenes.default_scene(res=(16, 16))))
    assert e.value.code == -3


def _create_err(desc):
    from micro_raytracer_amd import MrtError, Sampler, load_render
    with pytest.raises(MrtError) as e:
        Sampler().execute(load_render(desc))
    return e.value


def test_scene_validation_mirrors_reference_panics():
    """Everything the reference would panic on mid-render is rejected up front with MRT_ERR_SCENE (-2)."""
    from micro_raytracer_amd import scenes
    base = scenes.default_scene(res=(16, 16))
    d = json.loads(json.dumps(base)); d["scene"]["renderer"][0]["mat"] = {"emit": 1.5}
    assert _create_err(d).code == -2                                   # gen_bool(1.5) panics, src/rt.rs:968
    d = json.loads(json.dumps(base)); d["scene"]["renderer"][0]["mat"] = {"opacity": 1.2}
    assert _create_err(d).code == -2                                   # gen_bool(<0) panics, src/rt.rs:1054
    d = json.loads(json.dumps(base)); d["scene"]["renderer"] = [{"type": "triangle", "vtx": [[0, 0, 0], [1, 0, 0], [0, 0, 1]], "mat": {"tex": scenes.checker_texture(4, 4, 1)}}]
    assert _create_err(d).code == -2                                   # todo!() in Triangle::uv, src/rt.rs:546
    d = json.loads(json.dumps(base)); d["scene"]["renderer"][0]["mat"] = {"emap": {"w": 1, "h": 1, "dat": [[2.0, 0, 0]]}}
    assert _create_err(d).code == -2
    d = json.loads(json.dumps(base)); d["frame"]["ssaa"] = 0.01
    assert _create_err(d).code == -2                                   # 16 * 0.01 truncates to a 0-pixel frame
    d = json.loads(json.dumps(base)); d["scene"]["renderer"][0]["mat"] = {"opacity": -3.0, "emit": 1.0}
    assert _create_err(d).code == -3                                   # legal in the reference -> reaches the device check


def test_oracle_rejects_the_same_scenes(oracle_mod):
    from micro_raytracer_amd import scenes
    d = scenes.default_scene(res=(16, 16)); d["scene"]["renderer"][0]["mat"] = {"emit": -0.1}
    _, h = make_holder(d)
    with pytest.raises(ValueError):
        oracle_mod.Oracle(h)


def test_loader_defaults_match_reference_dump():
    """README.md:405 (`raytrace -v -d --obj sphere --light point: -0.5 -1 0.5`) is the reference's own dump of every default."""
    from micro_raytracer_amd import load_render
    from micro_raytracer_amd.scene import dump_render
    r = load_render({"scene": {"renderer": [{"type": "sphere", "r": 0.5}], "light": [{"type": "point", "pos": [-0.5, -1, 0.5]}]}})
    d = dump_render(r)
    f32 = lambda v: float(np.float32(v))
    assert (d["rt"]["bounce"], d["rt"]["sample"], f32(d["rt"]["loss"])) == (8, 16, f32(0.15))
    assert d["frame"]["res"] == [1280, 720] and d["frame"]["ssaa"] == 1.0
    cam = d["frame"]["cam"]
    assert cam["pos"] == [-0.0, -1.0, -0.0] and np.signbit(cam["pos"][0])
    assert cam["dir"] == [0.0, 0.0, 1.0, 0.0]
    assert [f32(cam[k]) for k in ("fov", "gamma", "exp", "aprt", "foc")] == [70.0, f32(0.8), f32(0.2), f32(0.001), 100.0]
    m = d["scene"]["renderer"][0]["mat"]
    assert m["albedo"] == [1.0, 1.0, 1.0] and (m["rough"], m["metal"], m["glass"], m["opacity"], m["emit"]) == (0, 0, 0, 1, 0)
    assert all(m[k] is None for k in ("tex", "rmap", "mmap", "gmap", "omap", "emap"))
    assert d["scene"]["light"][0] == {"type": "point", "pos": [-0.5, -1.0, 0.5], "pwr": 0.5, "color": [1.0, 1.0, 1.0]}
    assert d["scene"]["sky"] == {"color": [0.0, 0.0, 0.0], "pwr": 0.5}
    # default instance: pos 0, dir = Vec4f::backward() (src/parser.rs:847-851)
    inst = d["scene"]["renderer"][0]["inst"]
    assert inst == [[[0.0, 0.0, 0.0], [-0.0, -0.0, -1.0, -0.0]]]


def test_loader_hex_colors_inline_assets_and_instances():
    from micro_raytracer_amd import load_render, scenes
    from micro_raytracer_amd.scene import Texture, mesh_to_inline, parse_color
    assert np.array_equal(parse_color("#ffc177"), np.array([255, 193, 119], np.float32) / np.float32(255))
    t = Texture(2, 1, np.array([[0.5, 0.25, 1.0], [0, 0, 0]], np.float32))
    tris = scenes.icosphere(0, 0.3)
    d = {"scene": {"renderer": [
        {"type": "plane", "n": [0, 0, 1], "mat": {"tex": t.to_inline()}},
        {"type": "mesh", "mesh": mesh_to_inline(tris), "pos": [1, 2, 3], "inst": [[[0, 0, 1], [0, 0, -1, 0]]]},
    ]}}
    r = load_render(d)
    assert r.scene.renderer[0].mat.tex.w == 2 and np.array_equal(r.scene.renderer[0].mat.tex.dat, t.dat)
    assert np.array_equal(r.scene.renderer[1].mesh, tris)
    # pos given together with inst: (pos, backward) is inserted first (src/parser.rs:841-844)
    inst = r.scene.renderer[1].inst
    assert len(inst) == 2 and list(inst[0][0]) == [1, 2, 3] and list(inst[0][1]) == [-0.0, -0.0, -1.0, -0.0] and list(inst[1][0]) == [0, 0, 1]


def test_packer_octree_equals_oracle_octree(oracle_mod, emu_mod):
    from micro_raytracer_amd import scenes
    for tris in (scenes.bumpy_mesh(967), scenes.icosphere(1, 0.22, (1.0, 1.0, 1.3)), scenes.icosphere(0, 1.0)):
        d = {"frame": {"res": [8, 8]}, "scene": {"renderer": [{"type": "mesh", "mesh": [[[float(c) for c in v] for v in t] for t in tris]}]}}
        _, h = make_holder(d)
        ob, oc, oi = oracle_mod.Oracle(h).mesh_octree(0)
        eb, ec, ei = emu_mod.octree(tris)
        assert np.array_equal(ob.view(np.uint32), eb.view(np.uint32))
        assert np.array_equal(oc, ec) and np.array_equal(oi, ei)
        assert len(oc) <= 512 and oc.sum() == len(oi)


def test_instance_bvh_is_built_only_for_many_instances(emu_mod):
    import ctypes as C
    from micro_raytracer_amd import scenes
    L = emu_mod.lib()
    for desc, want in ((scenes.cornell_box(res=(8, 8)), False), (scenes.instance_grid(res=(8, 8), n=10), True),
                       (scenes.minecraft_like(res=(8, 8), ssaa=1), True), (scenes.instance_grid(res=(8, 8), n=2), False)):
        _, h = make_holder(desc)
        info = (C.c_uint32 * 8)()
        assert L.emu_pack(C.cast(h.ptr(), C.c_void_p), None, None, None, info) == 0
        feats = L.emu_features(C.cast(h.ptr(), C.c_void_p))
        assert bool(feats & 16) == want


def test_packer_texture_formats_and_sizes(emu_mod):
    from micro_raytracer_amd import scenes
    _, h = make_holder(scenes.minecraft_like(res=(16, 16), ssaa=1))
    info = emu_mod.pack(h)
    assert info["n_tex_u8"] == 11 and info["n_tex_f32"] == 0       # k/255 texels are stored as RGB8 + LUT
    assert info["n_inst"] == 85 and info["n_rend"] == 9
    assert info["blob_words"] * 4 < 160 * 1024                     # fits the LDS of one CU
    d = scenes.default_scene(res=(16, 16)); d["scene"]["renderer"][0]["mat"] = {"tex": {"w": 1, "h": 1, "dat": [[0.3, 0.3, 0.3]]}}
    _, h = make_holder(d)
    assert emu_mod.pack(h)["n_tex_f32"] == 1                       # 0.3 is not k/255: kept as f32


SCENES = {
    "default": lambda S: S.default_scene(res=(64, 36), sample=3),
    "cornell": lambda S: S.cornell_box(res=(48, 48), sample=4),
    "cornell2": lambda S: S.cornell_box2(res=(32, 32), ssaa=2, sample=3),
    "dof": lambda S: S.dof_scene(res=(64, 36), sample=3),
    "instance": lambda S: S.instance_grid(res=(48, 27), sample=2, n=4),
    "mesh": lambda S: S.mesh_scene(res=(48, 27), sample=2),
    "minecraft": lambda S: S.minecraft_like(res=(48, 27), ssaa=1, sample=2),
    "sink": lambda S: S.kitchen_sink(res=(64, 40), sample=6),
    "ragged": lambda S: S.cornell_box(res=(21, 13), ssaa=1.5, sample=2),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_kernel_lane_code_matches_oracle_on_x86(name, oracle_mod, emu_mod):
    """mrt_trace.h (the megakernel's per-lane body) compiled for x86 vs the oracle: <= 1e-4 on mean radiance,
    and the image bytes produced by the kernels' per-element bodies are identical."""
    from micro_raytracer_amd import scenes
    render, h = make_holder(SCENES[name](scenes))
    spp = render.rt.sample
    o = oracle_mod.Oracle(h, seed=5)
    o.execute(spp)
    ref, _ = o.accum()
    got, seg = emu_mod.render(h, 5, spp)
    assert (np.isnan(got) == np.isnan(ref)).all()
    assert np.nanmax(np.abs(got - ref)) / spp <= 1e-5
    assert 0 < seg <= o.segments          # the kernel stops a path at an emitting hit, the reference traces on
    o.set_accum(got, spp)
    ss, out = emu_mod.img(h, got, spp)
    assert np.array_equal(ss, o.img_ss()) and np.array_equal(out, o.img())


def test_oracle_is_independent_of_tiling_and_threads(oracle_mod):
    from micro_raytracer_amd import scenes
    _, h = make_holder(scenes.cornell_box(res=(40, 24), sample=2))
    a = oracle_mod.Oracle(h, seed=2); a.execute(2, threads=1, n_dim=64)
    b = oracle_mod.Oracle(h, seed=2); b.execute(1, threads=5, n_dim=3); b.execute(1, threads=2, n_dim=7)
    assert np.array_equal(a.accum()[0], b.accum()[0])
    c = oracle_mod.Oracle(h, seed=3); c.execute(2)
    assert not np.array_equal(a.accum()[0], c.accum()[0])


def test_lanczos_matches_reference_shape_properties(oracle_mod):
    """image 0.24 Lanczos3: identity when sizes match, constant images stay constant, taps normalised."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(12, 20, 3), dtype=np.uint8)
    assert np.array_equal(oracle_mod.lanczos3_resize(img, 20, 12), img)
    const = np.full((40, 60, 3), 137, np.uint8)
    assert (oracle_mod.lanczos3_resize(const, 30, 20) == 137).all()
    L = oracle_mod.lib()
    w = (C.c_float * 64)(); left = C.c_uint32()
    for o_idx in (0, 7, 29):
        n = L.orc_lanczos3_weights(60, 30, o_idx, C.byref(left), w, 64)
        assert 6 <= n <= 14 and abs(sum(w[i] for i in range(n)) - 1.0) < 1e-5


def test_tonemap_known_answers(oracle_mod):
    """(255 * t(c^gamma)) as u8 with t(v) = v (1 + v / (1-exp)^2) / (1 + v), src/sampler.rs:88-94."""
    for c, gamma, exp in ((0.25, 0.8, 0.2), (1.7, 0.5, 0.75), (0.0, 0.8, 0.2), (1e9, 0.6, 0.8)):
        got = oracle_mod.tonemap_px([c * 4, c * 4, c * 4], 4, gamma, exp)[0]
        g = float(c) ** gamma
        want = 255.0 * (g * (1 + g / (1 - exp) ** 2) / (1 + g))
        want = 0 if not want > 0 else min(255, int(want))
        assert abs(int(got) - want) <= 1
    # `as u8` saturates and maps NaN to 0; powf of a negative base is NaN
    assert oracle_mod.tonemap_px([np.nan, -1.0, 0.0], 1, 0.8, 0.2).tolist() == [0, 0, 0]


def test_image_writers_roundtrip(tmp_path):
    """mrt_save_image: the .ppm / .png files `img.save` would write (src/cli.rs:168,174) decode back to the same bytes."""
    from PIL import Image
    from micro_raytracer_amd import _lib
    rng = np.random.default_rng(5)
    for shape in ((7, 13, 3), (300, 257, 3)):                      # second one spans several 64 KB stored-deflate blocks
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        for ext in ("png", "ppm"):
            p = tmp_path / f"o{shape[0]}.{ext}"
            _lib.save_image(p, img)
            assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), img)
    with pytest.raises(_lib.MrtError):
        _lib.save_image(tmp_path / "o.jpg", img)


def test_mesh_tbvh_route_equals_the_octree_walk(emu_mod):
    """The triangle-BVH route of the mesh arm (mrt_trace.h) returns exactly what the reference's octree walk returns:
    hit / miss, entry and exit distance bits and triangle ids, any-hit answer -- on random, vertex-aimed (equal-t ties),
    axis-parallel and cell-boundary rays against meshes of several shapes (tests/mesh_probe.py runs the long version)."""
    import ctypes as C
    import mesh_probe
    from micro_raytracer_amd._abi import build_desc
    from micro_raytracer_amd.scene import load_render
    L = emu_mod.lib()
    L.emu_mesh_probe.restype = C.c_int
    ties = 0
    for seed in range(10):
        rng = np.random.default_rng(seed)
        tris = mesh_probe.random_mesh(rng, seed % 5)
        pos = [0.25, -0.5, 0.125] if seed % 2 else [0.0, 0.0, 0.0]
        desc = {"frame": {"res": [8, 8]}, "scene": {"renderer": [{"type": "mesh", "mesh": [[[float(c) for c in vv] for vv in t] for t in tris], "pos": pos}]}}
        h = build_desc(load_render(desc))
        o, d = mesh_probe.rays_for(rng, tris, 4000)
        # rays with a NaN direction (Vec3f::norm of a zero vector: a degenerate triangle's normal): the kernel answers them from
        # the first and last listed triangle instead of walking everything -- the probe checks that against the reference's walk
        d[:40] = np.nan
        d[40:50, 0] = np.nan                                     # (partly NaN: no shortcut)
        o[50:60] = np.nan
        o = np.ascontiguousarray(o + np.asarray(pos, np.float32))
        out = np.zeros((len(o), 10), np.uint32)
        stats = (C.c_uint32 * 2)()
        bad = L.emu_mesh_probe(C.cast(h.ptr(), C.c_void_p), len(o), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), stats)
        assert bad == 0 and stats[1] == 1 and stats[0] > 200
        ties += int(((out[:, 0] == 1) & (out[:, 1] == out[:, 3]) & (out[:, 2] != out[:, 4])).sum())
    assert ties > 50      # the first-minimum / last-maximum tie rules were exercised


def test_save_image_errors_carry_a_message(tmp_path):
    from micro_raytracer_amd import MrtError, _lib
    img = np.zeros((4, 5, 3), np.uint8)
    _lib.save_image(tmp_path / "ok.png", img)
    with pytest.raises(MrtError) as e:
        _lib.save_image(tmp_path / "x.jpg", img)
    assert "unsupported extension" in str(e.value) and "x.jpg" in str(e.value)
    with pytest.raises(MrtError) as e:
        _lib.save_image(tmp_path / "no_such_dir" / "x.png", img)
    assert "cannot open" in str(e.value) and "No such file" in str(e.value)


class _StubLib:
    """Just enough of libmrt_hip.so to watch which contexts the Python Sampler creates (no GPU here)."""

    def __init__(self):
        self.created, self.destroyed, self.set_accum_calls, self.binds = [], [], [], []
        self._next = 1000
        self._count = {}

    def mrt_create(self, desc, opts):
        self._next += 1
        self.created.append(self._next)
        self._count[self._next] = 0
        return self._next

    def mrt_destroy(self, ctx):
        self.destroyed.append(ctx)

    def mrt_dims(self, ctx, nw, nh, lr):
        nw._obj.value, nh._obj.value, lr._obj.value = 16, 9, 9
        return 0

    def mrt_execute(self, ctx, n, secs):
        self._count[ctx] += n
        return 0

    def mrt_accum(self, ctx, rgb, cnt):
        cnt._obj.value = self._count[ctx]
        return 0

    def mrt_set_accum(self, ctx, rgb, count):
        self.set_accum_calls.append((ctx, count))
        self._count[ctx] = count
        return 0

    def mrt_bind_accum(self, ctx, ptr, nbytes):
        self.binds.append((ctx, ptr.value, nbytes))
        return 0

    def mrt_last_status(self):
        return 0

    def mrt_last_error(self):
        return b""


def test_sampler_context_follows_the_description_not_its_address(monkeypatch):
    """Sampler::execute takes scene, frame and rt on every call (src/sampler.rs:28).  The Python mirror caches one
    device context; it must belong to the description passed NOW: two temporaries (CPython reuses the address of the
    first for the second) give two contexts, an in-place edit rebuilds, an unchanged description does not."""
    from micro_raytracer_amd import Sampler, _lib, load_render, scenes
    stub = _StubLib()
    monkeypatch.setattr(_lib, "lib", lambda: stub)
    s = Sampler()
    s.execute(load_render(scenes.default_scene(res=(16, 9))))
    s.execute(load_render(scenes.cornell_box(res=(16, 9))))          # a different temporary, very likely at the same address
    assert len(stub.created) == 2 and stub.destroyed == stub.created[:1]
    # what was accumulated is carried over when the frame size is unchanged (the reference adds into the same map)
    assert stub.set_accum_calls == [(stub.created[1], 1)]
    r = load_render(scenes.cornell_box(res=(16, 9)))
    s.execute(r)
    s.execute(r)
    s.execute(r)
    assert len(stub.created) == 3                                     # one context for the three passes
    r.frame.cam.pos[2] += 0.25                                        # in-place edits are picked up
    s.execute(r)
    assert len(stub.created) == 4
    r.rt.bounce = 3
    s.execute(r)
    assert len(stub.created) == 5
    r.scene.renderer[5].mat.glass = 0.5
    s.execute(r)
    assert len(stub.created) == 6
    s.execute(r.scene, r.frame, r.rt)                                 # the reference's three-argument form: same contents,
    s.execute(r.scene, r.frame, r.rt)                                 # one more context for the new Render wrapper at most
    assert len(stub.created) <= 7
    n = len(stub.created)
    s.execute(r.scene, r.frame, r.rt)
    assert len(stub.created) == n
    s.close()
    assert len(stub.destroyed) == len(stub.created)


def test_sampler_rebuild_keeps_bindings_and_refuses_to_desynchronise_a_shard(monkeypatch):
    """a rebuilt context must render into the caller's bound buffer again; a sharded context that already
    holds samples cannot be rebuilt behind the caller's back (its rows and count cannot be restored): that raises."""
    from micro_raytracer_amd import Sampler, _lib, load_render, scenes
    from micro_raytracer_amd._lib import MrtError
    stub = _StubLib()
    monkeypatch.setattr(_lib, "lib", lambda: stub)
    r = load_render(scenes.cornell_box(res=(16, 9)))
    s = Sampler().create(r)
    s.bind_accum(0xABC000, 16 * 9 * 12)
    s.execute(r)
    r.rt.bounce = 2                                                   # in-place edit -> rebuild
    s.execute(r)
    assert len(stub.created) == 2
    assert stub.binds == [(stub.created[0], 0xABC000, 16 * 9 * 12), (stub.created[1], 0xABC000, 16 * 9 * 12)]
    assert stub.set_accum_calls == [(stub.created[1], 1)]             # and the sums are carried into it
    s.close()
    # a shard with samples on board
    sh = Sampler(shard_index=1, shard_count=2).create(r)
    sh.bind_accum(0xDEF000, 8 * 16 * 12)
    r.rt.bounce = 4
    sh.execute(r)                                                     # nothing accumulated yet: rebuilt and re-bound
    assert stub.binds[-1] == (stub.created[-1], 0xDEF000, 8 * 16 * 12)
    n = len(stub.created)
    r.rt.bounce = 5
    with pytest.raises(MrtError) as e:
        sh.execute(r)
    assert "sharded" in str(e.value) and len(stub.created) == n
    sh.close()


def test_fingerprint_sees_bulk_edits_and_is_stable_for_lists():
    """Bulk arrays are fingerprinted by shape, dtype and a CRC of a strided sample: rewriting a mesh or a texture is seen,
    and a mesh assigned as a Python list (a fresh ndarray temporary on every call) does not look changed every time."""
    from micro_raytracer_amd import load_render, scenes
    from micro_raytracer_amd.sampler import _fingerprint
    r = load_render(scenes.mesh_scene(res=(16, 9), n_tris=200))
    f0 = _fingerprint(r)
    assert _fingerprint(r) == f0
    r.scene.renderer[0].mesh *= 1.01                                  # in-place rewrite of every vertex
    assert _fingerprint(r) != f0
    f1 = _fingerprint(r)
    r.scene.renderer[1].mat.tex.dat[::2] = 0.5                        # texels
    assert _fingerprint(r) != f1
    r.scene.renderer[0].mesh = r.scene.renderer[0].mesh.tolist()      # a list: converted and hashed whole, stable
    f2 = _fingerprint(r)
    assert _fingerprint(r) == f2
    r2 = load_render(scenes.instance_grid(res=(16, 9), n=4))
    g0 = _fingerprint(r2)
    r2.scene.renderer[0].inst[-1][0][0] += 1.0
    assert _fingerprint(r2) != g0


def test_level_ordered_triangle_bvh_walk_equals_the_depth_first_one(emu_mod):
    """F_DEEP (csrc/mrt_scene.h): the triangle-BVH table in level order with explicit child links, of which only a prefix is
    staged in LDS.  The lane code built that way must render the same bits as the depth-first form, whatever the prefix:
    1 node (just the root), a few levels, everything; one mesh, several mesh instances behind an instance BVH.  The same for
    the closest-hit walk that queues every leaf before testing any triangle (F_COLD): the candidate set does not depend on
    the order of the tests."""
    from micro_raytracer_amd import scenes
    import edge_cases
    cases = [scenes.mesh_scene(res=(48, 27), sample=2, n_tris=400), scenes.kitchen_sink(res=(48, 30), sample=3)]
    cases += [d for k, d in edge_cases.cases().items() if k.startswith("meshes_tbvh")]
    for desc in cases:
        render, h = make_holder(desc)
        spp = render.rt.sample
        ref, seg = emu_mod.render(h, 5, spp)
        n_nodes = emu_mod.layout(h)["n_tbvh_nodes"]
        n_mesh = sum(1 for o in render.scene.renderer if o.kind == "mesh")
        got, seg2 = emu_mod.render(h, 5, spp, deep_nodes=0xffffffff)          # F_COLD alone: queued walk, depth-first table
        assert seg2 == seg and np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(ref).view(np.uint32))
        for hot in sorted({max(1, n_mesh), 7, 64, n_nodes}):
            got, seg2 = emu_mod.render(h, 5, spp, deep_nodes=hot)
            assert seg2 == seg
            assert np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(ref).view(np.uint32)), hot


def test_launch_plan_staging_levels_and_shapes(monkeypatch):
    """The launch policy of csrc/mrt_api.cpp as data (mrt_plan_launch: host only): which part of the scene a workgroup stages
    in LDS and the workgroup size, for the scenes of the BASELINE configs and the meshes beyond the LDS."""
    from micro_raytracer_amd import _lib, load_render, scenes
    LDS = 160 * 1024
    plan = lambda d: _lib.plan_launch(load_render(d))
    for k in ("MRT_COLD", "MRT_DEEP_NODES", "MRT_SCENE_IN_L2", "MRT_BLOCK_THREADS"):
        monkeypatch.delenv(k, raising=False)
    # small scenes: the whole scene, four waves around one copy, 8 workgroups per CU; one-sample launches on the plain grid
    # (256 = F_IDENT: every instance untransformed; CornellBox2 has a rotated box)
    for d, feat in ((scenes.cornell_box(res=(1920, 1080)), 0 | 256), (scenes.cornell_box2(res=(1920, 1080)), 1), (scenes.default_scene(), 8 | 256)):
        p = plan(d)
        assert (p["staging"], p["block_threads"], p["kernel_features"], p["small_plain_grid"]) == ("all", 256, feat, 1), p
        assert p["staged_bytes"] == p["scene_bytes"] < 6 * 1024 and 8 * p["lds_bytes"] <= LDS
    # the 967-triangle mesh scene: warm (texels and octree leaf lists out), one 1024-thread workgroup with stash and walk areas
    p = plan(scenes.mesh_scene())
    assert (p["staging"], p["block_threads"], p["kernel_features"]) == ("warm", 1024, 15 | 64), p
    # triangles and membership tables staged; the leaf queue of the binary walk: 8 entries + what the LDS has left
    assert p["walk_cap"] == 9 and p["scene_bytes"] - p["staged_bytes"] < 13 * 1024        # (only the octree leaf lists stay out)
    assert p["staged_bytes"] < p["scene_bytes"] and p["lds_bytes"] == p["staged_bytes"] + 1024 * 4 * (7 + 9) <= LDS < p["lds_bytes"] + 4096
    assert p["tbvh_hot_nodes"] == p["tbvh_nodes"] > 1000            # binary nodes
    # the Minecraft-shaped scene: warm, 256-thread workgroups (6-wave kernel: six of them per CU), texels out of LDS
    p = plan(scenes.minecraft_like())
    assert (p["staging"], p["block_threads"], p["kernel_features"]) == ("warm", 256, 29 | 64), p
    assert p["scene_bytes"] - p["staged_bytes"] > 70 * 1024 and 6 * p["lds_bytes"] <= LDS < 7 * p["lds_bytes"]
    # meshes beyond the LDS: deep -- 4-wide triangle BVH in level order, as many top nodes as fit, triangles out
    for n in (5120, 20480):
        p = plan(scenes.mesh_scene(res=(64, 36), n_tris=n))
        assert (p["staging"], p["block_threads"], p["kernel_features"]) == ("deep", 1024, 15 | 64 | 128), p
        assert 300 < p["tbvh_hot_nodes"] < p["tbvh_nodes"] and LDS - 4096 < p["lds_bytes"] <= LDS and p["walk_cap"] == 16
    # 1000 instances, no texels: nothing to leave out, one copy for a 1024-thread workgroup
    p = plan(scenes.instance_grid())
    assert (p["staging"], p["block_threads"], p["kernel_features"]) == ("all", 1024, 8 | 16 | 256), p      # (256: all untransformed)
    # the knobs of the tests
    monkeypatch.setenv("MRT_SCENE_IN_L2", "1")
    p = plan(scenes.mesh_scene())
    assert (p["staging"], p["block_threads"], p["staged_bytes"], p["tbvh_hot_nodes"]) == ("none", 256, 0, 0), p
    monkeypatch.delenv("MRT_SCENE_IN_L2")
    monkeypatch.setenv("MRT_COLD", "0")
    p = plan(scenes.minecraft_like())
    assert (p["staging"], p["block_threads"], p["kernel_features"]) == ("all", 1024, 29), p
    p = plan(scenes.mesh_scene())
    assert (p["staging"], p["block_threads"], p["kernel_features"]) == ("all", 1024, 15), p
    monkeypatch.delenv("MRT_COLD")
    monkeypatch.setenv("MRT_DEEP_NODES", "5")
    p = plan(scenes.mesh_scene(res=(64, 36), n_tris=300))
    assert (p["staging"], p["tbvh_hot_nodes"], p["kernel_features"] & 192) == ("deep", 5, 192), p
    monkeypatch.delenv("MRT_DEEP_NODES")
    monkeypatch.setenv("MRT_BLOCK_THREADS", "64")
    p = plan(scenes.cornell_box())
    assert (p["block_threads"], p["small_plain_grid"]) == (64, 0), p
