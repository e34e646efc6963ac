"""Edge-case render descriptions shared by the CPU (x86 kernel headers) and GPU parity tests."""
import numpy as np


def _base(res=(24, 16), ssaa=1, sample=2, bounce=3, **cam):
    c = {"pos": [0, -1.5, 0.2], "fov": 60, "gamma": 0.7, "exp": 0.5}
    c.update(cam)
    return {"rt": {"sample": sample, "bounce": bounce, "loss": 0.15}, "frame": {"res": list(res), "ssaa": ssaa, "cam": c},
            "scene": {"renderer": [], "sky": {"color": [0.2, 0.3, 0.5], "pwr": 0.6}}}


def cases():
    out = {}
    d = _base()
    d["scene"].pop("renderer")
    out["empty_scene_sky_only"] = d                                        # primary miss everywhere: raw sky colour

    d = _base(bounce=0)
    d["scene"]["renderer"] = [{"type": "sphere", "r": 0.5, "mat": {"albedo": "#80ff40"}}]
    d["scene"]["light"] = [{"type": "point", "pos": [-1, -2, 1]}]
    out["bounce_zero"] = d                                                 # exactly one hit per path

    d = _base(bounce=40, sample=1, res=(16, 10))
    d["scene"]["renderer"] = [{"type": "box", "sizes": [4, 4, 4], "mat": {"rough": 0.2}},                # camera inside a closed box
                              {"type": "sphere", "r": 0.3, "pos": [0, 0.5, 0], "mat": {"metal": 1}}]
    out["deep_bounces_inside_box"] = d

    d = _base(res=(1, 1), sample=5)
    d["scene"]["renderer"] = [{"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.5]}]
    out["one_pixel"] = d

    d = _base(res=(40, 30), ssaa=0.5, sample=2)                            # ssaa < 1: 20x15 supersampled, Lanczos upscale
    d["scene"]["renderer"] = [{"type": "sphere", "r": 0.6}, {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.6], "mat": {"rough": 1}}]
    d["scene"]["light"] = [{"type": "dir", "dir": [0.2, 1, -1], "pwr": 0.8, "color": "#fff0e0"}]
    out["ssaa_below_one_upscale"] = d

    d = _base(res=(30, 20), ssaa=1.3, sample=2)                            # fractional ssaa: 39 x 26 (truncated) frame
    d["scene"]["renderer"] = [{"type": "triangle", "vtx": [[0.7, 0.2, -0.4], [0.0, 0.3, 0.7], [-0.7, 0.1, -0.4]], "mat": {"albedo": "#ff8040", "rough": 0.4}},
                              {"type": "triangle", "vtx": [[0.7, 0.6, -0.4], [-0.7, 0.6, -0.4], [0.0, 0.6, 0.7]], "mat": {"emit": 0.5}}]
    out["lone_triangles_two_sided"] = d

    d = _base(sample=3)
    d["scene"]["renderer"] = [{"type": "sphere", "r": 0.4, "mat": {"opacity": 0.0, "glass": 1.5}},      # strong refraction incl. total internal reflection
                              {"type": "sphere", "r": 0.2, "pos": [0, 0, 0], "mat": {"emit": 1, "albedo": "#ffd080"}},   # light inside the glass ball
                              {"type": "plane", "n": [0, 1, 0.2], "pos": [0, 1.5, 0], "dir": [0.3, 0, -1, 0.4], "mat": {"rough": 1}}]   # rotated plane
    out["glass_around_emitter_rotated_plane"] = d

    d = _base(sample=2)
    lights = [{"type": "point", "pos": [np.cos(a) * 2, -1 + np.sin(a), 1.0], "pwr": 0.1, "color": [1, 0.5 + 0.5 * np.cos(a), 0.7]} for a in np.linspace(0, 6, 9)]
    d["scene"]["renderer"] = [{"type": "box", "sizes": [0.6, 0.6, 0.6], "dir": [0.1, 0.4, -1, 0.2], "mat": {"rough": 0.5}},
                              {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.3], "mat": {"rough": 1}}]
    d["scene"]["light"] = lights
    out["nine_lights"] = d

    d = _base(sample=2, aprt=0.0)                                          # pinhole: the centre row has dir.z == -0 before the transform
    d["frame"]["cam"]["pos"] = [0, -1.5, 0.0]
    d["scene"]["renderer"] = [{"type": "plane", "n": [0, 0, 1], "pos": [0, 0, 0.0], "mat": {"rough": 1}},          # edge-on plane through the camera
                              {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.5], "mat": {"albedo": "#4080ff"}},
                              {"type": "box", "sizes": [0.5, 0.5, 0.5], "pos": [0, 0.5, 0]}]
    out["pinhole_signed_zero_rays"] = d

    d = _base(sample=2)
    d["scene"]["renderer"] = [{"type": "mesh", "mesh": [[[0.5, 0.2, -0.3], [0.0, 0.3, 0.6], [-0.5, 0.1, -0.3]]], "mat": {"rough": 0.3}},   # 1-triangle mesh (octree of one leaf)
                              {"type": "sphere", "r": 0.25, "inst": []}]                                                                     # renderer without instances
    out["tiny_mesh_and_instanceless_renderer"] = d

    d = _base(sample=2)
    d["scene"]["renderer"] = [{"type": "sphere", "r": 0.5, "mat": {"albedo": [1.7, 0.2, -0.3], "opacity": -2.0, "rough": 2.5, "metal": 0.5, "loss": 0}}]   # out-of-gamut but legal values
    d["rt"]["loss"] = 3.0                                                   # loss.min(1.0) -> every bounce has pwr 0
    out["out_of_range_but_legal_material"] = d
    # ---- many-instance scenes: the instance BVH must return the linear scan's hit (first minimum in instance order) ----
    from micro_raytracer_amd import scenes
    out["bvh_instance_grid_1000"] = scenes.instance_grid(res=(64, 36), sample=2, n=10)
    out["bvh_minecraft_85"] = scenes.minecraft_like(res=(48, 27), ssaa=1, sample=2)
    rng = np.random.default_rng(3)
    boxes = [[[float(x) for x in rng.uniform(-3, 3, 3)], [float(x) for x in rng.normal(size=4)]] for _ in range(40)]
    spheres = [[[float(x) for x in rng.uniform(-3, 3, 3)], [0, 0, -1, 0]] for _ in range(40)]
    dup = [[[0.4, 1.0, 0.2], [0, 0, -1, 0]]] * 3                       # coincident instances: exact ties, the first one must win
    d = scenes.kitchen_sink(res=(48, 32), sample=2)
    d["scene"]["renderer"] += [{"type": "box", "sizes": [0.4, 0.3, 0.5], "mat": {"rough": 0.6, "albedo": "#a0ffa0"}, "inst": boxes},
                               {"type": "sphere", "r": 0.25, "mat": {"metal": 0.8}, "inst": spheres},
                               {"type": "sphere", "r": 0.3, "mat": {"albedo": "#ff2020"}, "inst": dup},
                               {"type": "sphere", "r": 0.3, "mat": {"albedo": "#2020ff"}, "inst": dup}]
    out["bvh_mixed_rotated_and_coincident"] = d

    # ---- the sphere test's cancellation against the instance BVH's margin: 96 spheres (64 of them emissive) of radius 0.004 .. 0.02, 30 to 900 units
    # away, seen through a 1.6 degree lens: Sphere::intersect's discriminant (b*b - 4ac, ~1e6 in f32) answers "hit" for rays that pass
    # up to ~1e-3 x distance OUTSIDE such a sphere -- most "hits" of this frame are of that kind, and a BVH margin sized for exact
    # geometry would lose them
    d = _base(res=(96, 64), sample=2, bounce=2, fov=1.6, aprt=0.0)
    d["frame"]["cam"]["pos"] = [0, 0, 0]
    rng = np.random.default_rng(4242)
    far_inst = []
    for k in range(96):
        y = float(np.exp(rng.uniform(np.log(30.0), np.log(900.0))))
        far_inst.append([[float(rng.uniform(-0.012, 0.012) * y), y, float(rng.uniform(-0.008, 0.008) * y)], [0, 0, -1, 0]])
    d["scene"]["renderer"] = [{"type": "sphere", "r": 0.004, "mat": {"emit": 1.0, "albedo": "#ffd040"}, "inst": far_inst[:32]},
                              {"type": "sphere", "r": 0.02, "mat": {"emit": 1.0, "albedo": "#40d0ff"}, "inst": far_inst[32:64]},
                              {"type": "sphere", "r": 0.01, "mat": {"rough": 0.5, "albedo": "#ff6060"}, "inst": far_inst[64:]}]
    d["scene"]["light"] = [{"type": "dir", "dir": [0.2, 1, -0.5], "pwr": 0.8}]
    out["bvh_far_tiny_spheres"] = d

    # ---- meshes: triangle-BVH route next to a mesh that cannot be bounded (reference walk), shared between instances ----
    ico = [[[float(c) for c in v] for v in t] for t in scenes.icosphere(1, 0.35, (1.0, 1.2, 0.8))]
    far = [[[float(c) for c in v] for v in t] for t in scenes.icosphere(0, 0.3)]
    far[3][1][0] = 2.5e6                                                     # one vertex beyond the culling range: no TBVH for this mesh
    dupm = [[[-0.6, 0.9, 0.1], [0, 0, -1, 0]]] * 2                         # coincident mesh instances: exact ties between equal triangles
    d = _base(res=(40, 24), sample=3, bounce=4)
    d["scene"]["renderer"] = [
        {"type": "mesh", "mesh": ico, "mat": {"rough": 0.3, "albedo": "#ffb060"}, "inst": [[[0.5, 0.6, 0.0], [0, 0, -1, 0]], [[-0.1, 1.3, 0.3], [0.5, 0.3, -1, 0.2]]]},
        {"type": "mesh", "mesh": far, "pos": [-0.7, 0.2, -0.1], "mat": {"glass": 0.6, "opacity": 0.3}},
        {"type": "mesh", "mesh": ico, "mat": {"albedo": "#6080ff", "metal": 0.7}, "inst": dupm},
        {"type": "mesh", "mesh": [], "pos": [0, 0, 0]},                                                # empty mesh: never hit
        {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.5], "mat": {"rough": 1}}]
    d["scene"]["light"] = [{"type": "point", "pos": [1.0, -1.5, 1.5], "pwr": 0.6}]
    out["meshes_tbvh_unbounded_empty_coincident"] = d

    # ---- NaN rays against a mesh: the pinhole camera sits IN the edge-on plane, so every ray of the centre row lies in it: Plane::intersect
    # answers 0 / 0 = NaN, which passes `t <= 0` (src/rt.rs:409): a hit at NaN; the next rays (shadow, bounce) leave from a NaN point and fail
    # every comparison of the mesh's octree walk -- the kernel answers those from the two ends of the leaf lists (mrt_trace.h mesh_isect)
    d = _base(res=(32, 20), sample=3, bounce=5, aprt=0.0)
    d["frame"]["cam"]["pos"] = [0, -1.5, 0.0]
    d["scene"]["renderer"] = [
        {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, 0.0], "mat": {"rough": 1}},
        {"type": "mesh", "mesh": ico, "mat": {"rough": 0.4, "albedo": "#ffb060"}, "inst": [[[0.3, 0.8, 0.1], [0, 0, -1, 0]], [[-0.5, 1.1, 0.2], [0.5, 0.3, -1, 0.2]]]},
        {"type": "plane", "n": [0, 0, 1], "pos": [0, 0, -0.5], "mat": {"albedo": "#4080ff"}}]
    d["scene"]["light"] = [{"type": "point", "pos": [1.0, -1.5, 1.5], "pwr": 0.6}]
    out["meshes_tbvh_nan_origins"] = d
    return out
