/* layout.c — TEST INFRASTRUCTURE: sizeof / offsetof of every struct of include/mrt.h as this C compiler lays them out.
 * One line per struct ("struct <name> <size> <align>") and per field ("field <struct>.<name> <offset> <size>").
 * tests/test_shim_layout.py compares the output with shim/rust/layout.txt (committed) and with the layout the
 * #[repr(C)] declarations of shim/rust/sampler_hip.rs get under the same C rules. */
#include <stddef.h>
#include <stdio.h>
#include "../../include/mrt.h"

#define S(T) printf("struct %s %zu %zu\n", #T, sizeof(T), _Alignof(T))
#define F(T, f) printf("field %s.%s %zu %zu\n", #T, #f, offsetof(T, f), sizeof(((T *)0)->f))

int main(void)
{
    S(mrt_camera); F(mrt_camera, pos); F(mrt_camera, dir); F(mrt_camera, fov); F(mrt_camera, gamma); F(mrt_camera, exp); F(mrt_camera, aprt); F(mrt_camera, foc);
    S(mrt_frame); F(mrt_frame, res_w); F(mrt_frame, res_h); F(mrt_frame, ssaa); F(mrt_frame, cam);
    S(mrt_rt); F(mrt_rt, bounce); F(mrt_rt, sample); F(mrt_rt, loss);
    S(mrt_texture); F(mrt_texture, w); F(mrt_texture, h); F(mrt_texture, dat);
    S(mrt_material); F(mrt_material, albedo); F(mrt_material, rough); F(mrt_material, metal); F(mrt_material, glass); F(mrt_material, opacity); F(mrt_material, emit);
    F(mrt_material, tex); F(mrt_material, rmap); F(mrt_material, mmap); F(mrt_material, gmap); F(mrt_material, omap); F(mrt_material, emap);
    S(mrt_instance); F(mrt_instance, pos); F(mrt_instance, dir);
    S(mrt_renderer); F(mrt_renderer, kind); F(mrt_renderer, param); F(mrt_renderer, tris); F(mrt_renderer, n_tris); F(mrt_renderer, mat); F(mrt_renderer, inst); F(mrt_renderer, n_inst);
    S(mrt_light); F(mrt_light, kind); F(mrt_light, v); F(mrt_light, pwr); F(mrt_light, color);
    S(mrt_sky); F(mrt_sky, color); F(mrt_sky, pwr);
    S(mrt_scene); F(mrt_scene, renderer); F(mrt_scene, n_renderer); F(mrt_scene, light); F(mrt_scene, n_light); F(mrt_scene, sky); F(mrt_scene, textures); F(mrt_scene, n_textures);
    S(mrt_render_desc); F(mrt_render_desc, rt); F(mrt_render_desc, frame); F(mrt_render_desc, scene);
    S(mrt_opts); F(mrt_opts, abi_version); F(mrt_opts, seed); F(mrt_opts, device); F(mrt_opts, shard_index); F(mrt_opts, shard_count); F(mrt_opts, shard_rows);
    F(mrt_opts, n_devices); F(mrt_opts, flags); F(mrt_opts, reserved);
    S(mrt_stats); F(mrt_stats, kernel_ms); F(mrt_stats, gather_ms); F(mrt_stats, samples); F(mrt_stats, segments); F(mrt_stats, launches); F(mrt_stats, lds_bytes);
    F(mrt_stats, block_threads); F(mrt_stats, scene_bytes); F(mrt_stats, k_split); F(mrt_stats, deferred); F(mrt_stats, img_ms); F(mrt_stats, reduce_ms); F(mrt_stats, kernel_features); F(mrt_stats, scene_in_lds);
    S(mrt_plan); F(mrt_plan, staging); F(mrt_plan, block_threads); F(mrt_plan, lds_bytes); F(mrt_plan, staged_bytes); F(mrt_plan, scene_bytes);
    F(mrt_plan, kernel_features); F(mrt_plan, tbvh_nodes); F(mrt_plan, tbvh_hot_nodes); F(mrt_plan, small_plain_grid); F(mrt_plan, walk_cap); F(mrt_plan, reserved);
    return 0;
}
