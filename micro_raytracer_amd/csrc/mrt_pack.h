// mrt_pack.h — host side of mrt_create: validate a mrt_render_desc, rebuild the reference's mesh
// octree, hoist every scene-only quantity and write the packed device blob of mrt_scene.h.
#pragma once
#include <string>
#include <vector>

#include "../../include/mrt.h"
#include "mrt_scene.h"

namespace mrt {

struct Packed {
    Params P;                    // pointers left null
    std::vector<u32> blob;       // P.blob_words words
    u32 nw = 0, nh = 0;
    u32 res_w = 0, res_h = 0;
    float gamma = 0, exp = 0;
    u32 features = 0;            // F_* bits of mrt_trace.h the scene needs
    bool all_ident = false;      // every instance untransformed (TAG_IDENT): mrt_create adds F_IDENT to the features
    bool tbvh_wide = false;      // the triangle-BVH table is the 4-wide one (PackOpts)
    u32 n_tex_u8 = 0, n_tex_f32 = 0, n_nodes = 0, n_leaf_ids = 0, n_tris = 0, n_xf = 0, n_bvh_nodes = 0, n_lin = 0, n_tbvh_nodes = 0;
};

struct PackOpts {
    // Triangle BVHs as the 4-WIDE table of mrt_scene.h (nodes of all meshes in level order, so that a prefix of the table is
    // the top of every tree) instead of the binary threaded one: for kernels built with F_DEEP, i.e. meshes beyond the LDS.
    bool tbvh_wide = false;
};

// Returns MRT_OK or an MRT_ERR_* code with a message in err.
int pack_scene(const mrt_render_desc *desc, Packed &out, std::string &err, const PackOpts &opts = PackOpts());

// Flattened octree of one mesh in the layout of mrt_scene.h (exposed for tests).
struct OctreeFlat {
    std::vector<float> nodes;     // NODE_WORDS words each (stored as raw words in a float vector)
    std::vector<u32> leaf_ids;
    u32 root = NO_NODE;
    bool empty_root = false;      // root has neither content nor children: the reference would panic
};
void build_octree(const float *tris, u32 n_tris, OctreeFlat &out);

// Binary sweep-SAH triangle BVH of one mesh (depth-first, mesh-relative skip links; pack_scene collapses it into the 4-wide
// table of mrt_scene.h); order[new id] = old id.  False: mesh cannot be bounded.
bool build_tbvh(const float *tris, u32 n_tris, std::vector<float> &nodes, std::vector<u32> &order);

// Lanczos3 resampling taps of image 0.24's imageops::resize for one axis (src/sampler.rs:98).
struct ResampleTaps {
    std::vector<u32> left;        // first source index per output index
    std::vector<u32> count;       // taps per output index
    std::vector<float> weight;    // cap weights per output index, normalised
    u32 cap = 0;
};
void lanczos3_taps(u32 src, u32 dst, ResampleTaps &out);

}  // namespace mrt
