// mrt_scene.h — packed device layout of a flattened rt::Scene (reference src/rt.rs:82-190) and
// the kernel parameter block.  The host packer (mrt_pack.cpp) writes this layout once per
// mrt_create; every workgroup of the path-tracing kernel stages the whole blob into LDS.
//
// All tables are arrays of 32-bit words (f32 or u32); offsets are in words from the blob
// start, 16-byte aligned.  Everything that depends only on the scene (not on the ray) is
// hoisted here with the reference's own operation order, so the kernel reproduces the
// reference's per-call recomputation bit for bit:
//   plane      n^ = norm(n), d = (-n^).pos              src/rt.rs:404-405
//   sphere     r*r                                       src/rt.rs:342
//   box        0.5*size, (1/size)*2                      src/rt.rs:319, 416
//   triangle   e0 = v1-v0, e1 = v2-v0                    src/rt.rs:365-366
//   instance   R = rotate_y(-dir), L = lookat(-dir, up)  src/rt.rs:726-727
//   light      norm(-(norm(dir))) for directional lights src/rt.rs:1031-1034
#pragma once
#include "mrt_math.h"

namespace mrt {

// ---- record sizes (words) ----
constexpr u32 REND_WORDS = 16;
constexpr u32 INST_WORDS = 8;
constexpr u32 XF_WORDS = 20;
constexpr u32 MAT_WORDS = 16;
constexpr u32 LIGHT_WORDS = 8;
constexpr u32 TEX_WORDS = 4;
constexpr u32 MESH_WORDS = 12;
constexpr u32 TRI_WORDS = 9;
constexpr u32 NODE_WORDS = 8;

// REND: [0] kind  [1] inst_off  [2] inst_cnt  [3] flags  [4..15] geometry
enum : u32 { REND_KIND = 0, REND_INST_OFF = 1, REND_INST_CNT = 2, REND_FLAGS = 3, REND_GEO = 4 };
enum : u32 { RF_HAS_MAPS = 1u };
// geometry words: sphere [4] r*r | plane [4..6] n^, [7..9] n | box [4..6] half, [7..9] (1/size)*2 |
// triangle [4..6] v0, [7..9] e0, [10..12] e1 | mesh [4] mesh index

// INST (the flat traversal table, one record per renderer instance in the reference's iteration order;
// two aligned 16-byte reads, everything the common kinds need):
//   [0..2] pos
//   [3]    sphere r*r | plane d = (-n^).pos | box half.x
//   [4]    tag = kind | identity-valued transform ? 8 : 0 | (word offset of the transform inside the XF table) << 4
//   [5..7] plane n^ | box [5] half.y [6] half.z
// INSTX (read per lane after a hit): [0] renderer index  [1..3] plane world normal norm(R*(L*n))
enum : u32 { INST_POS = 0, INST_P3 = 3, INST_TAG = 4, INST_P5 = 5 };
enum : u32 { INSTX_REND = 0, INSTX_PLANE_NW = 1 };
constexpr u32 INSTX_WORDS = 4;
constexpr u32 TAG_KIND_MASK = 7u, TAG_IDENT = 8u, TAG_XF_SHIFT = 4u;

// XF: [0..8] L (lookat)  [9..17] R (rotate_y)  [18] 1 if both equal the identity as values
enum : u32 { XF_L = 0, XF_R = 9, XF_IDENT = 18 };

// MAT: [0..2] albedo [3] rough [4] metal [5] glass [6] opacity [7] emit [8..13] map ids (tex rmap mmap gmap omap emap, -1 none)
enum : u32 { MAT_ALBEDO = 0, MAT_ROUGH = 3, MAT_METAL = 4, MAT_GLASS = 5, MAT_OPACITY = 6, MAT_EMIT = 7, MAT_MAP = 8 };
enum : u32 { MAP_TEX = 0, MAP_ROUGH = 1, MAP_METAL = 2, MAP_GLASS = 3, MAP_OPACITY = 4, MAP_EMIT = 5 };

// LIGHT: [0] kind [1..3] point: pos / dir: norm(-(norm(dir)))  [4] pwr  [5..7] color
enum : u32 { LIGHT_KIND = 0, LIGHT_V = 1, LIGHT_PWR = 4, LIGHT_COLOR = 5 };

// TEX: [0] w [1] h [2] offset (words for f32 texels, bytes for u8 texels, from blob start) [3] format
enum : u32 { TEX_W = 0, TEX_H = 1, TEX_OFF = 2, TEX_FMT = 3 };
enum : u32 { TEXFMT_NONE = 0, TEXFMT_F32 = 1, TEXFMT_U8 = 2 };

// MESH: [0] first triangle [1] triangle count [2] root node (0xffffffff: no octree) [3] leaf-id base
//       [4] root of the mesh's triangle BVH: node index in the TBVH table, binary or 4-wide (0xffffffff: none)  [5] number of ids in the mesh's octree leaf lists
//       [6..8] centre, [9..11] half size of the mesh's bounds in mesh coordinates (the culling margin of a ray is derived from them)
enum : u32 { MESH_TRI0 = 0, MESH_NTRI = 1, MESH_ROOT = 2, MESH_LEAF0 = 3, MESH_TBVH = 4, MESH_NIDS = 5, MESH_BC = 6, MESH_BH = 9 };
constexpr u32 NO_NODE = 0xffffffffu;

// NODE: [0..2] 0.5*aabb  [3..5] rel_pos  [6] first child node | first leaf id  [7] count | leaf<<31
enum : u32 { NODE_HALF = 0, NODE_REL = 3, NODE_FIRST = 6, NODE_COUNT = 7 };

// BVH over renderer instances (scenes with many instances): threaded (stackless) node array in depth-first
// order.  NODE: [0..2] box centre  [3..5] box half size  [6] skip = next node when this one is missed or is a leaf  [7] leaf word:
// 0 for an internal node (next = this + 1), else count << 24 | first index into the instance-id list.
// Instances that cannot be bounded (planes, non-orthonormal transforms) are in the linear list instead.
//
// Triangle BVH of a mesh (TBVH): a pure accelerator for Renderer::intersect on meshes.  The reference answers with the first
// minimum / last maximum of the triangle hits among the *candidates* its octree walk collects (src/rt.rs:740-770); a triangle
// is a candidate when the ray passes the exact box tests of the root, ..., leaf chain of some octree leaf listing it.  The
// kernel finds the triangles the ray can hit through the TBVH (conservative culling), runs the exact triangle test and
// then decides candidacy from the membership table, so no box the ray misses and no triangle it misses is touched:
//   MEMB  (one word per triangle, off_memb + MESH_TRI0 + id): count << 24 | first entry
//   MEMBE (off_membe + entry): octree leaf node (relative to the mesh's root node) << 22 | slot, where slot is the
//         position of this occurrence in the mesh's leaf-id list = its rank in the reference's candidate order
//   PARENT (one word per octree node, off_parent + node): parent node, NO_NODE for a root
// The mesh's triangles are stored in TBVH leaf order; the octree leaf lists hold the permuted ids (a relabelling the
// reference cannot observe: ids only select a triangle and are compared for equality, src/rt.rs:756).
//
// Two forms of the table (pack_scene PackOpts::tbvh_wide; mrt_create picks by where the mesh lives):
// BINARY (meshes in LDS): the node record of the instance BVH above, mesh-local coordinates, leaf word = count << 24 | first
// triangle of the mesh, threaded in depth-first order.
// 4-WIDE (meshes beyond the LDS, kernels with F_DEEP): a node holds the boxes of its (up to) four children, so one visit -- one
// round of independent 16-byte reads -- decides four subtrees, and only children whose box the ray hits are ever visited: a
// fifth of the dependent round trips of the binary walk.  Node = B4_WORDS words, structure of arrays so that every read is an
// aligned 16-byte one (28 words: an odd multiple of 4, which spreads the nodes different lanes read over 16 bank groups):
//   [0..3] centre x of child 0..3  [4..7] centre y  [8..11] centre z  [12..15] half size x  [16..19] y  [20..23] z
//   [24..27] child word: 0 = empty slot; leaf: count << 24 | first triangle of the mesh (count 1..4, bit 31 clear);
//            internal: B4_INTERNAL | INDEX of the child node inside the TBVH table.  Internal children come first and are
//            consecutive nodes of the table: child k of a node is node (index of child 0) + k
// Built by collapsing the sweep-SAH binary tree (the child with the largest box is opened until four slots are full); the
// nodes of ALL meshes are stored in level order (roots first), so that a PREFIX of the table is the top of every tree:
// the first Params.n_tbvh_hot nodes are staged, the rest is read from global memory.
constexpr u32 B4_WORDS = 28;
constexpr u32 B4_INTERNAL = 0x80000000u;
constexpr u32 kWalkCapMin = 8u, kWalkCapMax = 16u, kWalkCapDefault = 16u;      // entries of a lane's walk area (Params.walk_cap)
enum : u32 { B4_CX = 0, B4_CY = 4, B4_CZ = 8, B4_HX = 12, B4_HY = 16, B4_HZ = 20, B4_CHILD = 24 };
constexpr u32 BVH_WORDS = 8;
constexpr u32 MEMB_SLOT_BITS = 22u, MEMB_SLOT_MASK = (1u << MEMB_SLOT_BITS) - 1u;
enum : u32 { BVH_C = 0, BVH_H = 3, BVH_SKIP = 6, BVH_LEAF = 7 };
constexpr u32 BVH_END = 0xffffffffu;

enum : u32 { KIND_SPHERE = 0, KIND_PLANE = 1, KIND_BOX = 2, KIND_TRIANGLE = 3, KIND_MESH = 4 };
enum : u32 { LK_POINT = 0, LK_DIR = 1 };

// samples per chunk of the canonical accumulation order (mrt_trace.h render_pixel)
constexpr u32 kChunk = 16;

struct Params {
    // frame / sampling
    u32 nw, nh;
    u32 local_rows, shard_index, shard_count, shard_rows;
    u32 n_samples, sample_base;
    u32 k_split;             // lanes per pixel (each owns every k_split-th sample chunk of the launch)
    u32 to_planes;           // look-ahead launches of the per-call path (k_split == 1): every sample is a chunk of its own, written to
                             // plane (sample - sample_base) of `partial`; the accumulator is not touched
    u32 seed_lo, seed_hi;
    u32 bounce;
    float q;                 // 1 - min(loss, 1), src/rt.rs:571
    float w, h, aspect;      // src/rt.rs:938-940
    float inv2tan;           // 1 / (2 tan(rad(fov/2))), src/rt.rs:902-906
    float cam_pos[3];
    float aprt, foc;
    float cam_L[9], cam_R[9];   // lookat(cam.dir, up), rotate_y(cam.dir), src/rt.rs:925-927
    u32 cam_ident;              // both equal the identity as values
    float sky[3];
    float sky_init[3];       // sky.color * sky.pwr, src/rt.rs:964
    // scene tables
    u32 n_rend, n_light, n_inst;
    u32 n_lin, n_bvh_nodes;   // instance BVH: linear-list length, node count (0: every instance is scanned linearly)
    float inst_ksq;           // ... and per unit of SQUARED origin distance (spheres among the bounded instances: 4e-6 / smallest radius)
    float inst_k, inst_kpos;  // instance BVH: culling margin per unit of origin distance / of coordinate magnitude (mrt_trace.h cull_margin):
                              // 4e-3 / 1e-5 when a sphere is among the bounded instances, 1e-4 / 2e-6 otherwise (pack_scene)
    u32 off_lin, off_bvh, off_bvhinst;
    u32 off_cam;              // cam_L, cam_R as 18 words of the blob: the kernel reads the matrices from there (only rolled / turned cameras
                              // need them: kept out of the scalar registers)
    u32 off_rend, off_inst, off_instx, off_xf, off_mat, off_light, off_tex, off_lut, off_mesh, off_tri, off_node, off_leaf;
    u32 off_tbvh, off_memb, off_membe, off_parent;
    u32 n_tbvh_hot;           // F_DEEP: triangle-BVH nodes with index < n_tbvh_hot are in LDS (set by mrt_create from the LDS budget)
    u32 walk_cap;             // entries of the per-lane walk area of the mesh kernels (LDS column: node stack from the bottom, leaf
                              // queue from the top; mrt_trace.h mesh_isect), set by mrt_create from the LDS budget
    u32 blob_words;
    u32 lds_words;            // words a workgroup stages in LDS: everything before the octree leaf lists when every mesh has a
                              // triangle BVH (the lists are then only read, from global memory, by rays that cannot be culled)
    // Shorter staging prefixes of the blob (table order: records, transforms, materials, LUT, node arrays | triangles,
    // membership tables | texels | leaf lists), for kernels that leave the rarely touched tables in global memory (L2):
    u32 lds_words_warm;       // F_COLD: everything up to and including triangles and membership tables; texels (one lookup per shaded
                              // hit) stay out.  (Until round 4 the membership tables -- two dependent reads per triangle HIT -- stayed out
                              // too: 3713 -> 3808 Msamples/s on the 967-triangle bench scene with them in LDS, at 9 instead of 13 queue entries.)
    u32 lds_words_hot;        // F_COLD | F_DEEP: every table a traversal step reads (records ... node arrays); triangles stay out too.
                              // mrt_create shrinks it to off_tbvh + n_tbvh_hot nodes when not even the node arrays fit
    u32 tiles_x, tiles_y;    // 8x8-pixel wave tiles per workgroup in x and y
    u32 count_segments;
    // device pointers
    const u32 *blob;
    float *accum;            // [local_rows][nw][3]
    float *partial;          // [chunks of this launch][padded_rows][nw][3], used when k_split > 1
    unsigned long long partial_stride;   // floats per chunk plane
    unsigned long long *segments;
    u32 *tile_counter;       // persistent launches: next (tile, sample-split lane) index, zeroed before the launch
    u32 persist_grid;        // workgroups of a persistent launch (0: one workgroup per tile block, blockIdx addresses the tiles)
};

}  // namespace mrt
