// ptx_kernels.h -- launch interface of the wavefront stages (ptx_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/ptx.h"
#include "bvh8.h"
#include "traverse.h"

namespace ptx {

// The ray queue of one bounce, structure-of-arrays: three planes of 16-byte elements so that a
// wave's loads and stores are contiguous kilobytes.
//   plane 0: origin.xyz, path          path = sample_in_chunk * n_pixels + pixel
//   plane 1: direction.xyz, uv_state   uv_state = hash2(floatBitsToUint(in_uv)), constant per pixel
//   plane 2: throughput.xyz, pdf       running product of attenuations (rgen.glsl:80); lit integrator: pdf of the direction
// The traversal reads planes 0 and 1 when it fetches a ray (1 / direction and the octant are derived there: round 1
// carried them through the queue as a fourth plane, 16 B written and 16 B read per ray for three reciprocals), the
// shading stage reads and writes all three.
struct RayQueue {
    float4 *org, *dir, *thr;
};
constexpr uint32_t kRayBytes = 48;      // per ray over the three planes
inline RayQueue make_ray_queue(void *base, uint64_t capacity)
{
    RayQueue q; q.org = (float4 *)base; q.dir = q.org + capacity; q.thr = q.dir + capacity;
    return q;
}

// Result of the closest-hit query of one ray, 8 B: the shading stage recomputes (u, v) from it.
struct alignas(8) HitOut {
    float t;
    uint32_t tri;    // index into the leaf-ordered TriRec array; 0xFFFFFFFF = miss
};
static_assert(sizeof(HitOut) == 8, "HitOut must be 8 bytes");

// Per-triangle shading inputs in leaf order: what rchit.glsl:160-198 gathers through Offsets ->
// indices -> 3 x Vertex -> material_offset -> Material, pre-joined at ptx_set_instances time, plus
// the world-space positions (for the barycentrics) and the ids.  One 128-byte record = one L2 line
// and ONE dependent fetch per shaded hit (the chain ray -> triangle -> material -> texture table
// -> texel was five round trips deep).
struct alignas(128) ShadeRec {
    float v0[3]; uint32_t texture;      // bindless index of the base-colour texture, 0xFFFFFFFF: none (shared.glsl:31-32)
    float v1[3]; float roughness;
    float v2[3]; float base_r;
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float base_g, base_b;
    uint32_t instance, prim;
    uint32_t material;                  // row of SceneDev::lit_materials (lit integrator only)
};
static_assert(sizeof(ShadeRec) == 128, "ShadeRec must be 128 bytes");

// Lit integrator (build extension): per-material inputs beyond the ShadeRec, and the light table.
struct alignas(16) LitMaterial {
    float emissive[3]; float metallic;
    float transmission, ior;
    float pdf_area;      // light pick density per area for a point on a triangle of this material (luminance / total power)
    uint32_t mr_texture; // metallic-roughness texture index + 1, 0 = none
    uint32_t normal_texture, emissive_texture;   // index + 1, 0 = none (ptx_material_ext)
    // The material's TEXEL SET (0 flags: none, the taps go to the three textures one by one): base colour, normal and
    // metallic-roughness texels of one texel position side by side, see kTexelSet* below.
    uint32_t set_flags;
    uint32_t set_wh;     // width | height << 16 of the set's textures
    const uint32_t *set_texels;
    float normal_scale;  // glTF normalTexture.scale
    uint32_t base_slot;  // host-side note (ensure_texel_sets): the base colour texture slot (index + 1) the ShadeRecs of this build carry; no kernel reads it
};
static_assert(sizeof(LitMaterial) == 64, "LitMaterial must be 64 bytes");
// Texel sets (lit integrator, built by ptx_api.cpp: ensure_texel_sets).  A hit of the lit integrator reads up to three
// textures at ONE uv -- 12 scattered one-dword loads with bilinear filtering, and the stage's time follows the number of
// such requests, not their bytes (nearest instead of bilinear, 3 loads instead of 12, takes 8.3
// instead of 11.0 ms per step).  When the textures of a material have one size and one wrap / filter mode, the build
// interleaves them into one array with one ENTRY per fetch position, holding everything that fetch reads:
//   word A = base colour r, g, b + metallic-roughness g in the top byte; word B = normal r, g, b + metallic-roughness b
//   (no decode reads a texel's alpha; the metallic-roughness texel is put together again from the two top bytes);
//   nearest:  entry (x, y) = A, B of texel (x, y): 8 bytes, row-major;
//   bilinear: entry (k0x + 1, k0y + 1) for the lower texel indices k0 in [-1, size] that linear_axis_k0 can return =
//             A00 A10 A01 A11 [B00 B10 B01 B11] of the 2 x 2 footprint with wrap / clamp already applied by the same
//             axis_pair the plain path uses: 16 or 32 bytes, rows of width + 2 entries.  One or two 16-byte loads.
// Same texels, same arithmetic after the loads: the image keeps its bits.
constexpr uint32_t kTexelSetOn = 1u, kTexelSetRepeat = 2u, kTexelSetLinear = 4u, kTexelSetHasB = 8u;
constexpr uint32_t kTexelSetBaseSrgb = 1u << 8, kTexelSetNormalSrgb = 1u << 9, kTexelSetMrSrgb = 1u << 10;
struct alignas(16) LightRec {          // one emissive triangle, world space
    float a[3]; float pdf_area;        // pick probability / area
    float b[3]; float le_r;
    float c[3]; float le_g;
    float n[3]; float le_b;
    float uv0[2], uv1[2];              // the triangle's texture coordinates and its material's emissive texture (index + 1,
    float uv2[2]; uint32_t etex, _pad; // 0 = none): the radiance of a light sample is le x texel at the sampled point
};
static_assert(sizeof(LightRec) == 96, "LightRec must be 96 bytes");

// Device copy of a texture.  The texels are NOT row-major: ptx_add_texture re-orders them into tiles of 8 x 4 texels =
// 128 bytes = one cache line, tile rows of tiles_x tiles (width and height padded up to whole tiles; texel (x, y) lives
// at ((y / 4) * tiles_x + x / 8) * 32 + (y % 4) * 8 + x % 8).  The 2 x 2 footprint of a bilinear fetch then lies in 1.4
// lines on average instead of 2.1, and the lit shading stage, whose HBM-side traffic was 3.3 x its algorithmic bytes
// (profiles/r03a_lit_pmc_summary.json: 730 B per ray, TCC hit rate 0.59), moves a third less for its texture taps.
struct TexDesc {
    const uint8_t *texels;   // RGBA8, tiled as above
    uint32_t width, height, srgb /* bit0 sRGB, bit1 wrap = repeat, bit2 bilinear */, tiles_x;
};
constexpr uint32_t kTexTileW = 8, kTexTileH = 4;
static_assert(sizeof(TexDesc) == 24, "TexDesc must be 24 bytes");
constexpr uint32_t kLdsTextures = 64;   // descriptors of the first textures are staged in LDS by the shading stage

struct SceneDev {
    const Bvh8Node *nodes; uint32_t n_nodes;
    const TriRec *tris;
    const uint32_t *tri_of_rank;        // TriRec::rank -> index into tris / shade (k_extend keeps its closest hit by rank)
    const ShadeRec *shade;
    const TexDesc *textures; uint32_t n_textures;
    const float *srgb_lut;   // 256 floats
    // lit integrator
    const LitMaterial *lit_materials; uint32_t n_lit_materials;
    const LightRec *lights; const float *light_cdf; uint32_t n_lights;
};

struct FrameDev {
    ptx_camera cam;
    uint32_t width, height, n_pixels;
    uint32_t seed;
    uint32_t mean_base;        // uniforms.accumulated_count of the chunk's first sample
    uint32_t rng_offset;       // params.rng_sample_offset
    uint32_t ids_sample;       // sample (within the chunk) whose first hit goes to the id image; 0xFFFFFFFF: none
    uint32_t max_depth;
    float t_min, t_max, normal_eps, z_near, z_far;
    uint32_t tex_times_factor; // PTX_SCENE_TEXTURE_TIMES_FACTOR
    float cam_origin[3];       // rgen.glsl:30 view_inverse * (0,0,0,1), evaluated once on the host (same expression, no contraction): the
                               // origin of every camera ray -- k_raygen does not store it, the bounce-0 stages take it from here
};

struct ImagesDev {
    float4 *albedo; float4 *normal; float *depth; uint2 *ids;
};

// counters[0 .. max_depth]: live rays entering each bounce; counters[kHitCounter]: hits of the chunk;
constexpr uint32_t kMaxDepthLimit = 64;
constexpr uint32_t kHitCounter = kMaxDepthLimit + 1;
constexpr uint32_t kNextBase = kMaxDepthLimit + 2;      // (unused since the fetch cursors became per-XCD heads, kept for the layout)
constexpr uint32_t kShadowCount = 2 * kMaxDepthLimit + 3; // counters[kShadowCount + b]: occlusion rays queued by bounce b (lit integrator)
constexpr uint32_t kShadowNext = 3 * kMaxDepthLimit + 4;  // their fetch cursor
// Fetch cursors of the traversal launches: the ray queue of a launch is cut into kFetchHeads contiguous segments, one head
// word each, 128 bytes apart (a single word takes ~88 M atomics/s chip-wide; MI355X_MICROARCH.md "dequeue": shard the head
// per XCD).  A wave pulls from the segment of its XCD (blockIdx % 8) and moves on to the others when that one is empty.
constexpr uint32_t kFetchHeads = 8;
constexpr uint32_t kHeadStride = 32;                      // words
constexpr uint32_t kHeadBase = 4 * kMaxDepthLimit + 5;    // heads of (bounce b, kind k: 0 path rays, 1 occlusion rays) at fetch_heads(b, k)
constexpr uint32_t kHeadsPerLaunch = kFetchHeads * kHeadStride;
constexpr uint32_t kNumCounters = kHeadBase + kMaxDepthLimit * 2 * kHeadsPerLaunch;
inline constexpr uint32_t fetch_heads(uint32_t bounce, uint32_t kind) { return kHeadBase + (bounce * 2u + kind) * kHeadsPerLaunch; }

// Leaf-ordered triangle indices fit 26 bits (the traversal kernel packs an index with a 6-bit lane number);
// ptx_set_instances refuses scenes with more triangles.
constexpr uint32_t kTriIndexBits = 26;
constexpr uint64_t kMaxSceneTriangles = (1ull << kTriIndexBits) - 1ull;

// traversal stack: entries held in LDS per lane, remainder in private memory
#ifndef PTX_LDS_STACK
#define PTX_LDS_STACK 5      // 5 x 8 B x 64 lanes = 2.5 KB of the traversal kernel's 5 KB-per-wave LDS budget (8 waves per SIMD; 5 measured = 6 = 7)
#endif
constexpr int kLdsStack = PTX_LDS_STACK;
constexpr int kOvfStack = 32 - PTX_LDS_STACK;
constexpr uint32_t kMaxBvhDepth = kLdsStack + kOvfStack;

void launch_raygen(hipStream_t s, const FrameDev &f, uint32_t n_paths, RayQueue rays, uint32_t *counters);
// `next`: the bounce's fetch cursor (zeroed by raygen); `max_waves`: resident-wave budget of the persistent grid
void launch_extend(hipStream_t s, const SceneDev &sc, const FrameDev &f, RayQueue rays, const uint32_t *count,
                   uint32_t *next, uint32_t capacity, uint32_t max_waves, HitOut *hits,
                   uint64_t *trav_totals /* NULL: uninstrumented */, bool any_hit = false, bool primary = false);
uint32_t extend_resident_waves(int device);
// every remaining bounce of the rays queued for bounce `depth0` (count in counters[depth0]), reference integrator
void launch_tail(hipStream_t s, const SceneDev &sc, const FrameDev &f, uint32_t depth0, RayQueue rays, uint32_t *counters,
                 uint32_t capacity, float4 *results);
// the same for the lit integrator (occlusion queries included); depth0 >= 1
void launch_tail_lit(hipStream_t s, const SceneDev &sc, const FrameDev &f, uint32_t depth0, RayQueue rays, uint32_t *counters,
                     uint32_t capacity, float4 *results);
void launch_shade(hipStream_t s, const SceneDev &sc, const FrameDev &f, const ImagesDev &img, uint32_t depth,
                  RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                  float4 *results, uint32_t grid_cap = 0 /* most blocks the launch starts; 0: the default (PTX_SHADE_GRID, 8192) */);
// Lit integrator: shading with next-event estimation.  Survivors go to rays_out, occlusion rays to
// `shadow` (count in counters[kShadowCount + depth]); radiance is accumulated in results[path].
void launch_shade_lit(hipStream_t s, const SceneDev &sc, const FrameDev &f, const ImagesDev &img, uint32_t depth,
                      RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                      RayQueue shadow, float4 *results, const uint32_t *order = nullptr /* k_material_sort's permutation */,
                      uint32_t grid_cap = 0);
// Material sorting option of the lit integrator: order[0 .. *count) = the queue positions grouped by the material class of
// their hit (sky, opaque, transmissive); class_count: 6 words of scratch
void launch_material_sort(hipStream_t s, const SceneDev &sc, const HitOut *hits, const uint32_t *count, uint32_t capacity,
                          uint32_t *class_count, uint32_t *order, uint32_t grid_cap = 0);
// adds the contribution of every unoccluded occlusion ray of bounce `depth` to its path's radiance
// texel sets of the lit integrator (LitMaterial): bytes of one set, and its build from the (tiled) textures
uint64_t texel_set_bytes(uint32_t w, uint32_t h, uint32_t set_flags);
void launch_build_texel_set(hipStream_t s, uint32_t *dst, const TexDesc &tb, const TexDesc &tn, const TexDesc &tm, uint32_t present,
                            uint32_t w, uint32_t h, uint32_t set_flags);
void launch_shadow_resolve(hipStream_t s, uint32_t depth, RayQueue shadow, const HitOut *hits, const uint32_t *counters,
                           uint32_t capacity, float4 *results, uint32_t grid_cap = 0);
uint32_t default_shade_grid();      // PTX_SHADE_GRID or 8192
void launch_resolve(hipStream_t s, const FrameDev &f, const ImagesDev &img, const float4 *results, uint32_t n_samples);
void launch_accumulate_stats(hipStream_t s, const uint32_t *counters, uint32_t max_depth, uint64_t n_samples,
                             uint64_t *totals);
constexpr int kNumTotals = 23;   // rays, hits, samples, nodes, tris, 6 x wave_sched, shadow rays, 8 x traversal phase cycles, append blocks / append loop passes / refill passes
void launch_scale_copy(hipStream_t s, const float4 *src, float4 *dst, uint64_t n, float scale);

// ---- device refit (ptx_refit.hip): instance motion without a host round trip
struct RefitLeafSrc { uint32_t instance; uint32_t v[3]; };                 // leaf-order triangle -> its instance and absolute vertex indices
struct RefitInstance { float xform[12]; float cof[9]; float _pad[3]; };    // 3x4 row-major transform, normal matrix (cofactors x sign of det)
static_assert(sizeof(RefitInstance) == 96, "RefitInstance must be 96 bytes");
struct RefitState {
    uint32_t ext_lo[3], ext_hi[3];    // scene extent as order-preserving uints (reset: lo = ~0, hi = 0)
    uint32_t non_finite, _pad;        // a transformed triangle has a non-finite coordinate: the host rebuilds
    double area_sum, root_area;       // sum of node half-areas, the root's: the cost figure of refit_bvh8
};
constexpr uint32_t kRefitBoxBytes = 24;   // sizeof(Box) in bvh_quant.h
void refit_state_reset(RefitState *host_copy);
void launch_refit_pack(hipStream_t s, const RefitLeafSrc *src, const ptx_vertex *verts, const RefitInstance *inst, uint32_t n_leaf,
                       bool world_normals, TriRec *tris, ShadeRec *shade, void *raw_box, RefitState *state);
void launch_refit_nodes(hipStream_t s, Bvh8Node *nodes, void *node_box, const void *raw_box, uint32_t first, uint32_t count,
                        RefitState *state);

}  // namespace ptx
