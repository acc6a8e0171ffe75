// ptx_kernels.h -- launch interface of the wavefront stages (ptx_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/ptx.h"
#include "bvh8.h"
#include "traverse.h"

namespace ptx {

// One in-flight path segment, 48 B = three 16-byte loads/stores, queue order.
struct alignas(16) RayRec {
    float ox, oy, oz; uint32_t path;        // path = sample_in_chunk * n_pixels + pixel
    float dx, dy, dz; uint32_t uv_state;    // hash2(floatBitsToUint(in_uv)), constant per pixel
    float tx, ty, tz; uint32_t _pad;        // running product of attenuations (rgen.glsl:80)
};
static_assert(sizeof(RayRec) == 48, "RayRec must be 48 bytes");

// Per-triangle shading inputs in leaf order, 64 B = four 16-byte loads: what rchit.glsl:160-190
// gathers through Offsets -> indices -> 3 x Vertex -> material_offset, pre-joined at
// ptx_set_instances time.
struct alignas(16) ShadeRec {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    uint32_t material;
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec must be 64 bytes");

struct TexDesc {
    const uint8_t *texels;   // RGBA8, row 0 first
    uint32_t width, height, srgb, _pad;
};

struct SceneDev {
    const Bvh8Node *nodes; uint32_t n_nodes;
    const TriRec *tris;
    const ShadeRec *shade;
    const ptx_material *materials;
    const TexDesc *textures;
    const float *srgb_lut;   // 256 floats
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
};

struct ImagesDev {
    float4 *albedo; float4 *normal; float *depth; uint2 *ids;
};

// counters[0 .. max_depth]: live rays entering each bounce; counters[kHitCounter]: hits of the chunk;
constexpr uint32_t kMaxDepthLimit = 64;
constexpr uint32_t kHitCounter = kMaxDepthLimit + 1;
constexpr uint32_t kNextBase = kMaxDepthLimit + 2;      // counters[kNextBase + b]: fetch cursor of bounce b's extend
constexpr uint32_t kNumCounters = 2 * kMaxDepthLimit + 3;

// traversal stack: entries held in LDS per lane, remainder in private memory
#ifndef PTX_LDS_STACK
#define PTX_LDS_STACK 8
#endif
constexpr int kLdsStack = PTX_LDS_STACK;
constexpr int kOvfStack = 32 - PTX_LDS_STACK;
constexpr uint32_t kMaxBvhDepth = kLdsStack + kOvfStack;

void launch_raygen(hipStream_t s, const FrameDev &f, uint32_t n_paths, RayRec *rays, uint32_t *counters);
// `next`: the bounce's fetch cursor (zeroed by raygen); `max_waves`: resident-wave budget of the persistent grid
void launch_extend(hipStream_t s, const SceneDev &sc, const FrameDev &f, const RayRec *rays, const uint32_t *count,
                   uint32_t *next, uint32_t capacity, uint32_t max_waves, HitRec *hits,
                   uint64_t *trav_totals /* NULL: uninstrumented */);
uint32_t extend_resident_waves(int device);
void launch_shade(hipStream_t s, const SceneDev &sc, const FrameDev &f, const ImagesDev &img, uint32_t depth,
                  const RayRec *rays_in, const HitRec *hits, uint32_t *counters, uint32_t capacity, RayRec *rays_out,
                  float4 *results);
void launch_resolve(hipStream_t s, const FrameDev &f, const ImagesDev &img, const float4 *results, uint32_t n_samples);
void launch_accumulate_stats(hipStream_t s, const uint32_t *counters, uint32_t max_depth, uint64_t n_samples,
                             uint64_t *totals);
void launch_scale_copy(hipStream_t s, const float4 *src, float4 *dst, uint64_t n, float scale);

}  // namespace ptx
