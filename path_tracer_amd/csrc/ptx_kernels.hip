// ptx_kernels.hip -- gfx950 wavefront path-tracing stages.  Hand-written HIP for CDNA4
// (wave64); no CUDA shims, no dual paths.
//
//   raygen   rgen.glsl:20-60     camera ray per (sample, pixel) path
//   extend   rgen.glsl:66-78     closest hit = software traversal of the compressed 8-wide BVH
//   shade    rchit.glsl:158-254 / rmiss.glsl:18-34 + rgen.glsl:80-107: scatter or terminate,
//            G-buffers, wave64 ballot + mbcnt compaction of the surviving rays
//   resolve  rgen.glsl:124-141   running-mean accumulation, samples applied in order
//
// All paths of a chunk advance bounce by bounce (one extend + one shade launch per bounce), so
// `depth` is uniform per launch and every pixel receives its samples in accumulated_count
// order -- the image is bit-identical to the reference's frame-by-frame loop.
#include <hip/hip_runtime.h>

#include "ptx_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace ptx {

namespace {

constexpr int kExtendBlock = 64;    // one wave per workgroup: resources free as soon as the wave retires
constexpr int kShadeBlock = 1024;   // one queue-tail atomic per 1024 rays: a single word takes only ~88 M atomics/s

__device__ __forceinline__ f3 mat4_mul(const float *m, float x, float y, float z, float w)
{
    // GLSL mat4 * vec4 = c0*x + c1*y + c2*z + c3*w, summed left to right
    f3 r;
    r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
    r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
    return r;
}

// ---------------------------------------------------------------- raygen
__global__ void __launch_bounds__(256) k_raygen(FrameDev f, uint32_t n_paths, RayRec *__restrict__ rays,
                                                uint32_t *__restrict__ counters)
{
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p == 0) {
        counters[0] = n_paths;
        for (uint32_t i = 1; i < kNumCounters; ++i) counters[i] = 0;
    }
    if (p >= n_paths) return;
    uint32_t pix = p % f.n_pixels;
    uint32_t y = pix / f.width, x = pix - y * f.width;
    // rgen.glsl:25-28
    float pcx = (float)x + 0.5f, pcy = (float)(f.height - y - 1u) + 0.5f;
    float uvx = pcx / (float)f.width, uvy = pcy / (float)f.height;
    float dx = uvx * 2.0f - 1.0f, dy = uvy * 2.0f - 1.0f;
    // rgen.glsl:30-34
    f3 origin = mat4_mul(f.cam.view_inverse, 0.0f, 0.0f, 0.0f, 1.0f);
    f3 target = mat4_mul(f.cam.proj_inverse, dx, dy, 1.0f, 1.0f);
    f3 tn = normalize(target);
    f3 dir = normalize(mat4_mul(f.cam.view_inverse, tn.x, tn.y, tn.z, 0.0f));

    float4 *out = reinterpret_cast<float4 *>(rays + p);
    out[0] = make_float4(origin.x, origin.y, origin.z, __uint_as_float(p));
    out[1] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(hash2(f2u(uvx), f2u(uvy))));
    out[2] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);   // rgen.glsl:62
}

// ---------------------------------------------------------------- extend
// Persistent waves with dynamic ray fetch: a wave keeps stepping its 64 traversal state machines
// and, whenever kRefill or more lanes have finished, refills them from the bounce's ray queue with
// one atomic per wave.  Path lengths through the BVH are heavy-tailed; without the refill a wave
// runs at the pace of its slowest ray (11 % VALU lane utilisation measured, profiles/r01a).
struct LaneStack {
    uint2 *lds;                 // this lane's column of the workgroup's LDS stack
    uint2 ovf[kOvfStack];
    int sp;
    __device__ __forceinline__ void clear() { sp = 0; }
    __device__ __forceinline__ void push(uint32_t base, uint32_t mask)
    {
        if (sp < kLdsStack) lds[sp * kExtendBlock] = make_uint2(base, mask);
        else ovf[sp - kLdsStack] = make_uint2(base, mask);
        ++sp;
    }
    __device__ __forceinline__ void pop(uint32_t &base, uint32_t &mask)
    {
        --sp;
        uint2 v = (sp < kLdsStack) ? lds[sp * kExtendBlock] : ovf[sp - kLdsStack];
        base = v.x; mask = v.y;
    }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
};


// Asynchronous fetch of one 80-byte node per lane straight into LDS (gfx950 LDS-DMA,
// global_load_lds_dwordx4: per-lane global address, destination = wave-uniform base + lane * 16;
// no VGPRs are held while the load is in flight).  Lanes that do not call it are exec-masked.
__device__ __forceinline__ void prefetch_node(const Bvh8Node *nodes, uint32_t ni, uint4 *slot)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(nodes + ni);
#pragma unroll
    for (int k = 0; k < 5; ++k)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k),
                                         (__attribute__((address_space(3))) void *)(slot + k * 64), 16, 0, 0);
}

struct ExtendSched { int refill, node_min, tri_min; uint32_t chunk; };   // lane-count thresholds (wave-uniform)

#ifndef PTX_EXTEND_WAVES_PER_EU
#define PTX_EXTEND_WAVES_PER_EU 4
#endif

template <bool kStats>
__global__ void __launch_bounds__(kExtendBlock) __attribute__((amdgpu_waves_per_eu(PTX_EXTEND_WAVES_PER_EU, PTX_EXTEND_WAVES_PER_EU)))
k_extend(SceneDev sc, float t_min, float t_max,
                                                          const RayRec *__restrict__ rays,
                                                          const uint32_t *__restrict__ count,
                                                          uint32_t *__restrict__ next,
                                                          HitRec *__restrict__ hits,
                                                          unsigned long long *__restrict__ trav_totals, ExtendSched sched)
{
    __shared__ uint2 s_stack[kLdsStack * kExtendBlock];
    // per-lane landing slot of the node prefetch: piece k of lane L at s_node[k * 64 + L]
    __shared__ uint4 s_node[kExtendBlock * 5];
    const uint32_t n = *count;
    const uint32_t lane_id = threadIdx.x;
    LaneStack st;
    st.lds = s_stack + threadIdx.x;
    st.sp = 0;
    Traversal<LaneStack, kStats> tr;
    tr.ng_mask = 0; tr.tg_mask = 0;
    bool active = false;        // this lane holds an unfinished ray
    bool exhausted = false;     // wave-uniform: the queue has been handed out completely
    uint32_t ray = 0;
    uint32_t next_ni = 0xFFFFFFFFu;   // node this lane visits next (already requested), 0xFFFFFFFF: none left
    unsigned long long tot_nodes = 0, tot_tris = 0;
    unsigned long long w_iters = 0, w_node_blocks = 0, w_node_lanes = 0, w_tri_blocks = 0, w_tri_lanes = 0, w_refills = 0;   // wave-level (kStats)

    // Rays are handed out in two levels: a wave reserves a chunk of the queue with one atomic and
    // refills its idle lanes from that chunk.  (One atomic per refill on a single word saturates
    // at ~88 M/s chip-wide and was the bottleneck: MI355X_MICROARCH.md "dequeue".)
    uint32_t chunk_next = 0, chunk_end = 0;          // wave-uniform
    const uint32_t chunk = sched.chunk;

    for (;;) {
        unsigned long long idle = __ballot(!active);
        if (!exhausted && __popcll(idle) >= sched.refill) {
            if (kStats) w_refills++;
            for (int pass = 0; pass < 2 && idle != 0ull; ++pass) {
                if (chunk_next == chunk_end) {
                    uint32_t base = 0;
                    if (threadIdx.x == 0) base = atomicAdd(next, chunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base >= n) { exhausted = true; break; }
                    chunk_next = base;
                    chunk_end = (n - base < chunk) ? n : base + chunk;
                }
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                uint32_t avail = chunk_end - chunk_next;
                if (!active && rank < avail) {
                    uint32_t i = chunk_next + rank;
                    const float4 *rp = reinterpret_cast<const float4 *>(rays + i);
                    float4 r0 = rp[0], r1 = rp[1];
                    bool go = tr.init(sc.n_nodes, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), t_min, t_max, st);
                    next_ni = go ? tr.node_pick(st) : 0xFFFFFFFFu;      // the root
                    if (go) prefetch_node(sc.nodes, next_ni, s_node);
                    ray = i;
                    active = true;
                }
                uint32_t n_idle = (uint32_t)__popcll(idle);
                chunk_next += (n_idle < avail) ? n_idle : avail;
                idle = __ballot(!active);
            }
        }
        unsigned long long live = __ballot(active);
        if (live == 0) {
            if (exhausted) break;
            continue;
        }
        // step until enough lanes have finished to make a refill worthwhile (or, once the queue
        // is empty, until every lane is done)
        const int stop_at = exhausted ? 0 : (64 - sched.refill);
        do {
            // Node visits and triangle tests are two long instruction streams; run each only when
            // enough lanes want it (or nothing else can make progress), so neither executes for a
            // handful of lanes while the rest of the wave idles.
            // A lane visits a node when it has no triangle pending; its node was requested one
            // step earlier (see prefetch_node) so the fetch overlapped the triangle tests.
            bool want_node = active && tr.wants_node();
            const int nb = __popcll(__ballot(want_node));
            const int tb = __popcll(__ballot(active && tr.wants_tri()));
            const bool run_node = nb > 0 && (nb >= sched.node_min || tb < sched.tri_min);
            if (kStats) { w_iters++; if (run_node) { w_node_blocks++; w_node_lanes += nb; } }
            if (run_node) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // prefetched nodes have landed
                if (want_node) {
                    if (next_ni != 0xFFFFFFFFu) {
                        Words80 nd;
                        uint4 a = s_node[lane_id], b = s_node[64 + lane_id], c = s_node[128 + lane_id],
                              d = s_node[192 + lane_id], e = s_node[256 + lane_id];
                        nd.w[0] = a.x; nd.w[1] = a.y; nd.w[2] = a.z; nd.w[3] = a.w;
                        nd.w[4] = b.x; nd.w[5] = b.y; nd.w[6] = b.z; nd.w[7] = b.w;
                        nd.w[8] = c.x; nd.w[9] = c.y; nd.w[10] = c.z; nd.w[11] = c.w;
                        nd.w[12] = d.x; nd.w[13] = d.y; nd.w[14] = d.z; nd.w[15] = d.w;
                        nd.w[16] = e.x; nd.w[17] = e.y; nd.w[18] = e.z; nd.w[19] = e.w;
                        tr.node_eval(nd);
                        // the next node to visit is already determined: request it now
                        next_ni = tr.next_group(st) ? tr.node_pick(st) : 0xFFFFFFFFu;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slot fully read before it is refilled
                        if (next_ni != 0xFFFFFFFFu) prefetch_node(sc.nodes, next_ni, s_node);
                    } else {
                        *reinterpret_cast<float4 *>(hits + ray) = make_float4(tr.hit.t, tr.hit.u, tr.hit.v, __uint_as_float(tr.hit.tri));
                        if (kStats) { tot_nodes += tr.stats.nodes; tot_tris += tr.stats.tris; }
                        active = false;
                    }
                }
            }
            bool want_tri = active && tr.wants_tri();
            const int tb2 = __popcll(__ballot(want_tri));
            if (tb2 > 0 && (!run_node || tb2 >= sched.tri_min)) {
                if (kStats) { w_tri_blocks++; w_tri_lanes += tb2; }
                if (want_tri) tr.tri_step(sc.tris);
            }
        } while (__popcll(__ballot(active)) > stop_at);
    }
    if (kStats) {
        // instrumented variant (ptx_set_profiling level 2): nodes fetched / triangles tested
        atomicAdd(&trav_totals[0], tot_nodes);
        atomicAdd(&trav_totals[1], tot_tris);
        if (threadIdx.x == 0) {
            atomicAdd(&trav_totals[2], w_iters); atomicAdd(&trav_totals[3], w_node_blocks); atomicAdd(&trav_totals[4], w_node_lanes);
            atomicAdd(&trav_totals[5], w_tri_blocks); atomicAdd(&trav_totals[6], w_tri_lanes); atomicAdd(&trav_totals[7], w_refills);
        }
    }
}

// ---------------------------------------------------------------- shade
__device__ __forceinline__ f3 sample_texture(const TexDesc &tx, const float *lut, float u, float v)
{
    // vulkano SamplerCreateInfo::default(): nearest, clamp-to-edge, LOD 0 (graphics.rs:453-459)
    float fu = floorf(u * (float)tx.width), fv = floorf(v * (float)tx.height);
    float mu = (float)(tx.width - 1u), mv = (float)(tx.height - 1u);
    if (!(fu >= 0.0f)) fu = 0.0f;
    if (fu > mu) fu = mu;
    if (!(fv >= 0.0f)) fv = 0.0f;
    if (fv > mv) fv = mv;
    uint32_t texel = *reinterpret_cast<const uint32_t *>(tx.texels + ((uint64_t)(uint32_t)fv * tx.width + (uint32_t)fu) * 4u);
    uint32_t r = texel & 0xFFu, g = (texel >> 8) & 0xFFu, b = (texel >> 16) & 0xFFu;
    if (tx.srgb) return mk3(lut[r], lut[g], lut[b]);
    return mk3((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f);
}

__global__ void __launch_bounds__(kShadeBlock) k_shade(SceneDev sc, FrameDev f, ImagesDev img, uint32_t depth,
                                                        const RayRec *__restrict__ rays_in,
                                                        const HitRec *__restrict__ hits,
                                                        uint32_t *__restrict__ counters,
                                                        RayRec *__restrict__ rays_out, float4 *__restrict__ results)
{
    __shared__ float s_lut[256];
    __shared__ uint32_t s_wave_live[kShadeBlock / 64], s_wave_hits[kShadeBlock / 64];
    __shared__ uint32_t s_base;
    if (threadIdx.x < 256) s_lut[threadIdx.x] = sc.srgb_lut[threadIdx.x];
    __syncthreads();

    uint32_t i = blockIdx.x * kShadeBlock + threadIdx.x;
    bool valid = i < counters[depth];
    bool alive = false, was_hit = false;
    float4 o0, o1, o2;
    if (valid) {
        const float4 *rp = reinterpret_cast<const float4 *>(rays_in + i);
        float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        float4 hr = *reinterpret_cast<const float4 *>(hits + i);
        f3 ro = mk3(r0.x, r0.y, r0.z), rd = mk3(r1.x, r1.y, r1.z);
        uint32_t path = __float_as_uint(r0.w), uv_state = __float_as_uint(r1.w);
        uint32_t sample = path / f.n_pixels, pix = path - sample * f.n_pixels;
        uint32_t acc = f.mean_base + sample;            // uniforms.accumulated_count of this path's frame
        uint32_t tri = __float_as_uint(hr.w);

        f3 att, normal, new_o = ro, new_d = rd;
        float dist;
        int done;
        uint32_t id_inst = 0xFFFFFFFFu, id_prim = 0xFFFFFFFFu;
        if (tri == 0xFFFFFFFFu) {
            // rmiss.glsl:18-34
            att = sky_color(rd);
            done = 1;
            dist = 3.402823466e+38f;
            normal = mk3(0.0f, 0.0f, 0.0f);
        } else {
            was_hit = true;
            // rchit.glsl:160-183
            const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
            float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
            f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
            float bx = 1.0f - hr.y - hr.z, by = hr.y, bz = hr.z;
            normal = normalize((n0 * bx + n1 * by) + n2 * bz);
            float tu = (s2.y * bx + s2.w * by) + s3.y * bz;
            float tv = (s2.z * bx + s3.x * by) + s3.z * bz;
            f3 world_pos = ro + rd * hr.x;                                   // :186
            uint32_t mi = __float_as_uint(s3.w);
            const float4 *mp = reinterpret_cast<const float4 *>(sc.materials + mi);
            float4 m0 = mp[0], m1 = mp[1];                                   // :190
            f3 albedo;
            if (__float_as_uint(m1.x)) {                                     // :193-198
                albedo = sample_texture(sc.textures[__float_as_uint(m1.y)], s_lut, tu, tv);
            } else {
                albedo = mk3(m0.x, m0.y, m0.z);
            }
            float roughness = m1.w;                                          // :200
            float k2 = 2.0f * dot(normal, rd);                               // :209 reflect()
            f3 reflected = rd - normal * k2;
            f3 rr = normalize(cosine_hemisphere_sample(uv_state, depth, acc + f.rng_offset, f.seed, normal));
            float om = 1.0f - roughness;                                     // :220 mix(), unnormalised
            f3 scatter = reflected * om + rr * roughness;
            if (dot(scatter, normal) <= 0.0f) {                              // :223-230
                done = 1;
                att = mk3(0.0f, 0.0f, 0.0f);
            } else {
                done = 0;
                att = albedo;
            }
            new_o = world_pos + normal * f.normal_eps;                       // :242
            new_d = scatter;
            dist = length(new_o - ro);                                       // :254
            if (depth == 0 && sample == f.ids_sample) {
                const uint32_t *tw = reinterpret_cast<const uint32_t *>(sc.tris + tri);
                id_inst = tw[3]; id_prim = tw[7];
            }
        }
        f3 thr = mk3(r2.x, r2.y, r2.z) * att;                                // rgen.glsl:80

        if (depth == 0) {                                                    // rgen.glsl:83-99
            if (sample == f.ids_sample) img.ids[pix] = make_uint2(id_inst, id_prim);
            if (acc == 0) {
                float z_range = f.z_far - f.z_near;
                float zn = (dist - f.z_near) / z_range;
                zn = fminf(fmaxf(zn, 0.0f), 1.0f);
                img.depth[pix] = zn;
                img.normal[pix] = make_float4(normal.x, normal.y, normal.z, 1.0f);
            }
        }
        // rgen.glsl:101-103
        if (done == 1 || depth + 1u >= f.max_depth) {
            results[path] = make_float4(thr.x, thr.y, thr.z, 1.0f);
        } else {
            alive = true;
            o0 = make_float4(new_o.x, new_o.y, new_o.z, r0.w);
            o1 = make_float4(new_d.x, new_d.y, new_d.z, r1.w);
            o2 = make_float4(thr.x, thr.y, thr.z, 0.0f);
        }
    }

    // Compaction: wave64 ballots give each wave its survivor count, the block's 16 counts are
    // summed through LDS and ONE atomic per block reserves a contiguous run of the next queue.
    unsigned long long live = __ballot(alive);
    unsigned long long hitm = __ballot(was_hit);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) { s_wave_live[wave] = (uint32_t)__popcll(live); s_wave_hits[wave] = (uint32_t)__popcll(hitm); }
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t c = lane < kShadeBlock / 64 ? s_wave_live[lane] : 0u;
        uint32_t h = lane < kShadeBlock / 64 ? s_wave_hits[lane] : 0u;
        // inclusive scan over the first 16 lanes
#pragma unroll
        for (int d = 1; d < kShadeBlock / 64; d <<= 1) {
            uint32_t t = __shfl_up(c, d, 64);
            if ((int)lane >= d) c += t;
        }
#pragma unroll
        for (int d = kShadeBlock / 128; d >= 1; d >>= 1) h += __shfl_down(h, d, 64);
        if (lane < kShadeBlock / 64) s_wave_live[lane] = c;      // inclusive prefix
        if (lane == kShadeBlock / 64 - 1) {
            uint32_t base = c ? atomicAdd(&counters[depth + 1u], c) : 0u;
            s_base = base;
        }
        if (lane == 0 && h) atomicAdd(&counters[kHitCounter], h);
    }
    __syncthreads();
    if (alive) {
        uint32_t before = wave ? s_wave_live[wave - 1] : 0u;
        uint32_t slot = s_base + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        float4 *op = reinterpret_cast<float4 *>(rays_out + slot);
        op[0] = o0; op[1] = o1; op[2] = o2;
    }
}

// ---------------------------------------------------------------- resolve
__global__ void __launch_bounds__(256) k_resolve(FrameDev f, ImagesDev img, const float4 *__restrict__ results,
                                                 uint32_t n_samples)
{
    uint32_t pix = blockIdx.x * 256u + threadIdx.x;
    if (pix >= f.n_pixels) return;
    float4 cur = img.albedo[pix];
    f3 c = mk3(cur.x, cur.y, cur.z);
    for (uint32_t s = 0; s < n_samples; ++s) {
        // rgen.glsl:124-139
        float4 a = results[(uint64_t)s * f.n_pixels + pix];
        float n = (float)(f.mean_base + s);
        f3 scaled_cur = mk3(c.x * n / (n + 1.0f), c.y * n / (n + 1.0f), c.z * n / (n + 1.0f));
        f3 scaled_att = mk3(a.x / (n + 1.0f), a.y / (n + 1.0f), a.z / (n + 1.0f));
        c = scaled_att + scaled_cur;
    }
    img.albedo[pix] = make_float4(c.x, c.y, c.z, 1.0f);                      // :141
}

__global__ void k_accumulate_stats(const uint32_t *__restrict__ counters, uint32_t max_depth, uint64_t n_samples,
                                   uint64_t *__restrict__ totals)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t rays = 0;
        for (uint32_t b = 0; b < max_depth; ++b) rays += counters[b];
        totals[0] += rays;
        totals[1] += counters[kHitCounter];
        totals[2] += n_samples;
    }
}

__global__ void __launch_bounds__(256) k_scale_copy(const float4 *src, float4 *dst, uint64_t n, float scale)   // src may equal dst
{
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 v = src[i];
    dst[i] = make_float4(v.x * scale, v.y * scale, v.z * scale, 1.0f);
}

}  // namespace

void launch_raygen(hipStream_t s, const FrameDev &f, uint32_t n_paths, RayRec *rays, uint32_t *counters)
{
    uint32_t blocks = (n_paths + 255u) / 256u;
    hipLaunchKernelGGL(k_raygen, dim3(blocks ? blocks : 1u), dim3(256), 0, s, f, n_paths, rays, counters);
}

void launch_extend(hipStream_t s, const SceneDev &sc, const FrameDev &f, const RayRec *rays, const uint32_t *count,
                   uint32_t *next, uint32_t capacity, uint32_t max_waves, HitRec *hits, uint64_t *trav_totals)
{
    static const ExtendSched refill = [] {
        auto knob = [](const char *name, int dflt) { const char *e = getenv(name); int v = e ? atoi(e) : dflt; return v < 1 ? 1 : (v > 64 ? 64 : v); };
        ExtendSched k; k.refill = knob("PTX_REFILL", 8); k.node_min = knob("PTX_NODE_MIN", 1); k.tri_min = knob("PTX_TRI_MIN", 1);
        const char *ce = getenv("PTX_CHUNK"); int cv = ce ? atoi(ce) : 128; k.chunk = (uint32_t)(cv < 64 ? 64 : cv);
        return k;
    }();
    uint32_t blocks = std::min((capacity + kExtendBlock - 1u) / kExtendBlock, max_waves);
    if (!blocks) return;
    if (trav_totals)
        hipLaunchKernelGGL(k_extend<true>, dim3(blocks), dim3(kExtendBlock), 0, s, sc, f.t_min, f.t_max, rays, count, next,
                           hits, reinterpret_cast<unsigned long long *>(trav_totals), refill);
    else
        hipLaunchKernelGGL(k_extend<false>, dim3(blocks), dim3(kExtendBlock), 0, s, sc, f.t_min, f.t_max, rays, count, next,
                           hits, (unsigned long long *)nullptr, refill);
}

uint32_t extend_resident_waves(int device)
{
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_extend<false>, kExtendBlock, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    return (uint32_t)per_cu * (uint32_t)cus;
}

void launch_shade(hipStream_t s, const SceneDev &sc, const FrameDev &f, const ImagesDev &img, uint32_t depth,
                  const RayRec *rays_in, const HitRec *hits, uint32_t *counters, uint32_t capacity, RayRec *rays_out,
                  float4 *results)
{
    uint32_t blocks = (capacity + kShadeBlock - 1u) / kShadeBlock;
    if (!blocks) return;
    hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(kShadeBlock), 0, s, sc, f, img, depth, rays_in, hits, counters,
                       rays_out, results);
}

void launch_resolve(hipStream_t s, const FrameDev &f, const ImagesDev &img, const float4 *results, uint32_t n_samples)
{
    uint32_t blocks = (f.n_pixels + 255u) / 256u;
    if (!blocks) return;
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, s, f, img, results, n_samples);
}

void launch_accumulate_stats(hipStream_t s, const uint32_t *counters, uint32_t max_depth, uint64_t n_samples,
                             uint64_t *totals)
{
    hipLaunchKernelGGL(k_accumulate_stats, dim3(1), dim3(64), 0, s, counters, max_depth, n_samples, totals);
}

void launch_scale_copy(hipStream_t s, const float4 *src, float4 *dst, uint64_t n, float scale)
{
    uint32_t blocks = (uint32_t)((n + 255u) / 256u);
    if (!blocks) return;
    hipLaunchKernelGGL(k_scale_copy, dim3(blocks), dim3(256), 0, s, src, dst, n, scale);
}

}  // namespace ptx
