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
// order -- the image is bit-identical to the frame-by-frame loop of oracle/pt_oracle.c, the CPU
// restatement of the reference's shaders (parity with the Vulkan reference itself is unpinned:
// it holds no output fixtures and cannot run here; e.g. it draws a fresh rand::random() seed per
// frame, renderer.rs:1005,1021, where this build takes the caller's).
#include <hip/hip_runtime.h>

#include "ptx_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace ptx {

namespace {

constexpr int kExtendBlock = 64;    // one wave per workgroup: resources free as soon as the wave retires
#ifndef PTX_SHADE_BLOCK
#define PTX_SHADE_BLOCK 512
#endif
constexpr int kShadeBlock = PTX_SHADE_BLOCK;   // one queue-tail atomic per block (a single word takes only ~88 M atomics/s); 512 threads x 99 VGPRs = 4 waves/SIMD (256 and 1024 threads measured slower)
#ifndef PTX_LIT_BLOCK
#define PTX_LIT_BLOCK 256
#endif
constexpr int kLitBlock = PTX_LIT_BLOCK;       // k_shade_lit: a tile lasts as long as its slowest wave (divergent material code), so smaller blocks wait less: 256 threads 7.6 ms of shading per step, 512 threads 8.1, 128 threads 11.0 -- two queue-tail atomics per block and tile

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
__global__ void __launch_bounds__(256) k_raygen(FrameDev f, uint32_t n_paths, RayQueue rays,
                                                uint32_t *__restrict__ counters)
{
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0) {      // the chunk's counters: queue lengths, hit count, and the fetch heads of the bounces this frame can reach
        const uint32_t depth = f.max_depth < 1u ? 1u : (f.max_depth < kMaxDepthLimit ? f.max_depth : kMaxDepthLimit);
        const uint32_t n_clear = fetch_heads(depth, 0);
        for (uint32_t i = threadIdx.x; i < n_clear; i += 256u) counters[i] = i == 0 ? n_paths : 0u;
    }
    if (p >= n_paths) return;
    uint32_t pix = p % f.n_pixels;
    uint32_t y = pix / f.width, x = pix - y * f.width;
    // rgen.glsl:25-28
    float pcx = (float)x + 0.5f, pcy = (float)(f.height - y - 1u) + 0.5f;
    float uvx = pcx / (float)f.width, uvy = pcy / (float)f.height;
    float dx = uvx * 2.0f - 1.0f, dy = uvy * 2.0f - 1.0f;
    // rgen.glsl:30-34
    f3 target = mat4_mul(f.cam.proj_inverse, dx, dy, 1.0f, 1.0f);
    f3 tn = normalize(target);
    f3 dir = normalize(mat4_mul(f.cam.view_inverse, tn.x, tn.y, tn.z, 0.0f));

    // the origin plane (camera position, path index = queue index) is not written either: bounce 0 derives it
    rays.dir[p] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(hash2(f2u(uvx), f2u(uvy))));
    // rgen.glsl:62: the throughput of a camera ray is (1, 1, 1) -- not written; the bounce-0 shading stages know it
}

// ---------------------------------------------------------------- extend
// Persistent one-wave workgroups walk the BVH with 64 independent per-lane state machines.
//
//  * Dynamic ray fetch: whenever `refill` or more lanes have finished, the wave refills them from
//    the bounce's ray queue.  Path lengths through the BVH are heavy-tailed; without the refill a
//    wave runs at the pace of its slowest ray (11 % VALU lane utilisation measured, profiles/r01a).
//    Rays are handed out in two levels -- a wave reserves a chunk of the queue with one atomic and
//    refills from that chunk -- because one atomic per refill on a single word saturates at
//    ~88 M/s chip-wide (MI355X_MICROARCH.md "dequeue"); for the same reason the queue has eight
//    head words, one per XCD-affine segment, with stealing between them.
//  * Wave-level triangle queue: a lane that hits leaves does not test their triangles itself (the
//    triangle code would run for the ~20 of 64 lanes that happen to have one pending, r01e).  The
//    triangles go to a per-wave FIFO in LDS as (triangle, owner lane) pairs and the owner carries
//    on down the tree; when enough pairs have collected (36), ANY lane tests ANY pair (the owner's ray --
//    origin, direction, current closest t -- read from the wave's LDS block) and successful tests are
//    handed back to their owners with an LDS atomic minimum.  Closest
//    hit with the (instance, primitive) tie-break is order-independent, so the image is unchanged;
//    a ray is complete once its node walk is done and the FIFO head has passed its last pair.
//  * Nodes are loaded straight into registers; what pays is occupancy without register spills -- 8 waves/SIMD
//    at 64 VGPRs and 5.0 KB of LDS per wave (the 27 overflow entries of the traversal stack are a private
//    array: 224 B of scratch per lane, touched only by rays that nest deeper than 5 groups).  At that
//    occupancy the kernel uses 62 % of the SIMDs' vector issue capacity (profiles/valu.json, priced with
//    the per-class issue costs of profiles/r02_valu_calibration.json); 30 % of a wave's life is waiting
//    for memory, 43 % waiting for an issue slot.
// Traversal stack of one lane: the first kLdsStack entries in the wave's LDS block (column
// `lane`), the rest in a private array.  The private array is referenced through a pointer so that
// `sp` and the LDS column stay in registers (a struct holding the array itself is demoted to
// scratch as a whole, stack pointer included).
typedef __attribute__((address_space(3))) uint2 lds_uint2;
typedef __attribute__((address_space(5))) uint2 prv_uint2;
struct LaneStack {
    lds_uint2 *lds;             // this lane's column of the workgroup's LDS stack
    prv_uint2 *ovf;             // kOvfStack private entries
    int sp;
    __device__ __forceinline__ void clear() { sp = 0; }
    __device__ __forceinline__ void push(uint32_t base, uint32_t mask)
    {
        if (__builtin_expect(sp < kLdsStack, 1)) { lds[sp * kExtendBlock].x = base; lds[sp * kExtendBlock].y = mask; }
        else { ovf[sp - kLdsStack].x = base; ovf[sp - kLdsStack].y = mask; }
        ++sp;
    }
    __device__ __forceinline__ void pop(uint32_t &base, uint32_t &mask)
    {
        --sp;
        if (__builtin_expect(sp < kLdsStack, 1)) { base = lds[sp * kExtendBlock].x; mask = lds[sp * kExtendBlock].y; }
        else { base = ovf[sp - kLdsStack].x; mask = ovf[sp - kLdsStack].y; }
    }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
};

// wave-wide maximum (instrumented build only): every lane ends up with the value of lane 63 after an inclusive max scan
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = o > v ? o : v; }
    return v;
}

// __ballot() takes an int: the bool is widened and compared again (v_cndmask + v_cmp per call in the traversal loop)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// After a node test: make the next node group current and take its nearest pending child -- Traversal::next_group +
// node_pick written without divergent regions (the probe sweep of tools/probe_sweep.sh prices a scalar instruction
// of this kernel at 0.13 % of its run time, a vector one at 0.19 - 0.26 %; the branchy form spent ~25 scalar
// instructions per visit on EXEC bookkeeping).  Every lane reads the top entry of its LDS stack and writes the slot
// above it whether or not it pops / pushes; selects keep what applies.  Lanes whose stack reaches into the private
// overflow array are rare: then the whole wave takes the general path.  Returns the node index, 0xFFFFFFFF: walk over.
template <class Trav>
__device__ __forceinline__ uint32_t advance_walk(Trav &tr, LaneStack &st)
{
    if (__builtin_expect(ballot64(st.sp >= kLdsStack) != 0ull, 0)) return tr.next_group(st) ? tr.node_pick(st) : 0xFFFFFFFFu;
    const bool has = (tr.ng_mask & 0xFF000000u) != 0u;
    const int top_i = st.sp > 0 ? st.sp - 1 : 0;
    const uint32_t top_base = st.lds[top_i * kExtendBlock].x, top_mask = st.lds[top_i * kExtendBlock].y;
    const bool pop = !has && st.sp > 0;
    const uint32_t base = pop ? top_base : tr.ng_base;
    const uint32_t mask = pop ? top_mask : tr.ng_mask;
    const int sp = st.sp - (pop ? 1 : 0);
    const bool ok = has || pop;
    const uint32_t b = msb32(mask | 1u);                     // the nearest pending child (garbage when !ok: discarded below)
    const uint32_t rest = mask & ~(1u << b);
    const bool push = ok && (rest & 0xFF000000u) != 0u;
    st.lds[sp * kExtendBlock].x = base; st.lds[sp * kExtendBlock].y = rest;      // sp <= kLdsStack - 1: the free slot above the top
    st.sp = sp + (push ? 1 : 0);
    const uint32_t slot = (b - 24u) ^ tr.octinv;
    const uint32_t rel = (uint32_t)__popc(mask & 0xFFu & ((1u << (slot & 31u)) - 1u));
    tr.ng_base = ok ? base : tr.ng_base;
    tr.ng_mask = ok ? rest : tr.ng_mask;
    return ok ? base + rel : 0xFFFFFFFFu;
}

// wave-uniform scheduling thresholds (lane counts) and the ray-fetch chunk
struct ExtendSched {
    int refill, tri_min, flush_min; uint32_t chunk, chunk_grow;
    uint32_t primary;       // camera rays: origin = cam_o, path index = queue index (k_raygen stores neither)
    uint32_t seed_t;        // PTX_LAB_SEED builds only (tools/lab/seed_bound.sh): start every ray with the closest t of a previous pass
    float cam_o[3];
};

// 8 waves per SIMD: 64 VGPRs (no register spills) and <= 5120 B of LDS per one-wave workgroup (160 KB / 32).  Round 1 ran
// 7 (the packed-FMA node test needed 68-72 registers); with the rebalanced node test the eighth wave is worth 3 %.
#ifndef PTX_EXTEND_WAVES_PER_EU
#define PTX_EXTEND_WAVES_PER_EU 8
#endif

constexpr uint32_t kTriQueue = 128;   // entries, power of two
static_assert(kTriIndexBits == 26, "a queue entry is 26 bits of triangle index + 6 bits of owner lane");

__device__ __forceinline__ void wave_lds_sync()
{
    // one wave per workgroup and DS operations of a wave complete in order: only the compiler
    // needs to be kept from moving LDS traffic across this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive prefix sum over the 64 lanes: four row_shr steps inside each row of 16, then the two
// row broadcasts of gfx9 DPP.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
    return (uint32_t)x;
}

//  * kAnyHit (occlusion rays of the lit integrator): the first accepted triangle ends the ray's walk;
//    hits[ray].tri != miss then only means "occluded".
// pt_math.h tri_test_t with the same expression trees and the same accept / reject decisions, evaluated without early
// exits: in a block of ~59 pairs some lane passes the edge test practically always, so the division ran anyway, and each
// early exit cost a handful of scalar EXEC instructions per block.  `ht`: the owner's current closest t.
__device__ __forceinline__ bool tri_test_flat(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float t_min, float t_max, float ht, float &t_out)
{
    const f3 A = v0 - o, B = v1 - o, C = v2 - o;
    const float U = dot(d, cross(C - B, C + B));
    const float V = dot(d, cross(A - C, A + C));
    const float W = dot(d, cross(B - A, B + A));
    const bool mixed = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
    const float det = (U + V) + W;
    const f3 Ng = cross(B - A, C - A);
    const float den = dot(d, Ng);
    const float t = dot(A, Ng) / den;
    t_out = t;
    return !mixed & (det != 0.0f) & (t > t_min) & (t < t_max) & (t <= ht);
}

constexpr uint32_t kWalked = 0xFFFFFFFFu, kIdle = 0xFFFFFFFEu, kNoRay = 0xFFFFFFFFu;
template <bool kStats, bool kAnyHit>
__global__ void __launch_bounds__(kExtendBlock) __attribute__((amdgpu_waves_per_eu(PTX_EXTEND_WAVES_PER_EU, PTX_EXTEND_WAVES_PER_EU)))
k_extend(SceneDev sc, float t_min, float t_max, RayQueue rays, const uint32_t *__restrict__ count,
           uint32_t *__restrict__ next, HitOut *__restrict__ hits, unsigned long long *__restrict__ trav_totals,
           ExtendSched sched)
{
    __shared__ uint2 s_stack[kLdsStack * kExtendBlock];
    __shared__ uint32_t s_qtri[kTriQueue];      // pending (triangle, owner lane) pairs: triangle index | owner << kTriIndexBits
    __shared__ unsigned long long s_best[kExtendBlock];   // closest hit of each lane's ray as one ordered word: float bits of t << 32 | TriRec::rank
    __shared__ float s_dir[3][kExtendBlock];      // direction of each lane's ray: the triangle block reads its pair owner's
    __shared__ float s_org[3][kExtendBlock];      // ... and its origin
    const uint32_t n = *count;
    const uint32_t lane_id = threadIdx.x;
    uint2 ovf_store[kOvfStack];
    LaneStack st;
    st.lds = (lds_uint2 *)s_stack + threadIdx.x;
    st.ovf = (prv_uint2 *)ovf_store;
    st.sp = 0;
    Traversal<LaneStack, kStats> tr;
    tr.ng_mask = 0; tr.tg_mask = 0;
    tr.o = mk3(0.0f, 0.0f, 0.0f); tr.hit.t = 0.0f;
    tr.t_min = t_min; tr.t_max = t_max;     // wave-uniform: keeps them out of the VGPRs
    // Lane state without separate flags (r03): a boolean carried round the loop lives in an SGPR pair, and every merge
    // of it across a divergent region costs the compiler three scalar mask operations, every ballot of it a v_cndmask +
    // v_cmp.  next_ni says it all: a node index (the walk goes on), kWalked (the walk is over, the ray waits for its last
    // queued triangles) or kIdle (no live ray).  ray != kNoRay on an idle lane: its finished ray has not been stored yet --
    // all such lanes store together at the next refill.
    uint32_t ray = kNoRay;
    uint32_t next_ni = kIdle;
    uint32_t q_head = 0, q_tail = 0;    // wave-uniform, monotone (mod 2^32)
    uint32_t my_tail = 0;               // q_tail just after this lane's latest pair
    unsigned long long tot_nodes = 0, tot_tris = 0;
    // instrumented build only: wave lifetime per phase (s_memtime deltas; a stamp first drains the wave's memory
    // counters, so "wait" phases hold exactly the time until the data was there)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = kStats ? __builtin_amdgcn_s_memtime() : 0ull;
#define PTX_PHASE(i)                                                                     \
    do {                                                                                 \
        if (kStats) {                                                                    \
            __builtin_amdgcn_s_waitcnt(0);                                               \
            unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
            ph[i] += now_ - ph_t; ph_t = now_;                                           \
        }                                                                                \
    } while (0)
    unsigned long long w_iters = 0, w_node_blocks = 0, w_node_lanes = 0, w_tri_blocks = 0, w_tri_lanes = 0, w_refills = 0;
    unsigned long long w_append_blocks = 0, w_append_passes = 0, w_refill_passes = 0;
    // The "[phase: ...]" comments below are read by tools/isa_budget.py: with the line table of the compiled kernel they
    // assign every instruction of the loop to a phase, and the counters of the instrumented build weight the phases.
    uint32_t chunk_next = 0, chunk_end = 0;
    // Rays are handed out in chunks, one atomic each.  A single word takes ~88 M atomics/s, so a long queue of cheap
    // rays (33 M primary rays that mostly miss: 260 k chunks of 128) is bound by that rate alone -- 3.1 ms measured
    // where the rays themselves need 1 ms; a short queue wants small chunks to keep the waves balanced.  Hence the
    // chunk grows with the queue of camera rays: about 8 chunks per wave, between sched.chunk (128) and 4 x sched.chunk.
    uint32_t chunk = sched.chunk;
    {
        const uint32_t per_wave = n / (gridDim.x * 8u);
        const uint32_t cap = sched.chunk * sched.chunk_grow;
        if (per_wave > chunk) chunk = (per_wave < cap ? per_wave : cap) & ~63u;
    }
    // The queue is cut into kFetchHeads segments with a head word each; this wave starts on the segment of its XCD
    // (workgroups are dealt round-robin over the 8 XCDs: speed only, nothing depends on it) and goes round the
    // others when a segment is empty.  `dead`: segments seen empty (wave-uniform).
    const uint32_t seg = ((n + kFetchHeads - 1u) / kFetchHeads + 63u) & ~63u;
    uint32_t cur_seg = blockIdx.x & (kFetchHeads - 1u);
    uint32_t dead = 0;
    // wave-uniform: the queue has been handed out completely.  At most ceil(seg / chunk) waves per segment can
    // ever receive rays: on a short queue (late bounces, small frames) the rest of the grid starts out
    // "exhausted" and leaves without paying an atomic.
    bool exhausted = (uint64_t)(blockIdx.x / kFetchHeads) * chunk >= seg || n == 0;

    for (;;) {
        // [phase: refill]
        unsigned long long idle = ballot64(next_ni == kIdle);
        if (!exhausted && __popcll(idle) >= sched.refill) {
            if (kStats) w_refills++;
            // The results of the rays that finished since the last refill leave in ONE store here.  A store per
            // iteration (a few lanes retire in almost every one) sat in front of every node fetch: vmcnt counts
            // loads and stores together and in order, so each fetch also waited for the store before it.  The closest
            // hit was kept by rank; the stages downstream index triangles in leaf order (an occlusion query only
            // reports hit / miss): that gather goes out first and comes back behind the new rays' loads for free.
            const bool pending = next_ni == kIdle && ray != kNoRay;
            const uint32_t old_ray = ray;
            const float old_t = tr.hit.t;
            uint32_t old_tri = tr.hit.tri;
            if (!kAnyHit && pending && old_tri != 0xFFFFFFFFu) old_tri = sc.tri_of_rank[old_tri];
            if (pending) ray = kNoRay;
            for (int pass = 0; pass < 2 && idle != 0ull; ++pass) {
                if (kStats) w_refill_passes++;
                if (chunk_next == chunk_end) {
                    bool got = false;
                    while (dead != (1u << kFetchHeads) - 1u) {
                        const uint32_t lo = cur_seg * seg;
                        if (!(dead & (1u << cur_seg)) && lo < n) {
                            const uint32_t len = (n - lo < seg) ? n - lo : seg;
                            uint32_t off = 0;
                            if (threadIdx.x == 0) off = atomicAdd(next + cur_seg * kHeadStride, chunk);
                            off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
                            if (off < len) {
                                chunk_next = lo + off;
                                chunk_end = (len - off < chunk) ? lo + len : lo + off + chunk;
                                got = true;
                                break;
                            }
                        }
                        dead |= 1u << cur_seg;
                        cur_seg = (cur_seg + 1u) & (kFetchHeads - 1u);
                    }
                    if (!got) { exhausted = true; break; }
                }
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                uint32_t avail = chunk_end - chunk_next;
                if (next_ni == kIdle && rank < avail) {
                    uint32_t i = chunk_next + rank;
                    const float4 r1 = rays.dir[i];
                    const float4 r0 = sched.primary ? make_float4(sched.cam_o[0], sched.cam_o[1], sched.cam_o[2], __uint_as_float(i)) : rays.org[i];
                    // 1 / direction by v_rcp_f32 (1 ulp): the box tests only have to be conservative, and the padding of
                    // the triangle boxes (bvh_build.cpp) is orders of magnitude above that
                    float ix, iy, iz;
                    const uint32_t oct = ray_slab_setup_rcp(mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), ix, iy, iz);
                    bool go = tr.init_prepared(sc.n_nodes, mk3(r0.x, r0.y, r0.z), ix, iy, iz, oct, t_min, t_max, st);
                    next_ni = go ? tr.node_pick(st) : kWalked;
                    ray = i;
                    my_tail = q_head;
                    s_dir[0][lane_id] = r1.x; s_dir[1][lane_id] = r1.y; s_dir[2][lane_id] = r1.z;
                    s_org[0][lane_id] = r0.x; s_org[1][lane_id] = r0.y; s_org[2][lane_id] = r0.z;
                    s_best[lane_id] = ((unsigned long long)__float_as_uint(t_max) << 32) | 0xFFFFFFFFull;
#ifdef PTX_LAB_SEED
                    // developer experiment: an upper bound for ANY culling / ordering improvement -- the ray starts with the
                    // distance of its closest hit (from a first, ordinary pass over the same queue) as its limit
                    if (sched.seed_t && hits[i].tri != 0xFFFFFFFFu) {
                        const float lim = hits[i].t * (1.0f + 0x1p-20f);
                        tr.hit.t = lim;
                        s_best[lane_id] = ((unsigned long long)__float_as_uint(lim) << 32) | 0xFFFFFFFFull;
                    }
#endif
                }
                uint32_t n_idle = (uint32_t)__popcll(idle);
                chunk_next += (n_idle < avail) ? n_idle : avail;
                idle = ballot64(next_ni == kIdle);
            }
            if (pending) *reinterpret_cast<float2 *>(hits + old_ray) = make_float2(old_t, __uint_as_float(old_tri));
        }
        PTX_PHASE(0);
        if (ballot64(next_ni != kIdle) == 0) {
            if (exhausted) break;
            continue;
        }
        const int stop_at = exhausted ? 0 : (64 - sched.refill);
        do {
            // [phase: node]
            // ---- node visits: every lane whose walk is not finished and whose last leaf hits are queued
            // (a ballot per comparison, combined as scalars: the ballot of a conjunction is widened to a VGPR and compared
            // again, v_cndmask + v_cmp, because the compiler cannot see that the AND of two v_cmp results lies inside EXEC)
            const bool no_tris = tr.tg_mask == 0, walking = next_ni < kIdle;
            const bool want_node = no_tris && walking;
            if (kStats) {
                const int nb = __popcll(ballot64(no_tris) & ballot64(walking));
                w_iters++;
                if (nb) { w_node_blocks++; w_node_lanes += nb; }
                Words80 nw;
                if (want_node) nw = load_node(sc.nodes, next_ni);
                PTX_PHASE(1);
                if (want_node) {
                    tr.node_eval(nw);
                    next_ni = advance_walk(tr, st);
                }
                PTX_PHASE(2);
            } else if (want_node) {      // (no wave-uniform guard: an empty EXEC skips the block by itself)
                tr.node_eval(load_node(sc.nodes, next_ni));
                next_ni = advance_walk(tr, st);
            }
            // [phase: append]
            // ---- queue the triangles of the leaves just hit: wave prefix sum of the per-lane counts
            //      (six DPP adds), then every lane writes its own run of the FIFO
            if (ballot64(tr.tg_mask != 0) != 0ull) {
                if (kStats) w_append_blocks++;
                const uint32_t room = kTriQueue - (q_tail - q_head);
                const uint32_t cnt = (uint32_t)__popc(tr.tg_mask);
                const uint32_t incl = wave_inclusive_sum(cnt);
                // lanes whose whole group fits form a prefix (incl is monotone); the others keep
                // their group and stay out of the node block until the FIFO has drained
                const bool fits = incl <= room;
                const uint32_t n_fit = (uint32_t)__popcll(ballot64(fits));
                const uint32_t total = n_fit ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)n_fit - 1) : 0u;
                if (fits && cnt) {
                    my_tail = q_tail + incl;
                    // the loop runs max-over-lanes(count) times per wave iteration (4.4 on the Sponza stand-in: 12 % of all
                    // issued instructions, profiles/r03_isa_budget.json), so its body is kept to the bit scan, the entry, the
                    // byte address of the ring slot and the store
                    uint32_t byte_at = ((q_tail + incl - cnt) & (kTriQueue - 1u)) << 2;
                    const uint32_t entry0 = tr.tg_base | (lane_id << kTriIndexBits);      // triangle indices stay below 2^26: + bit never carries into the lane
                    uint32_t bits = tr.tg_mask;
                    do {
                        // [phase: append-loop]
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(s_qtri) + byte_at) = entry0 + lsb32(bits);
                        bits &= bits - 1u;
                        byte_at = (byte_at + 4u) & ((kTriQueue - 1u) << 2);
                    } while (bits != 0);
                    tr.tg_mask = 0;
                }
                // [phase: append]
                if (kStats) w_append_passes += (unsigned long long)__builtin_amdgcn_readfirstlane((int)wave_max_u32(fits ? cnt : 0u));
                q_tail += total;
            }
            PTX_PHASE(3);
            // [phase: tri-decision]
            // ---- triangle tests, a full wave's worth at a time
            {
                const uint32_t qn = q_tail - q_head;
                const unsigned long long m_no_tris = ballot64(tr.tg_mask == 0);
                const int n_wait = __popcll(ballot64(next_ni == kWalked) & m_no_tris & ballot64((int)(q_head - my_tail) < 0));
                const int n_node = __popcll(ballot64(next_ni < kIdle) & m_no_tris);
                if (qn >= (uint32_t)sched.tri_min || (qn > 0u && (n_wait >= sched.flush_min || n_node == 0))) {
                    // [phase: tri-block]
                    wave_lds_sync();
                    const uint32_t nt = qn < 64u ? qn : 64u;
                    if (kStats) { w_tri_blocks++; w_tri_lanes += nt; }
                    const bool has = lane_id < nt;
                    // One pair per lane.  Everything about the owner's ray comes out of the wave's LDS block -- origin,
                    // direction, and the current closest t from s_best[own], the master copy of (t, rank) that tr.hit only
                    // mirrors -- so the whole test sits under ONE predicate (r02 fetched origin and t with four ds_bpermute,
                    // which every lane has to execute, and so split the block in two with 14 registers zeroed in between).
                    uint32_t own = lane_id;
                    float t = 0.0f, rank_bits = 0.0f;
                    bool cand = false;
                    f3 o, dd; float ht; float4 r0, r1, r2;
                    auto fetch = [&]() {
                        const uint32_t entry = s_qtri[(q_head + lane_id) & (kTriQueue - 1u)];
                        const uint32_t ti = entry & ((1u << kTriIndexBits) - 1u);
                        own = entry >> kTriIndexBits;
                        const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + ti);
                        r0 = tp[0]; r1 = tp[1]; r2 = tp[2];
                        o = mk3(s_org[0][own], s_org[1][own], s_org[2][own]);
                        dd = mk3(s_dir[0][own], s_dir[1][own], s_dir[2][own]);
                        ht = __uint_as_float(reinterpret_cast<const uint32_t *>(s_best)[own * 2u + 1u]);
                    };
                    auto test = [&]() {
                        cand = tri_test_flat(o, dd, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), mk3(r2.x, r2.y, r2.z), t_min, t_max, ht, t);
                        rank_bits = r2.w;
                        if (kStats) tot_tris++;
                    };
                    if (kStats) {
                        if (has) fetch();
                        PTX_PHASE(4);
                        if (has) test();
                        PTX_PHASE(5);
                    } else if (has) {
                        fetch();
                        test();
                    }
                    q_head += nt;
                    // Hand the successful tests to their owners: one LDS atomic minimum on the owner's (t, rank) word.  t is
                    // positive, so its float bits order like the value; equal distances (shared edges, coincident surfaces)
                    // fall through to the rank = lowest (instance, primitive) wins, the oracle's rule; several pairs of one
                    // owner, in any order, leave the same minimum.  Then every lane reloads its own word.
                    if (cand) atomicMin(&s_best[own], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)__float_as_uint(rank_bits));
                    wave_lds_sync();
                    {
                        // (every lane: the word of an idle lane has not changed since its ray retired)
                        const unsigned long long best = s_best[lane_id];
                        const uint32_t best_rank = (uint32_t)best;
                        tr.hit.t = __uint_as_float((uint32_t)(best >> 32)); tr.hit.tri = best_rank;
                        if (kAnyHit && next_ni != kIdle && best_rank != 0xFFFFFFFFu) { next_ni = kWalked; st.sp = 0; tr.ng_mask = 0; }
                    }
                    PTX_PHASE(6);
                }
            }
            // [phase: retire]
            // ---- rays whose walk is over and whose pairs have all been tested
            {
                const bool retire = next_ni == kWalked && tr.tg_mask == 0 && (int)(q_head - my_tail) >= 0;
                if (kStats && retire) tot_nodes += tr.stats.nodes;
                next_ni = retire ? kIdle : next_ni;      // (t, rank) stay in the registers until the next refill stores them
            }
            PTX_PHASE(7);
        } while (__popcll(ballot64(next_ni != kIdle)) > stop_at);
    }
    if (next_ni == kIdle && ray != kNoRay) {
        uint32_t tri = tr.hit.tri;
        if (!kAnyHit && tri != 0xFFFFFFFFu) tri = sc.tri_of_rank[tri];
        *reinterpret_cast<float2 *>(hits + ray) = make_float2(tr.hit.t, __uint_as_float(tri));
    }
    if (kStats) {
        atomicAdd(&trav_totals[0], tot_nodes);
        atomicAdd(&trav_totals[1], tot_tris);
        if (threadIdx.x == 0) {
            atomicAdd(&trav_totals[2], w_iters); atomicAdd(&trav_totals[3], w_node_blocks); atomicAdd(&trav_totals[4], w_node_lanes);
            atomicAdd(&trav_totals[5], w_tri_blocks); atomicAdd(&trav_totals[6], w_tri_lanes); atomicAdd(&trav_totals[7], w_refills);
            for (int i = 0; i < 8; ++i) atomicAdd(&trav_totals[9 + i], ph[i]);      // totals[12 + i]: trav_totals = totals + 3
            atomicAdd(&trav_totals[17], w_append_blocks); atomicAdd(&trav_totals[18], w_append_passes); atomicAdd(&trav_totals[19], w_refill_passes);
        }
#undef PTX_PHASE
    }
}

// ---------------------------------------------------------------- shade
__device__ __forceinline__ uint32_t texel_raw(const TexDesc &tx, uint32_t x, uint32_t y)
{
    // texels are stored in tiles of 8 x 4 (one 128-byte line, see TexDesc): same texel, other address
    const uint32_t at = ((y >> 2) * tx.tiles_x + (x >> 3)) * 32u + ((y & 3u) << 3) + (x & 7u);
    return *reinterpret_cast<const uint32_t *>(tx.texels + (uint64_t)at * 4u);
}
__device__ __forceinline__ f3 texel_decode(uint32_t texel, uint32_t flags, const float *lut)
{
    uint32_t r = texel & 0xFFu, g = (texel >> 8) & 0xFFu, b = (texel >> 16) & 0xFFu;
    if (flags & 1u) return mk3(lut[r], lut[g], lut[b]);
    return mk3((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f);
}

// PTX_TEXTURE_LINEAR: texel centres at i + 0.5.  linear_axis_k0: the lower texel index of the pair, in [-1, size], and
// the blend weight; axis_pair: the two texel indices with wrap / clamp applied.  (Two functions because the texel sets
// of the lit integrator tabulate axis_pair for every k0: their builder and the plain fetch share this code.)
__device__ __forceinline__ int32_t linear_axis_k0(float u, uint32_t size, float &a)
{
    float x = u * (float)size - 0.5f;
    float xf = floorf(x);
    float fr = x - xf;
    if (!(fr >= 0.0f)) fr = 0.0f;
    if (fr > 1.0f) fr = 1.0f;
    float hi = (float)size;
    if (!(xf >= -1.0f)) xf = -1.0f;
    if (xf > hi) xf = hi;
    a = fr;
    return (int32_t)xf;
}
__device__ __forceinline__ void axis_pair(int32_t k0, uint32_t size, bool repeat, uint32_t &i0, uint32_t &i1)
{
    int32_t k1 = k0 + 1, n = (int32_t)size;
    if (repeat) {
        if (k0 < 0) k0 += n;
        if (k0 >= n) k0 -= n;
        if (k1 >= n) k1 -= n;
        if (k1 >= n) k1 -= n;
    } else {
        if (k0 < 0) k0 = 0;
        if (k0 > n - 1) k0 = n - 1;
        if (k1 > n - 1) k1 = n - 1;
    }
    i0 = (uint32_t)k0; i1 = (uint32_t)k1;
}
__device__ __forceinline__ void linear_axis(float u, uint32_t size, bool repeat, uint32_t &i0, uint32_t &i1, float &a)
{
    axis_pair(linear_axis_k0(u, size, a), size, repeat, i0, i1);
}

// A texture fetch in two halves, so that a shading stage can put the loads of several fetches in flight together and
// decode them where their values are needed (the lit integrator takes up to three taps per hit -- base colour,
// metallic-roughness, normal -- whose texels are independent of each other but sat in three conditional blocks, i.e.
// three serial round trips to a 100 MB working set).  tap_issue: addressing + the raw texel loads; tap_resolve: sRGB /
// linear decode and the bilinear blend.  sample_texture = resolve(issue): one set of expression trees.
struct TexTap {
    uint32_t t00, t10, t01, t11;    // raw RGBA8 texels (nearest: t00 only)
    float ax, ay;                   // bilinear weights
    uint32_t flags;                 // the texture's flag word (bit0 sRGB, bit2 bilinear); 0xFFFFFFFF: no tap issued
};
__device__ __forceinline__ TexTap tap_issue(const TexDesc &tx, float u, float v)
{
    TexTap t; t.flags = tx.srgb; t.ax = 0.0f; t.ay = 0.0f; t.t10 = t.t01 = t.t11 = 0u;
    // vulkano SamplerCreateInfo::default(): nearest, clamp-to-edge, LOD 0 (graphics.rs:453-459)
    if (tx.srgb & 2u) { u = u - floorf(u); v = v - floorf(v); }   // build extension: wrap = repeat
    if (tx.srgb & 4u) {                                           // build extension: bilinear, blended after the sRGB decode
        uint32_t x0, x1, y0, y1;
        linear_axis(u, tx.width, (tx.srgb & 2u) != 0u, x0, x1, t.ax);
        linear_axis(v, tx.height, (tx.srgb & 2u) != 0u, y0, y1, t.ay);
        t.t00 = texel_raw(tx, x0, y0); t.t10 = texel_raw(tx, x1, y0); t.t01 = texel_raw(tx, x0, y1); t.t11 = texel_raw(tx, x1, y1);
        return t;
    }
    float fu = floorf(u * (float)tx.width), fv = floorf(v * (float)tx.height);
    float mu = (float)(tx.width - 1u), mv = (float)(tx.height - 1u);
    if (!(fu >= 0.0f)) fu = 0.0f;
    if (fu > mu) fu = mu;
    if (!(fv >= 0.0f)) fv = 0.0f;
    if (fv > mv) fv = mv;
    t.t00 = texel_raw(tx, (uint32_t)fu, (uint32_t)fv);
    return t;
}
__device__ __forceinline__ f3 tap_resolve(const TexTap &t, const float *lut)
{
    if (t.flags & 4u) {
        f3 c00 = texel_decode(t.t00, t.flags, lut), c10 = texel_decode(t.t10, t.flags, lut);
        f3 c01 = texel_decode(t.t01, t.flags, lut), c11 = texel_decode(t.t11, t.flags, lut);
        float bx = 1.0f - t.ax, by = 1.0f - t.ay;
        f3 top = c00 * bx + c10 * t.ax;
        f3 bot = c01 * bx + c11 * t.ax;
        return top * by + bot * t.ay;
    }
    return texel_decode(t.t00, t.flags, lut);
}
__device__ __forceinline__ f3 sample_texture(const TexDesc &tx, const float *lut, float u, float v)
{
    return tap_resolve(tap_issue(tx, u, v), lut);
}

#ifdef PTX_SHADE_WAVES
__attribute__((amdgpu_waves_per_eu(PTX_SHADE_WAVES, PTX_SHADE_WAVES)))
#endif
__global__ void __launch_bounds__(kShadeBlock) k_shade(SceneDev sc, FrameDev f, ImagesDev img, uint32_t depth,
                                                        RayQueue rays_in,
                                                        const HitOut *__restrict__ hits,
                                                        uint32_t *__restrict__ counters, uint32_t capacity,
                                                        RayQueue rays_out, float4 *__restrict__ results)
{
    __shared__ float s_lut[256];
    __shared__ TexDesc s_tex[kLdsTextures];
    __shared__ uint32_t s_wave_live[kShadeBlock / 64], s_wave_hits[kShadeBlock / 64];
    __shared__ uint32_t s_base;

    // The grid is capped (launch_shade): a block walks tiles of kShadeBlock rays until the live queue ends, so a
    // launch over a nearly empty queue (late bounces) costs a few thousand block starts, not capacity / 512 of them.
    const uint32_t n_live = counters[depth];
    if ((uint64_t)blockIdx.x * kShadeBlock >= n_live) return;
    if (threadIdx.x < 256) s_lut[threadIdx.x] = sc.srgb_lut[threadIdx.x];
    {
        const uint32_t nw = (sc.n_textures < kLdsTextures ? sc.n_textures : kLdsTextures) * (uint32_t)(sizeof(TexDesc) / 4);
        for (uint32_t w = threadIdx.x; w < nw; w += kShadeBlock)
            reinterpret_cast<uint32_t *>(s_tex)[w] = reinterpret_cast<const uint32_t *>(sc.textures)[w];
    }
    // The tile loop is software-pipelined.  A tile costs three dependent memory round trips -- ray + hit, the
    // triangle's record (its index comes with the hit), the block's queue-tail atomic -- and the stage spends 71 % of
    // its wave time waiting for them (profiles/valu.json).  So the NEXT tile's ray + hit are fetched before this tile
    // is shaded, and its record is fetched behind this tile's atomic, before anybody waits for that: per tile one
    // round trip and the arithmetic remain on the critical path.
    const float4 kZero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto fetch_ray = [&](uint32_t tile_, float4 &a0, float4 &a1, float4 &a2, float2 &h) {
        const uint32_t i_ = tile_ * kShadeBlock + threadIdx.x;
        a0 = kZero4; a1 = kZero4; a2 = kZero4; h = make_float2(0.0f, __uint_as_float(0xFFFFFFFFu));
        if ((uint64_t)tile_ * kShadeBlock < n_live && i_ < capacity) {
            a1 = rays_in.dir[i_];
            // camera rays: origin, path index (= queue index) and unit throughput (rgen.glsl:62) are never stored
            a0 = depth == 0 ? make_float4(f.cam_origin[0], f.cam_origin[1], f.cam_origin[2], __uint_as_float(i_)) : rays_in.org[i_];
            a2 = depth == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : rays_in.thr[i_];
            h = *reinterpret_cast<const float2 *>(hits + i_);
        }
    };
    auto tri_of = [&](uint32_t tile_, const float2 &h) -> uint32_t {
        const uint32_t i_ = tile_ * kShadeBlock + threadIdx.x;
        return (i_ < capacity && i_ < n_live) ? __float_as_uint(h.y) : 0xFFFFFFFFu;
    };
    float4 r0, r1, r2; float2 hr;
    fetch_ray(blockIdx.x, r0, r1, r2, hr);
    float4 t0 = kZero4, t1 = kZero4, t2 = kZero4, s0 = kZero4, s1 = kZero4, s2 = kZero4, s3 = kZero4, s4 = kZero4;
    {
        const uint32_t tri0 = tri_of(blockIdx.x, hr);
        if (tri0 != 0xFFFFFFFFu) {
            const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri0);
            t0 = sp[0]; t1 = sp[1]; t2 = sp[2]; s0 = sp[3]; s1 = sp[4]; s2 = sp[5]; s3 = sp[6]; s4 = sp[7];
        }
    }
    __syncthreads();      // the tables are in LDS
    for (uint32_t tile = blockIdx.x; (uint64_t)tile * kShadeBlock < n_live; tile += gridDim.x) {
    const uint32_t i = tile * kShadeBlock + threadIdx.x;
    const bool valid = i < capacity && i < n_live;
    const uint32_t tri = valid ? __float_as_uint(hr.y) : 0xFFFFFFFFu;
    // the next tile's ray and hit, in flight while this tile is shaded
    float4 nr0, nr1, nr2; float2 nhr;
    fetch_ray(tile + gridDim.x, nr0, nr1, nr2, nhr);

    bool alive = false, was_hit = false;
    float4 o0, o1, o2;
    if (valid) {
        f3 ro = mk3(r0.x, r0.y, r0.z), rd = mk3(r1.x, r1.y, r1.z);
        uint32_t path = __float_as_uint(r0.w), uv_state = __float_as_uint(r1.w);
        uint32_t sample = path / f.n_pixels, pix = path - sample * f.n_pixels;
        uint32_t acc = f.mean_base + sample;            // uniforms.accumulated_count of this path's frame

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
            f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
            // barycentrics of the hit, recomputed from the triangle (tri_uv == tri_test's u, v)
            float hu, hv;
            tri_uv(ro, rd, mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), hu, hv);
            float bx = 1.0f - hu - hv, by = hu, bz = hv;
            normal = normalize((n0 * bx + n1 * by) + n2 * bz);
            float tu = (s2.y * bx + s2.w * by) + s3.y * bz;
            float tv = (s2.z * bx + s3.x * by) + s3.z * bz;
            f3 world_pos = ro + rd * hr.x;                                   // :186
            f3 albedo;
            const uint32_t tex = __float_as_uint(t0.w);                      // :190-198
            if (tex != 0xFFFFFFFFu) {
                if (tex < kLdsTextures) albedo = sample_texture(s_tex[tex], s_lut, tu, tv);
                else albedo = sample_texture(sc.textures[tex], s_lut, tu, tv);
                if (f.tex_times_factor) albedo = albedo * mk3(t2.w, s3.w, s4.x);   // build extension; the reference drops the factor
            } else {
                albedo = mk3(t2.w, s3.w, s4.x);
            }
            float roughness = t1.w;                                          // :200
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
                id_inst = __float_as_uint(s4.y); id_prim = __float_as_uint(s4.z);
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
    // the next tile's record: its fetch goes out behind the atomic, before the barrier that waits for the atomic's result
    {
        const uint32_t ntri = tri_of(tile + gridDim.x, nhr);
        t0 = kZero4; t1 = kZero4; t2 = kZero4; s0 = kZero4; s1 = kZero4; s2 = kZero4; s3 = kZero4; s4 = kZero4;
        if ((uint64_t)(tile + gridDim.x) * kShadeBlock < n_live && ntri != 0xFFFFFFFFu) {
            const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + ntri);
            t0 = sp[0]; t1 = sp[1]; t2 = sp[2]; s0 = sp[3]; s1 = sp[4]; s2 = sp[5]; s3 = sp[6]; s4 = sp[7];
        }
    }
    __syncthreads();
    if (alive) {
        uint32_t before = wave ? s_wave_live[wave - 1] : 0u;
        uint32_t slot = s_base + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        rays_out.org[slot] = o0; rays_out.dir[slot] = o1; rays_out.thr[slot] = o2;
    }
    r0 = nr0; r1 = nr1; r2 = nr2; hr = nhr;
    __syncthreads();      // the LDS scan words are reused by the next tile
    }
}

// ---------------------------------------------------------------- tail: the last bounces of a chunk in one launch
// The late bounces of a shallow scene are queues of a few hundred rays.  A traversal launch over such a queue is the
// dependent chain of its slowest ray on an otherwise idle chip (29 ... 90 us measured), and a frame pays it once per
// bounce.  From the bounce where the host expects a short queue (the lengths of the latest finished chunk: a guess that
// only affects speed) one launch takes every remaining ray to the end of its path: a lane owns a path, walks the tree
// for itself (Traversal::step, the scalar form of k_extend's walk: closest hit with the same tie-break, so the same
// hit) and shades it with k_shade's expression trees.  All chains of all remaining bounces then run side by side.
// Reference integrator only; bounce 0 (G-buffers, ids) never comes here.
__global__ void __launch_bounds__(64) k_tail(SceneDev sc, FrameDev f, uint32_t depth0, RayQueue rays,
                                             uint32_t *__restrict__ counters, float4 *__restrict__ results)
{
    __shared__ float s_lut[256];
    __shared__ TexDesc s_tex[kLdsTextures];
    __shared__ uint2 s_stack[kLdsStack * 64];
    const uint32_t n = counters[depth0];
    if (blockIdx.x * 64u >= n) return;
    const uint32_t lane = threadIdx.x;
    for (uint32_t w = lane; w < 256u; w += 64u) s_lut[w] = sc.srgb_lut[w];
    {
        const uint32_t nw = (sc.n_textures < kLdsTextures ? sc.n_textures : kLdsTextures) * (uint32_t)(sizeof(TexDesc) / 4);
        for (uint32_t w = lane; w < nw; w += 64u)
            reinterpret_cast<uint32_t *>(s_tex)[w] = reinterpret_cast<const uint32_t *>(sc.textures)[w];
    }
    __syncthreads();
    uint2 ovf_store[kOvfStack];
    LaneStack st;
    st.lds = (lds_uint2 *)s_stack + lane;      // kExtendBlock == 64: the column stride of LaneStack
    st.ovf = (prv_uint2 *)ovf_store;
    st.sp = 0;
    uint32_t n_hits = 0;
    for (uint32_t base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {
        const uint32_t i = base + lane;
        bool alive = i < n;
        f3 ro = mk3(0.0f, 0.0f, 0.0f), rd = ro, thr = ro;
        uint32_t path = 0, uv_state = 0;
        if (alive) {
            const float4 r0 = rays.org[i], r1 = rays.dir[i], r2 = rays.thr[i];
            ro = mk3(r0.x, r0.y, r0.z); rd = mk3(r1.x, r1.y, r1.z); thr = mk3(r2.x, r2.y, r2.z);
            path = __float_as_uint(r0.w); uv_state = __float_as_uint(r1.w);
        }
        const uint32_t sample = path / f.n_pixels;
        const uint32_t acc = f.mean_base + sample;
        for (uint32_t depth = depth0; ; ++depth) {
            const unsigned long long live = __ballot(alive);
            if (live == 0ull) break;
            // the queue lengths of the bounces that never get a queue: ptx_get_stats counts rays from them
            if (depth > depth0 && lane == 0) atomicAdd(&counters[depth], (uint32_t)__popcll(live));
            if (!alive) continue;
            Traversal<LaneStack, false> tr;
            tr.ng_mask = 0; tr.tg_mask = 0;
            if (tr.init(sc.n_nodes, ro, rd, f.t_min, f.t_max, st))
                while (!tr.step(sc.nodes, sc.tris, st)) {}
            const uint32_t tri = tr.hit.tri;
            f3 att, new_o = ro, new_d = rd;
            int done;
            if (tri == 0xFFFFFFFFu) {
                att = sky_color(rd);                                             // rmiss.glsl:18-34
                done = 1;
            } else {
                ++n_hits;
                const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
                const float4 t0 = sp[0], t1 = sp[1], t2 = sp[2], s0 = sp[3], s1 = sp[4], s2 = sp[5], s3 = sp[6], s4 = sp[7];
                // rchit.glsl:160-254, the expression trees of k_shade
                f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
                float hu, hv;
                tri_uv(ro, rd, mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), hu, hv);
                float bx = 1.0f - hu - hv, by = hu, bz = hv;
                f3 normal = normalize((n0 * bx + n1 * by) + n2 * bz);
                float tu = (s2.y * bx + s2.w * by) + s3.y * bz;
                float tv = (s2.z * bx + s3.x * by) + s3.z * bz;
                f3 world_pos = ro + rd * tr.hit.t;
                f3 albedo;
                const uint32_t tex = __float_as_uint(t0.w);
                if (tex != 0xFFFFFFFFu) {
                    if (tex < kLdsTextures) albedo = sample_texture(s_tex[tex], s_lut, tu, tv);
                    else albedo = sample_texture(sc.textures[tex], s_lut, tu, tv);
                    if (f.tex_times_factor) albedo = albedo * mk3(t2.w, s3.w, s4.x);
                } else {
                    albedo = mk3(t2.w, s3.w, s4.x);
                }
                float roughness = t1.w;
                float k2 = 2.0f * dot(normal, rd);
                f3 reflected = rd - normal * k2;
                f3 rr = normalize(cosine_hemisphere_sample(uv_state, depth, acc + f.rng_offset, f.seed, normal));
                float om = 1.0f - roughness;
                f3 scatter = reflected * om + rr * roughness;
                if (dot(scatter, normal) <= 0.0f) {
                    done = 1;
                    att = mk3(0.0f, 0.0f, 0.0f);
                } else {
                    done = 0;
                    att = albedo;
                }
                new_o = world_pos + normal * f.normal_eps;
                new_d = scatter;
            }
            thr = thr * att;                                                     // rgen.glsl:80
            if (done == 1 || depth + 1u >= f.max_depth) {                        // rgen.glsl:101-103
                results[path] = make_float4(thr.x, thr.y, thr.z, 1.0f);
                alive = false;
            } else {
                ro = new_o; rd = new_d;
            }
        }
    }
    const uint32_t h = wave_inclusive_sum(n_hits);       // lane 63 holds the wave's total
    if (lane == 63u && h) atomicAdd(&counters[kHitCounter], h);
}

// ---------------------------------------------------------------- shade, lit integrator
// Build extension: the stage that replaces k_shade when params.integrator is
// PTX_INTEGRATOR_LIT.  Specification: oracle/pt_oracle.c raygen_pixel_lit -- same expression trees, so
// the image is bit-identical to the oracle's.  Per hit: emission (camera / specular paths), one
// next-event-estimation sample (light picked by binary search of the power CDF, occlusion ray queued
// for an any-hit traversal), BSDF sampling (Lambert + GGX or smooth dielectric), Russian roulette.
// Two block-level compactions: surviving rays and occlusion rays, one queue-tail atomic each.
struct BlockScan {
    uint32_t *wave_cnt;   // kShadeBlock / 64 words of LDS
    uint32_t *base;       // one word of LDS
};
__device__ __forceinline__ uint32_t block_reserve(bool flag, const BlockScan &bs, uint32_t *counter)
{
    const unsigned long long m = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) bs.wave_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t c = lane < kShadeBlock / 64 ? bs.wave_cnt[lane] : 0u;
#pragma unroll
        for (int d = 1; d < kShadeBlock / 64; d <<= 1) {
            uint32_t t = __shfl_up(c, d, 64);
            if ((int)lane >= d) c += t;
        }
        if (lane < kShadeBlock / 64) bs.wave_cnt[lane] = c;
        if (lane == kShadeBlock / 64 - 1) *bs.base = c ? atomicAdd(counter, c) : 0u;
    }
    __syncthreads();
    const uint32_t before = wave ? bs.wave_cnt[wave - 1] : 0u;
    const uint32_t slot = *bs.base + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    __syncthreads();      // the LDS words are reused by the next reservation
    return slot;
}

// Both reservations of a k_shade_lit tile in ONE pass: the per-wave counts of the two flags travel packed in one word
// (<= 64 each), one lane issues both queue-tail atomics back to back, and the LDS words alternate between two buffers
// from tile to tile (`buf`), so that no barrier is needed behind the reads: two barriers per tile instead of six.  (A
// wave can only write buffer b again two tiles later, behind the two barriers of the tile in between, which every wave
// reaches after its reads of b.)  `wave_cnt`: 2 x kLitBlock / 64 words, `base`: 2 x 2 words.
__device__ __forceinline__ void block_reserve2(bool flag_a, bool flag_b, uint32_t *wave_cnt, uint32_t *base, uint32_t buf,
                                               uint32_t *counter_a, uint32_t *counter_b, uint32_t &slot_a, uint32_t &slot_b)
{
    constexpr uint32_t kWaves = kLitBlock / 64;
    const unsigned long long ma = __ballot(flag_a), mb = __ballot(flag_b);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *cnt = wave_cnt + buf * kWaves, *bs = base + buf * 2u;
    if (lane == 0) cnt[wave] = (uint32_t)__popcll(ma) | ((uint32_t)__popcll(mb) << 16);
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t c = lane < kWaves ? cnt[lane] : 0u;
#pragma unroll
        for (int d = 1; d < (int)kWaves; d <<= 1) {
            uint32_t t = __shfl_up(c, d, 64);
            if ((int)lane >= d) c += t;               // the halves cannot carry into each other: at most kLitBlock each
        }
        if (lane < kWaves) cnt[lane] = c;
        if (lane == kWaves - 1) {
            const uint32_t na = c & 0xFFFFu, nb = c >> 16;
            const uint32_t ba = na ? atomicAdd(counter_a, na) : 0u;
            const uint32_t bb = nb ? atomicAdd(counter_b, nb) : 0u;
            bs[0] = ba; bs[1] = bb;
        }
    }
    __syncthreads();
    const uint32_t before = wave ? cnt[wave - 1] : 0u;
    slot_a = bs[0] + (before & 0xFFFFu) + __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
    slot_b = bs[1] + (before >> 16) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
}

// One hit of the lit integrator (oracle: raygen_pixel_lit, the `hit` branch): emission, one next-event-estimation sample
// (its occlusion ray comes back in q0 .. q2), BSDF sampling, Russian roulette.  Shared by the wavefront stage (k_shade_lit)
// and the fused tail (k_tail_lit): one set of expression trees, so both give the oracle's bits.
constexpr uint32_t kLdsCdf = 4096;          // entries of k_shade_lit's LDS light table (16 KB)
constexpr uint32_t kLdsLitMaterials = 128;  // rows of SceneDev::lit_materials k_shade_lit keeps in LDS (8 KB): the row index comes out of the
                                            // triangle's record, so a global fetch of it is one more dependent round trip per hit
struct LitState {
    f3 thr, add, gb_normal, new_o, new_d;
    float gb_dist, next_pdf;
    bool have_add, done, was_hit, want_shadow;
    uint32_t id_inst, id_prim;
    float4 q0, q1, q2;
};
__device__ __forceinline__ void lit_shade_hit(const SceneDev &sc, const FrameDev &f, const TexDesc *s_tex, const float *s_lut,
                                              uint32_t depth, f3 o, f3 dir, uint32_t uv_state, uint32_t rng_acc, float prev_pdf,
                                              float hit_t, float path_bits, bool want_ids, const float *lds_cdf, uint32_t cdf_stride, const LitMaterial *lds_lm,
                                              const float4 &t0, const float4 &t1, const float4 &t2, const float4 &s0,
                                              const float4 &s1, const float4 &s2, const float4 &s3, const float4 &s4, LitState &S)
{
    f3 &thr = S.thr, &add = S.add, &gb_normal = S.gb_normal, &new_o = S.new_o, &new_d = S.new_d;
    float &gb_dist = S.gb_dist, &next_pdf = S.next_pdf;
    bool &have_add = S.have_add, &done = S.done, &was_hit = S.was_hit, &want_shadow = S.want_shadow;
    uint32_t &id_inst = S.id_inst, &id_prim = S.id_prim;
    float4 &q0 = S.q0, &q1 = S.q1, &q2 = S.q2;
    was_hit = true;
    const f3 va = mk3(t0.x, t0.y, t0.z), vb = mk3(t1.x, t1.y, t1.z), vc = mk3(t2.x, t2.y, t2.z);
    float hu, hv;
    tri_uv(o, dir, va, vb, vc, hu, hv);
    const float bx = 1.0f - hu - hv, by = hu, bz = hv;
    const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
    f3 ns = normalize((n0 * bx + n1 * by) + n2 * bz);
    f3 ng = normalize(cross(vb - va, vc - va));
    bool front = true;
    if (dot(ng, dir) > 0.0f) { ng = ng * -1.0f; front = false; }
    if (dot(ns, ng) < 0.0f) ns = ns * -1.0f;
    const float tu = (s2.y * bx + s2.w * by) + s3.y * bz;
    const float tv = (s2.z * bx + s3.x * by) + s3.z * bz;
    const f3 P = o + dir * hit_t;
    const uint32_t lm_row = __float_as_uint(s4.w);
    const LitMaterial lm = (lds_lm && lm_row < kLdsLitMaterials) ? lds_lm[lm_row] : sc.lit_materials[lm_row];
    f3 albedo = mk3(t2.w, s3.w, s4.x);
    const uint32_t tex = __float_as_uint(t0.w);
    // the hit's texture taps -- base colour, normal, metallic-roughness -- go out together (one round trip, not three);
    // each is decoded where the oracle's raygen_pixel_lit samples it.  The metallic-roughness tap is issued whatever lobe
    // is chosen below (its value is only read on the opaque branch).
    TexTap tap_b, tap_n, tap_m;
    tap_b.flags = tap_n.flags = tap_m.flags = 0xFFFFFFFFu;
    if (lm.set_flags) {
        // the material's texel set (ptx_kernels.h): everything the three fetches read sits in one entry -- one or two
        // 16-byte loads instead of up to twelve scattered dwords; the taps are then filled with the same raw texels
        const uint32_t sf = lm.set_flags, sw = lm.set_wh & 0xFFFFu, sh = lm.set_wh >> 16;
        float u = tu, v = tv;
        if (sf & kTexelSetRepeat) { u = u - floorf(u); v = v - floorf(v); }
        uint32_t a00 = 0, a10 = 0, a01 = 0, a11 = 0, b00 = 0, b10 = 0, b01 = 0, b11 = 0;
        float ax = 0.0f, ay = 0.0f;
        if (sf & kTexelSetLinear) {
            const int32_t kx = linear_axis_k0(u, sw, ax), ky = linear_axis_k0(v, sh, ay);
            const uint32_t entry = (uint32_t)(ky + 1) * (sw + 2u) + (uint32_t)(kx + 1);
            const uint4 *e = reinterpret_cast<const uint4 *>(lm.set_texels) + (uint64_t)entry * ((sf & kTexelSetHasB) ? 2u : 1u);
            const uint4 qa = e[0];
            a00 = qa.x; a10 = qa.y; a01 = qa.z; a11 = qa.w;
            if (sf & kTexelSetHasB) { const uint4 qb = e[1]; b00 = qb.x; b10 = qb.y; b01 = qb.z; b11 = qb.w; }
        } else {
            float fu = floorf(u * (float)sw), fv = floorf(v * (float)sh);
            const float mu = (float)(sw - 1u), mv = (float)(sh - 1u);
            if (!(fu >= 0.0f)) fu = 0.0f;
            if (fu > mu) fu = mu;
            if (!(fv >= 0.0f)) fv = 0.0f;
            if (fv > mv) fv = mv;
            const uint2 q = reinterpret_cast<const uint2 *>(lm.set_texels)[(uint64_t)((uint32_t)fv * sw + (uint32_t)fu)];
            a00 = q.x; b00 = q.y;
        }
        const uint32_t shared = sf & (kTexelSetRepeat | kTexelSetLinear);
        if (tex != 0xFFFFFFFFu) {
            tap_b.t00 = a00; tap_b.t10 = a10; tap_b.t01 = a01; tap_b.t11 = a11; tap_b.ax = ax; tap_b.ay = ay;
            tap_b.flags = shared | ((sf & kTexelSetBaseSrgb) ? 1u : 0u);
        }
        if (lm.normal_texture) {
            tap_n.t00 = b00; tap_n.t10 = b10; tap_n.t01 = b01; tap_n.t11 = b11; tap_n.ax = ax; tap_n.ay = ay;
            tap_n.flags = shared | ((sf & kTexelSetNormalSrgb) ? 1u : 0u);
        }
        if (lm.mr_texture) {
            // g from word A's top byte, b from word B's (the decode reads bytes 1 and 2)
            tap_m.t00 = ((a00 >> 24) << 8) | ((b00 >> 24) << 16); tap_m.t10 = ((a10 >> 24) << 8) | ((b10 >> 24) << 16);
            tap_m.t01 = ((a01 >> 24) << 8) | ((b01 >> 24) << 16); tap_m.t11 = ((a11 >> 24) << 8) | ((b11 >> 24) << 16);
            tap_m.ax = ax; tap_m.ay = ay;
            tap_m.flags = shared | ((sf & kTexelSetMrSrgb) ? 1u : 0u);
        }
    } else {
    if (tex != 0xFFFFFFFFu) tap_b = tex < kLdsTextures ? tap_issue(s_tex[tex], tu, tv) : tap_issue(sc.textures[tex], tu, tv);
    if (lm.normal_texture) {
        const uint32_t nt = lm.normal_texture - 1u;
        tap_n = nt < kLdsTextures ? tap_issue(s_tex[nt], tu, tv) : tap_issue(sc.textures[nt], tu, tv);
    }
    if (lm.mr_texture) {
        const uint32_t mt = lm.mr_texture - 1u;
        tap_m = mt < kLdsTextures ? tap_issue(s_tex[mt], tu, tv) : tap_issue(sc.textures[mt], tu, tv);
    }
    }
    if (tex != 0xFFFFFFFFu) {
        f3 tx = tap_resolve(tap_b, s_lut);
        albedo = tx * albedo;
    }
    if (lm.normal_texture) {
        // glTF normalTexture (oracle: raygen_pixel_lit): tangent-space normal, +Y up, the frame derived from the
        // triangle -- T along dP/du, B along -dP/dv, both made orthogonal to the shading normal; a degenerate uv
        // mapping or a perturbed normal below the geometric surface leaves the shading normal as it is
        const f3 tn = tap_resolve(tap_n, s_lut);
        const float mx = (tn.x * 2.0f - 1.0f) * lm.normal_scale, my = (tn.y * 2.0f - 1.0f) * lm.normal_scale, mz = tn.z * 2.0f - 1.0f;   // glTF normalTexture.scale
        const f3 e1 = vb - va, e2 = vc - va;
        const float du1 = s2.w - s2.y, dv1 = s3.x - s2.z, du2 = s3.y - s2.y, dv2 = s3.z - s2.z;
        const float det = du1 * dv2 - du2 * dv1;
        const float sg = det < 0.0f ? -1.0f : 1.0f;
        f3 tt = (e1 * dv2 - e2 * dv1) * sg;
        const f3 bt = (e1 * du2 - e2 * du1) * sg;
        tt = tt - ns * dot(ns, tt);
        const float tl2 = dot(tt, tt);
        if (det != 0.0f && tl2 > 0.0f && isfinite(tl2)) {
            tt = tt * (1.0f / sqrtf(tl2));
            f3 bb = cross(ns, tt);
            if (dot(bb, bt) < 0.0f) bb = bb * -1.0f;
            f3 np = (tt * mx + bb * my) + ns * mz;
            const float nl2 = dot(np, np);
            if (nl2 > 0.0f && isfinite(nl2)) {
                np = np * (1.0f / sqrtf(nl2));
                if (dot(np, ng) > 0.0f) ns = np;
            }
        }
    }
    const f3 wo = dir * -1.0f;
    const f3 up_o = P + ng * f.normal_eps;
    gb_normal = ns;
    gb_dist = length(up_o - o);
    if (want_ids) { id_inst = __float_as_uint(s4.y); id_prim = __float_as_uint(s4.z); }

    if (lm.emissive[0] > 0.0f || lm.emissive[1] > 0.0f || lm.emissive[2] > 0.0f) {
        f3 em = mk3(lm.emissive[0], lm.emissive[1], lm.emissive[2]);
        if (lm.emissive_texture) {
            const uint32_t et = lm.emissive_texture - 1u;
            em = em * (et < kLdsTextures ? sample_texture(s_tex[et], s_lut, tu, tv) : sample_texture(sc.textures[et], s_lut, tu, tv));
        }
        const f3 le = thr * em;
        if (!(prev_pdf > 0.0f)) {
            add = le;
            have_add = true;
        } else if (sc.n_lights) {
            // the light sampling of the previous vertex could have produced this point too
            const float cos_l = fabsf(dot(ng, dir));
            const float p_l = (lm.pdf_area * (hit_t * hit_t)) / cos_l;
            add = le * lit_mis_weight(prev_pdf, p_l);
            have_add = true;
        }
    }
    const float u_lobe = rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedLobe);
    const float u1 = rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedU1);
    const float u2 = rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedU2);
    if (lm.transmission > 0.0f && u_lobe < lm.transmission) {
        // smooth dielectric
        const float ior = lm.ior;
        const float eta = front ? 1.0f / ior : ior;
        float cosi = dot(wo, ns);
        if (cosi < 0.0f) cosi = 0.0f;
        if (cosi > 1.0f) cosi = 1.0f;
        const float sin2t = (eta * eta) * (1.0f - cosi * cosi);
        bool reflect = true;
        float cost = 0.0f;
        if (sin2t < 1.0f) {
            cost = sqrtf(1.0f - sin2t);
            float r0s = (1.0f - ior) / (1.0f + ior);
            r0s = r0s * r0s;
            const float c = front ? cosi : cost;
            const float fr = r0s + (1.0f - r0s) * pow5(1.0f - c);
            reflect = u1 < fr;
        }
        if (reflect) {
            new_d = normalize(dir + ns * (2.0f * cosi));
            new_o = up_o;
        } else {
            new_d = normalize(dir * eta + ns * (eta * cosi - cost));
            new_o = P - ng * f.normal_eps;
            thr = thr * albedo;
        }
        next_pdf = 0.0f;
    } else {
        float rough = t1.w, metal = lm.metallic;
        if (lm.mr_texture) {        // glTF: G = roughness, B = metallic, linear texels, multiplied into the factors
            const f3 mr = tap_resolve(tap_m, s_lut);
            rough = rough * mr.y;
            metal = metal * mr.z;
        }
        float alpha = rough * rough;
        if (alpha < 0.0025f) alpha = 0.0025f;
        const float a2 = alpha * alpha;
        const float om = 1.0f - metal;
        const f3 f0 = mk3(0.04f * om + albedo.x * metal, 0.04f * om + albedo.y * metal, 0.04f * om + albedo.z * metal);
        const f3 diff = albedo * om;
        const float p_spec = 0.25f + 0.75f * metal;

        // next-event estimation
        if (sc.n_lights) {
            const float ul = rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedLight);
            // first light whose cumulative probability exceeds ul (the oracle's binary search: the CDF is monotone, so any
            // search for that index agrees).  With the block's LDS table -- the last CDF value of every `cdf_stride` lights,
            // the whole CDF when there are no more than kLdsCdf -- the dozen dependent global loads of a plain binary search
            // become LDS reads plus log2(cdf_stride) global steps.
            uint32_t lo = 0, hi = sc.n_lights;
            if (lds_cdf) {
                const uint32_t nb = (sc.n_lights + cdf_stride - 1u) / cdf_stride;
                uint32_t bl = 0, bh = nb;
                while (bl < bh) { uint32_t mid = (bl + bh) / 2u; if (lds_cdf[mid] <= ul) bl = mid + 1u; else bh = mid; }
                lo = bl * cdf_stride;
                hi = lo + cdf_stride < sc.n_lights ? lo + cdf_stride : sc.n_lights;
                if (bl >= nb) { lo = sc.n_lights; hi = sc.n_lights; }
                if (cdf_stride == 1u) hi = lo;          // the table is the CDF itself
            }
            while (lo < hi) { uint32_t mid = (lo + hi) / 2u; if (sc.light_cdf[mid] <= ul) lo = mid + 1u; else hi = mid; }
            if (lo >= sc.n_lights) lo = sc.n_lights - 1u;
            const float4 *lp = reinterpret_cast<const float4 *>(sc.lights + lo);
            const float4 l0 = lp[0], l1 = lp[1], l2 = lp[2], l3 = lp[3];
            const float a = rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedLu);
            const float b = rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedLv);
            const float su = sqrtf(a);
            const float w0 = 1.0f - su, w1 = su * (1.0f - b), w2 = su * b;
            const f3 pl = (mk3(l0.x, l0.y, l0.z) * w0 + mk3(l1.x, l1.y, l1.z) * w1) + mk3(l2.x, l2.y, l2.z) * w2;
            const f3 dl = pl - up_o;
            const float dist2 = dot(dl, dl);
            const float inv_dist = 1.0f / sqrtf(dist2);
            const f3 wl = dl * inv_dist;
            const float cos_g = dot(ng, wl);
            const float cos_l = fabsf(dot(mk3(l3.x, l3.y, l3.z), wl));
            f3 fb; float p_b;
            if (dist2 > 0.0f && cos_g > 0.0f && cos_l > 0.0f && bsdf_eval(ns, wo, wl, diff, f0, a2, p_spec, fb, p_b)) {
                const float cos_s = dot(ns, wl);
                const float gterm = (cos_s * cos_l) / (dist2 * l0.w);
                const float p_l = (l0.w * dist2) / cos_l;
                f3 le_l = mk3(l1.w, l2.w, l3.w);
                const float4 l5 = lp[5];
                if (__float_as_uint(l5.z)) {
                    const float4 l4 = lp[4];
                    const uint32_t et = __float_as_uint(l5.z) - 1u;
                    const float lu = (l4.x * w0 + l4.z * w1) + l5.x * w2;
                    const float lv = (l4.y * w0 + l4.w * w1) + l5.y * w2;
                    le_l = le_l * (et < kLdsTextures ? sample_texture(s_tex[et], s_lut, lu, lv) : sample_texture(sc.textures[et], s_lut, lu, lv));
                }
                const f3 c = (((thr * fb) * le_l) * gterm) * lit_mis_weight(p_l, p_b);
                want_shadow = true;
                q0 = make_float4(up_o.x, up_o.y, up_o.z, path_bits);
                q1 = make_float4(dl.x, dl.y, dl.z, 0.0f);
                q2 = make_float4(c.x, c.y, c.z, 0.0f);
            }
        }

        // sample the next direction
        const float u_mix = rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedMix);
        f3 wi;
        if (u_mix < p_spec) {
            const f3 hv3 = ggx_sample_vndf(ns, wo, alpha, u1, u2);
            wi = normalize(hv3 * (2.0f * dot(wo, hv3)) - wo);
        } else {
            const float r = sqrtf(u1), phi = 6.283185307f * u2;
            float sn, cs;
            sincos_2pi(phi, sn, cs);
            const float lx = r * cs, ly = r * sn, oz = 1.0f - u1;
            const float lz = sqrtf(0.0f < oz ? oz : 0.0f);
            f3 tt, bb;
            create_basis(ns, tt, bb);
            wi = normalize((tt * lx + bb * ly) + ns * lz);
        }
        f3 fb; float pdf;
        if (dot(wi, ng) > 0.0f && bsdf_eval(ns, wo, wi, diff, f0, a2, p_spec, fb, pdf) && pdf > 0.0f) {
            const float wgt = dot(ns, wi) / pdf;
            thr = thr * (fb * wgt);
            new_d = wi;
            new_o = up_o;
            next_pdf = pdf;
        } else {
            done = true;
        }
    }
    // Russian roulette
    if (!done && depth >= 3u) {
        float pr = fmaxf(thr.x, fmaxf(thr.y, thr.z));
        if (pr > 1.0f) pr = 1.0f;
        if (pr < 0.05f) pr = 0.05f;
        if (rand_from(uv_state, depth, rng_acc, f.seed + kLitSeedRr) >= pr) done = true;
        else thr = thr * (1.0f / pr);
    }
    if (!done && !(finite3(thr) && finite3(new_d))) done = true;
}

// `order` (optional): the queue positions grouped by the material class of their hit (k_material_sort): thread s of
// the launch then shades ray order[s], so that a wave holds one kind of work -- sky, opaque lobes, glass.
__global__ void __launch_bounds__(kLitBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) k_shade_lit(SceneDev sc, FrameDev f, ImagesDev img, uint32_t depth,
                                                            RayQueue rays_in, const HitOut *__restrict__ hits,
                                                            uint32_t *__restrict__ counters, uint32_t capacity,
                                                            RayQueue rays_out, RayQueue shadow, float4 *__restrict__ results,
                                                            const uint32_t *__restrict__ order)
{
    __shared__ float s_lut[256];
    __shared__ TexDesc s_tex[kLdsTextures];
    __shared__ uint32_t s_cnt[2 * (kLitBlock / 64)];
    __shared__ uint32_t s_base[4], s_hits;
    __shared__ float s_cdf[kLdsCdf];        // light pick: last CDF value of every cdf_stride lights (see lit_shade_hit)
    __shared__ LitMaterial s_lm[kLdsLitMaterials];

    // capped grid, tiles of kLitBlock rays (see k_shade)
    const uint32_t n_live = counters[depth];
    if ((uint64_t)blockIdx.x * kLitBlock >= n_live) return;
    if (threadIdx.x == 0) s_hits = 0;
    const uint32_t cdf_stride = (sc.n_lights + kLdsCdf - 1u) / kLdsCdf > 0u ? (sc.n_lights + kLdsCdf - 1u) / kLdsCdf : 1u;
    for (uint32_t j = threadIdx.x; j * cdf_stride < sc.n_lights; j += kLitBlock) {
        const uint32_t last = (j + 1u) * cdf_stride - 1u;
        s_cdf[j] = sc.light_cdf[last < sc.n_lights ? last : sc.n_lights - 1u];
    }
    for (uint32_t j = threadIdx.x; j < 256u; j += kLitBlock) s_lut[j] = sc.srgb_lut[j];
    {
        const uint32_t nw = (sc.n_textures < kLdsTextures ? sc.n_textures : kLdsTextures) * (uint32_t)(sizeof(TexDesc) / 4);
        for (uint32_t w = threadIdx.x; w < nw; w += kLitBlock)
            reinterpret_cast<uint32_t *>(s_tex)[w] = reinterpret_cast<const uint32_t *>(sc.textures)[w];
        const uint32_t nm = (sc.n_lit_materials < kLdsLitMaterials ? sc.n_lit_materials : kLdsLitMaterials) * (uint32_t)(sizeof(LitMaterial) / 4);
        for (uint32_t w = threadIdx.x; w < nm; w += kLitBlock)
            reinterpret_cast<uint32_t *>(s_lm)[w] = reinterpret_cast<const uint32_t *>(sc.lit_materials)[w];
    }
    // The next tile's ray + hit are fetched behind this tile's shading, in front of its two queue reservations (two atomic
    // round trips, five barriers): one of a tile's dependent round trips leaves the critical path, and no register is
    // held across the shading itself (124 VGPRs keep two blocks per CU).
    const float4 kZero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto fetch_ray = [&](uint32_t tile_, float4 &a0, float4 &a1, float4 &a2, float2 &h) {
        const uint32_t slot_ = tile_ * kLitBlock + threadIdx.x;
        a0 = kZero4; a1 = kZero4; a2 = kZero4; h = make_float2(0.0f, __uint_as_float(0xFFFFFFFFu));
        if ((uint64_t)tile_ * kLitBlock >= n_live) return;
        const bool valid_ = slot_ < capacity && slot_ < n_live;
        const uint32_t i_ = (order && valid_) ? order[slot_] : slot_;
        if (i_ < capacity) {
            a1 = rays_in.dir[i_];
            // camera rays: origin, path index (= queue index) and unit throughput (rgen.glsl:62) are never stored
            a0 = depth == 0 ? make_float4(f.cam_origin[0], f.cam_origin[1], f.cam_origin[2], __uint_as_float(i_)) : rays_in.org[i_];
            a2 = depth == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : rays_in.thr[i_];
            h = *reinterpret_cast<const float2 *>(hits + i_);
        }
    };
    float4 r0, r1, r2; float2 hr;
    fetch_ray(blockIdx.x, r0, r1, r2, hr);
    __syncthreads();      // the LDS tables are complete; the tile loop's only barriers are the two of block_reserve2
    uint32_t buf = 0;
    for (uint32_t tile = blockIdx.x; (uint64_t)tile * kLitBlock < n_live; tile += gridDim.x, buf ^= 1u) {
    const uint32_t slot_in = tile * kLitBlock + threadIdx.x;
    const bool valid = slot_in < capacity && slot_in < n_live;
    const uint32_t tri = valid ? __float_as_uint(hr.y) : 0xFFFFFFFFu;
    float4 t0 = kZero4, t1 = kZero4, t2 = kZero4, s0 = kZero4, s1 = kZero4, s2 = kZero4, s3 = kZero4, s4 = kZero4;
    if (tri != 0xFFFFFFFFu) {
        const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
        t0 = sp[0]; t1 = sp[1]; t2 = sp[2]; s0 = sp[3]; s1 = sp[4]; s2 = sp[5]; s3 = sp[6]; s4 = sp[7];
    }

    bool alive = false, was_hit = false, want_shadow = false;
    float4 o0, o1, o2, q0, q1, q2;
    if (valid) {
        const f3 o = mk3(r0.x, r0.y, r0.z), dir = mk3(r1.x, r1.y, r1.z);
        const uint32_t path = __float_as_uint(r0.w), uv_state = __float_as_uint(r1.w);
        const uint32_t sample = path / f.n_pixels, pix = path - sample * f.n_pixels;
        const uint32_t acc = f.mean_base + sample;
        const uint32_t rng_acc = acc + f.rng_offset;
        const float prev_pdf = r2.w;      // solid-angle pdf the BSDF sampling gave this direction; <= 0: camera ray or delta lobe
        LitState S;
        S.thr = mk3(r2.x, r2.y, r2.z); S.add = mk3(0.0f, 0.0f, 0.0f); S.have_add = false;
        S.gb_normal = mk3(0.0f, 0.0f, 0.0f); S.gb_dist = 3.402823466e+38f;
        S.done = false; S.next_pdf = 0.0f; S.new_o = o; S.new_d = dir;
        S.id_inst = 0xFFFFFFFFu; S.id_prim = 0xFFFFFFFFu; S.was_hit = false; S.want_shadow = false;
        if (tri == 0xFFFFFFFFu) {
            S.add = S.thr * sky_color(dir);
            S.have_add = true;
            S.done = true;
        } else {
            lit_shade_hit(sc, f, s_tex, s_lut, depth, o, dir, uv_state, rng_acc, prev_pdf, hr.x, r0.w, depth == 0 && sample == f.ids_sample,
                          sc.n_lights ? s_cdf : nullptr, cdf_stride, s_lm, t0, t1, t2, s0, s1, s2, s3, s4, S);
        }

        was_hit = S.was_hit; want_shadow = S.want_shadow;
        q0 = S.q0; q1 = S.q1; q2 = S.q2;
        // radiance so far: results[path] is this path's L, started at bounce 0
        if (depth == 0 || S.have_add) {
            float4 cur = depth == 0 ? make_float4(0.0f, 0.0f, 0.0f, 1.0f) : results[path];
            if (S.have_add) { cur.x = cur.x + S.add.x; cur.y = cur.y + S.add.y; cur.z = cur.z + S.add.z; }
            cur.w = 1.0f;
            results[path] = cur;
        }
        if (depth == 0) {
            if (sample == f.ids_sample) img.ids[pix] = make_uint2(S.id_inst, S.id_prim);
            if (acc == 0) {
                float z_range = f.z_far - f.z_near;
                float zn = (S.gb_dist - f.z_near) / z_range;
                zn = fminf(fmaxf(zn, 0.0f), 1.0f);
                img.depth[pix] = zn;
                img.normal[pix] = make_float4(S.gb_normal.x, S.gb_normal.y, S.gb_normal.z, 1.0f);
            }
        }
        if (!S.done && depth + 1u < f.max_depth) {
            alive = true;
            o0 = make_float4(S.new_o.x, S.new_o.y, S.new_o.z, r0.w);
            o1 = make_float4(S.new_d.x, S.new_d.y, S.new_d.z, r1.w);
            o2 = make_float4(S.thr.x, S.thr.y, S.thr.z, S.next_pdf);
        }
    }

    fetch_ray(tile + gridDim.x, r0, r1, r2, hr);      // this tile's ray registers are dead: everything it produced is in o0 .. q2
    {
        // hits of the block: LDS atomic per wave, one global atomic per block (a single word takes ~88 M atomics/s)
        unsigned long long hitm = __ballot(was_hit);
        if ((threadIdx.x & 63u) == 0 && hitm) atomicAdd(&s_hits, (uint32_t)__popcll(hitm));
    }
    uint32_t slot, sslot;
    block_reserve2(alive, want_shadow, s_cnt, s_base, buf, &counters[depth + 1u], &counters[kShadowCount + depth], slot, sslot);
    if (alive) { rays_out.org[slot] = o0; rays_out.dir[slot] = o1; rays_out.thr[slot] = o2; }
    if (want_shadow) { shadow.org[sslot] = q0; shadow.dir[sslot] = q1; shadow.thr[sslot] = q2; }
    }
    __syncthreads();      // every wave's LDS add of its hits is done
    if (threadIdx.x == 0 && s_hits) atomicAdd(&counters[kHitCounter], s_hits);
}

// Fused tail of the lit integrator: k_tail's idea for PTX_INTEGRATOR_LIT.  From the first bounce whose expected queue is
// short, ONE launch takes every remaining path to its end -- a lane owns a path: closest-hit walk (Traversal::step),
// lit_shade_hit, the occlusion query of its light sample as a second scalar walk that stops at the first accepted
// triangle, the next bounce.  A wavefront bounce is four launches here (extend, shade, any-hit extend, shadow resolve),
// each at least the dependent chain of its slowest ray.  Radiance is summed in the order the wavefront stages use
// (results[path] + emission, then + the unoccluded light sample, bounce by bounce), so the image keeps its bits.
__global__ void __launch_bounds__(64) k_tail_lit(SceneDev sc, FrameDev f, uint32_t depth0, RayQueue rays,
                                                 uint32_t *__restrict__ counters, float4 *__restrict__ results)
{
    __shared__ float s_lut[256];
    __shared__ TexDesc s_tex[kLdsTextures];
    __shared__ uint2 s_stack[kLdsStack * 64];
    const uint32_t n = counters[depth0];
    if (blockIdx.x * 64u >= n) return;
    const uint32_t lane = threadIdx.x;
    for (uint32_t w = lane; w < 256u; w += 64u) s_lut[w] = sc.srgb_lut[w];
    {
        const uint32_t nw = (sc.n_textures < kLdsTextures ? sc.n_textures : kLdsTextures) * (uint32_t)(sizeof(TexDesc) / 4);
        for (uint32_t w = lane; w < nw; w += 64u)
            reinterpret_cast<uint32_t *>(s_tex)[w] = reinterpret_cast<const uint32_t *>(sc.textures)[w];
    }
    __syncthreads();
    uint2 ovf_store[kOvfStack];
    LaneStack st;
    st.lds = (lds_uint2 *)s_stack + lane;      // kExtendBlock == 64: the column stride of LaneStack
    st.ovf = (prv_uint2 *)ovf_store;
    st.sp = 0;
    uint32_t n_hits = 0;
    for (uint32_t base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {
        const uint32_t i = base + lane;
        bool alive = i < n;
        f3 o = mk3(0.0f, 0.0f, 0.0f), dir = o, thr = o;
        float prev_pdf = 0.0f, path_bits = 0.0f;
        uint32_t path = 0, uv_state = 0;
        float4 L = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        if (alive) {
            const float4 r0 = rays.org[i], r1 = rays.dir[i], r2 = rays.thr[i];
            o = mk3(r0.x, r0.y, r0.z); dir = mk3(r1.x, r1.y, r1.z); thr = mk3(r2.x, r2.y, r2.z);
            prev_pdf = r2.w; path_bits = r0.w;
            path = __float_as_uint(r0.w); uv_state = __float_as_uint(r1.w);
            L = results[path];          // depth0 >= 1: the path's radiance so far
        }
        const uint32_t sample = path / f.n_pixels;
        const uint32_t rng_acc = (f.mean_base + sample) + f.rng_offset;
        for (uint32_t depth = depth0; ; ++depth) {
            const unsigned long long live = __ballot(alive);
            if (live == 0ull) break;
            // the queue lengths of the bounces that never get a queue: ptx_get_stats counts rays from them
            if (depth > depth0 && lane == 0) atomicAdd(&counters[depth], (uint32_t)__popcll(live));
            LitState S;
            S.thr = thr; S.add = mk3(0.0f, 0.0f, 0.0f); S.have_add = false;
            S.gb_normal = mk3(0.0f, 0.0f, 0.0f); S.gb_dist = 3.402823466e+38f;
            S.done = false; S.next_pdf = 0.0f; S.new_o = o; S.new_d = dir;
            S.id_inst = 0xFFFFFFFFu; S.id_prim = 0xFFFFFFFFu; S.was_hit = false; S.want_shadow = false;
            if (alive) {
                Traversal<LaneStack, false> tr;
                tr.ng_mask = 0; tr.tg_mask = 0;
                if (tr.init(sc.n_nodes, o, dir, f.t_min, f.t_max, st))
                    while (!tr.step(sc.nodes, sc.tris, st)) {}
                const uint32_t tri = tr.hit.tri;
                if (tri == 0xFFFFFFFFu) {
                    S.add = S.thr * sky_color(dir);
                    S.have_add = true;
                    S.done = true;
                } else {
                    ++n_hits;
                    const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
                    const float4 t0 = sp[0], t1 = sp[1], t2 = sp[2], s0 = sp[3], s1 = sp[4], s2 = sp[5], s3 = sp[6], s4 = sp[7];
                    lit_shade_hit(sc, f, s_tex, s_lut, depth, o, dir, uv_state, rng_acc, prev_pdf, tr.hit.t, path_bits, false,
                                  nullptr, 1u, nullptr, t0, t1, t2, s0, s1, s2, s3, s4, S);
                }
                if (S.have_add) { L.x = L.x + S.add.x; L.y = L.y + S.add.y; L.z = L.z + S.add.z; }
            }
            // the occlusion ray of this bounce's light sample (counted like the wavefront's occlusion queue)
            const bool want = alive && S.want_shadow;
            const unsigned long long wm = __ballot(want);
            if (wm != 0ull && lane == 0) atomicAdd(&counters[kShadowCount + depth], (uint32_t)__popcll(wm));
            if (want) {
                Traversal<LaneStack, false> sh;
                sh.ng_mask = 0; sh.tg_mask = 0;
                bool occluded = false;
                if (sh.init(sc.n_nodes, mk3(S.q0.x, S.q0.y, S.q0.z), mk3(S.q1.x, S.q1.y, S.q1.z), f.t_min, kLitShadowTmax, st))
                    while (!sh.step(sc.nodes, sc.tris, st)) { if (sh.hit.tri != 0xFFFFFFFFu) break; }
                occluded = sh.hit.tri != 0xFFFFFFFFu;
                if (!occluded) { L.x = L.x + S.q2.x; L.y = L.y + S.q2.y; L.z = L.z + S.q2.z; }
            }
            if (alive) {
                if (S.done || depth + 1u >= f.max_depth) {
                    L.w = 1.0f;
                    results[path] = L;
                    alive = false;
                } else {
                    o = S.new_o; dir = S.new_d; thr = S.thr; prev_pdf = S.next_pdf;
                }
            }
        }
    }
    const uint32_t h = wave_inclusive_sum(n_hits);       // lane 63 holds the wave's total
    if (lane == 63u && h) atomicAdd(&counters[kHitCounter], h);
}

// Material sorting (BASELINE.json north_star: "ballot / prefix-sum primitives drive ray-queue compaction and material
// sorting"), measured as an option of the lit integrator (PTX_MATERIAL_SORT=1): between a bounce's
// traversal and its shading the queue positions are grouped by what the shading stage will do with them -- class 0:
// the ray left the scene (sky), 1: opaque material (Lambert + GGX lobes, next-event estimation), 2: transmissive
// material (glass lobe).  Per block: wave64 ballots rank the lanes of each class, the block's waves are scanned through
// LDS, one atomic per class and block reserves a run of that class's segment.  The result is a permutation
// (order[0 .. n)): class 0 positions first, then class 1, then class 2 -- the segment bases come from a first counting
// launch (kCount = true), the scatter from a second.
constexpr uint32_t kSortClasses = 3;
template <bool kCount>
__global__ void __launch_bounds__(kShadeBlock) k_material_sort(SceneDev sc, const HitOut *__restrict__ hits, const uint32_t *__restrict__ count,
                                                                uint32_t *__restrict__ class_count /* [3] totals, [3] cursors */,
                                                                uint32_t *__restrict__ order)
{
    __shared__ uint32_t s_cnt[kShadeBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_class[kSortClasses];
    const uint32_t n = *count;
    if ((uint64_t)blockIdx.x * kShadeBlock >= n) return;
    if (kCount) {
        if (threadIdx.x < kSortClasses) s_class[threadIdx.x] = 0;
        __syncthreads();
    }
    for (uint32_t tile = blockIdx.x; (uint64_t)tile * kShadeBlock < n; tile += gridDim.x) {
        const uint32_t i = tile * kShadeBlock + threadIdx.x;
        uint32_t cls = kSortClasses;      // none
        if (i < n) {
            const uint32_t tri = hits[i].tri;
            cls = 0;
            if (tri != 0xFFFFFFFFu) cls = sc.lit_materials[sc.shade[tri].material].transmission > 0.0f ? 2u : 1u;
        }
        BlockScan bs; bs.wave_cnt = s_cnt; bs.base = &s_base;
        if (kCount) {
#pragma unroll
            for (uint32_t c = 0; c < kSortClasses; ++c) {
                const unsigned long long m = __ballot(cls == c);
                if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&s_class[c], (uint32_t)__popcll(m));      // LDS; one global atomic per class and block below
            }
        } else {
            const uint32_t b1 = class_count[0], b2 = b1 + class_count[1];
#pragma unroll
            for (uint32_t c = 0; c < kSortClasses; ++c) {
                const uint32_t slot = block_reserve(cls == c, bs, &class_count[kSortClasses + c]);
                if (cls == c) order[(c == 0 ? 0u : (c == 1 ? b1 : b2)) + slot] = i;
            }
        }
    }
    if (kCount) {
        __syncthreads();
        if (threadIdx.x < kSortClasses && s_class[threadIdx.x]) atomicAdd(&class_count[threadIdx.x], s_class[threadIdx.x]);
    }
}

__global__ void __launch_bounds__(256) k_shadow_resolve(uint32_t depth, RayQueue shadow, const HitOut *__restrict__ hits,
                                                        const uint32_t *__restrict__ counters, float4 *__restrict__ results)
{
    const uint32_t n = counters[kShadowCount + depth];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {      // capped grid
        if (hits[i].tri != 0xFFFFFFFFu) continue;        // occluded
        const uint32_t path = __float_as_uint(shadow.org[i].w);      // one occlusion ray per path and bounce: no two lanes share a path
        const float4 c = shadow.thr[i];
        float4 cur = results[path];
        cur.x = cur.x + c.x; cur.y = cur.y + c.y; cur.z = cur.z + c.z;
        results[path] = cur;
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
        uint64_t rays = 0, shadow = 0;
        for (uint32_t b = 0; b < max_depth; ++b) { rays += counters[b]; shadow += counters[kShadowCount + b]; }
        totals[0] += rays;
        totals[11] += shadow;
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

void launch_raygen(hipStream_t s, const FrameDev &f, uint32_t n_paths, RayQueue rays, uint32_t *counters)
{
    uint32_t blocks = (n_paths + 255u) / 256u;
    hipLaunchKernelGGL(k_raygen, dim3(blocks ? blocks : 1u), dim3(256), 0, s, f, n_paths, rays, counters);
}

static const ExtendSched &extend_sched()
{
    static const ExtendSched sched = [] {
        auto knob = [](const char *name, int dflt) { const char *e = getenv(name); int v = e ? atoi(e) : dflt; return v < 1 ? 1 : (v > 64 ? 64 : v); };
        ExtendSched k{}; k.refill = knob("PTX_REFILL", 14); k.tri_min = knob("PTX_TRI_MIN", 36); k.flush_min = knob("PTX_FLUSH_MIN", 8);
        const char *ce = getenv("PTX_CHUNK"); int cv = ce ? atoi(ce) : 128; k.chunk = (uint32_t)(cv < 64 ? 64 : cv);
        k.chunk_grow = (uint32_t)knob("PTX_CHUNK_GROW", 4);
        return k;
    }();
    return sched;
}

void launch_extend(hipStream_t s, const SceneDev &sc, const FrameDev &f, RayQueue rays, const uint32_t *count,
                   uint32_t *next, uint32_t capacity, uint32_t max_waves, HitOut *hits, uint64_t *trav_totals, bool any_hit,
                   bool primary)
{
    ExtendSched sched = extend_sched();
    // only the camera rays form long queues of cheap, uniform rays; later bounces keep the small chunk (their rays are
    // heavy-tailed and a large chunk lengthens the kernel's tail: Sponza 16-spp batches lose 0.7 % with it)
    if (!primary) sched.chunk_grow = 1;
    sched.primary = primary ? 1u : 0u;
    sched.cam_o[0] = f.cam_origin[0]; sched.cam_o[1] = f.cam_origin[1]; sched.cam_o[2] = f.cam_origin[2];
    uint32_t blocks = std::min((capacity + kExtendBlock - 1u) / kExtendBlock, max_waves);
    if (!blocks) return;
    unsigned long long *tt = reinterpret_cast<unsigned long long *>(trav_totals);
    sched.seed_t = 0;
#ifdef PTX_LAB_SEED
    if (!any_hit && getenv("PTX_LAB_SEED")) {       // an ordinary pass first, then (below) the same queue again with seeded limits
        if (tt) hipLaunchKernelGGL((k_extend<true, false>), dim3(blocks), dim3(kExtendBlock), 0, s, sc, f.t_min, f.t_max, rays, count, next, hits, tt, sched);
        else hipLaunchKernelGGL((k_extend<false, false>), dim3(blocks), dim3(kExtendBlock), 0, s, sc, f.t_min, f.t_max, rays, count, next, hits, tt, sched);
        (void)hipMemsetAsync(next, 0, kHeadsPerLaunch * sizeof(uint32_t), s);
        sched.seed_t = 1;
    }
#endif
    if (any_hit) hipLaunchKernelGGL((k_extend<false, true>), dim3(blocks), dim3(kExtendBlock), 0, s, sc, f.t_min, kLitShadowTmax, rays, count, next, hits, (unsigned long long *)nullptr, sched);
    else if (tt) hipLaunchKernelGGL((k_extend<true, false>), dim3(blocks), dim3(kExtendBlock), 0, s, sc, f.t_min, f.t_max, rays, count, next, hits, tt, sched);
    else hipLaunchKernelGGL((k_extend<false, false>), dim3(blocks), dim3(kExtendBlock), 0, s, sc, f.t_min, f.t_max, rays, count, next, hits, tt, sched);
}

uint32_t extend_resident_waves(int device)
{
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_extend<false, false>, kExtendBlock, 0) != hipSuccess || per_cu <= 0) per_cu = 24;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    return (uint32_t)per_cu * (uint32_t)cus;
}

// most blocks a shading launch starts; a block walks tiles beyond that (ptx_set_option PTX_OPTION_SHADE_GRID; the
// environment variable PTX_SHADE_GRID only sets the default of new contexts)
uint32_t default_shade_grid()
{
    static const uint32_t cap = [] { const char *e = getenv("PTX_SHADE_GRID"); int v = e ? atoi(e) : 8192; return (uint32_t)(v < 64 ? 64 : v); }();
    return cap;
}
static uint32_t shade_grid_cap(uint32_t grid_cap) { return grid_cap ? (grid_cap < 64u ? 64u : grid_cap) : default_shade_grid(); }

void launch_shade(hipStream_t s, const SceneDev &sc, const FrameDev &f, const ImagesDev &img, uint32_t depth,
                  RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                  float4 *results, uint32_t grid_cap)
{
    uint32_t blocks = (capacity + kShadeBlock - 1u) / kShadeBlock;
    if (!blocks) return;
    blocks = std::min(blocks, shade_grid_cap(grid_cap));
    hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(kShadeBlock), 0, s, sc, f, img, depth, rays_in, hits, counters,
                       capacity, rays_out, results);
}

void launch_tail(hipStream_t s, const SceneDev &sc, const FrameDev &f, uint32_t depth0, RayQueue rays, uint32_t *counters,
                 uint32_t capacity, float4 *results)
{
    uint32_t blocks = std::min((capacity + 63u) / 64u, 4096u);
    if (!blocks) return;
    hipLaunchKernelGGL(k_tail, dim3(blocks), dim3(64), 0, s, sc, f, depth0, rays, counters, results);
}

void launch_tail_lit(hipStream_t s, const SceneDev &sc, const FrameDev &f, uint32_t depth0, RayQueue rays, uint32_t *counters,
                     uint32_t capacity, float4 *results)
{
    uint32_t blocks = std::min((capacity + 63u) / 64u, 4096u);
    if (!blocks) return;
    hipLaunchKernelGGL(k_tail_lit, dim3(blocks), dim3(64), 0, s, sc, f, depth0, rays, counters, results);
}

void launch_shade_lit(hipStream_t s, const SceneDev &sc, const FrameDev &f, const ImagesDev &img, uint32_t depth,
                      RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                      RayQueue shadow, float4 *results, const uint32_t *order, uint32_t grid_cap)
{
    uint32_t blocks = (capacity + kLitBlock - 1u) / kLitBlock;
    if (!blocks) return;
    blocks = std::min(blocks, shade_grid_cap(grid_cap));
    hipLaunchKernelGGL(k_shade_lit, dim3(blocks), dim3(kLitBlock), 0, s, sc, f, img, depth, rays_in, hits, counters,
                       capacity, rays_out, shadow, results, order);
}

void launch_material_sort(hipStream_t s, const SceneDev &sc, const HitOut *hits, const uint32_t *count, uint32_t capacity,
                          uint32_t *class_count, uint32_t *order, uint32_t grid_cap)
{
    uint32_t blocks = (capacity + kShadeBlock - 1u) / kShadeBlock;
    if (!blocks) return;
    blocks = std::min(blocks, shade_grid_cap(grid_cap));
    (void)hipMemsetAsync(class_count, 0, 2 * kSortClasses * sizeof(uint32_t), s);
    hipLaunchKernelGGL((k_material_sort<true>), dim3(blocks), dim3(kShadeBlock), 0, s, sc, hits, count, class_count, order);
    hipLaunchKernelGGL((k_material_sort<false>), dim3(blocks), dim3(kShadeBlock), 0, s, sc, hits, count, class_count, order);
}

void launch_shadow_resolve(hipStream_t s, uint32_t depth, RayQueue shadow, const HitOut *hits, const uint32_t *counters,
                           uint32_t capacity, float4 *results, uint32_t grid_cap)
{
    uint32_t blocks = (capacity + 255u) / 256u;
    if (!blocks) return;
    blocks = std::min(blocks, shade_grid_cap(grid_cap) * 2u);
    hipLaunchKernelGGL(k_shadow_resolve, dim3(blocks), dim3(256), 0, s, depth, shadow, hits, counters, results);
}

// Texel sets (ptx_kernels.h): one thread per entry gathers what the three fetches of that position read.  `present`:
// bit0 base, bit1 normal, bit2 metallic-roughness; the descriptors of absent planes are not read.
__global__ void __launch_bounds__(256) k_build_texel_set(uint32_t *__restrict__ dst, TexDesc tb, TexDesc tn, TexDesc tm, uint32_t present,
                                                          uint32_t w, uint32_t h, uint32_t set_flags)
{
    const bool linear = (set_flags & kTexelSetLinear) != 0u, has_b = (set_flags & kTexelSetHasB) != 0u;
    const uint32_t ew = linear ? w + 2u : w, eh = linear ? h + 2u : h;
    const uint64_t n = (uint64_t)ew * eh;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ex = (uint32_t)(i % ew), ey = (uint32_t)(i / ew);
        uint32_t xs[2], ys[2];
        if (linear) {
            axis_pair((int32_t)ex - 1, w, (set_flags & kTexelSetRepeat) != 0u, xs[0], xs[1]);
            axis_pair((int32_t)ey - 1, h, (set_flags & kTexelSetRepeat) != 0u, ys[0], ys[1]);
        } else {
            xs[0] = xs[1] = ex; ys[0] = ys[1] = ey;
        }
        uint32_t wa[4], wb[4];
        const int np = linear ? 4 : 1;
        for (int k = 0; k < np; ++k) {
            const uint32_t x = xs[k & 1], y = ys[k >> 1];
            const uint32_t b = (present & 1u) ? texel_raw(tb, x, y) : 0u;
            const uint32_t nr = (present & 2u) ? texel_raw(tn, x, y) : 0u;
            const uint32_t m = (present & 4u) ? texel_raw(tm, x, y) : 0u;
            wa[k] = (b & 0x00FFFFFFu) | (((m >> 8) & 0xFFu) << 24);
            wb[k] = (nr & 0x00FFFFFFu) | (((m >> 16) & 0xFFu) << 24);
        }
        if (linear) {
            uint32_t *e = dst + i * (has_b ? 8u : 4u);
            for (int k = 0; k < 4; ++k) e[k] = wa[k];
            if (has_b) for (int k = 0; k < 4; ++k) e[4 + k] = wb[k];
        } else {
            dst[i * 2u] = wa[0]; dst[i * 2u + 1u] = wb[0];
        }
    }
}

uint64_t texel_set_bytes(uint32_t w, uint32_t h, uint32_t set_flags)
{
    if (set_flags & kTexelSetLinear) return (uint64_t)(w + 2u) * (h + 2u) * ((set_flags & kTexelSetHasB) ? 32u : 16u);
    return (uint64_t)w * h * 8u;
}

void launch_build_texel_set(hipStream_t s, uint32_t *dst, const TexDesc &tb, const TexDesc &tn, const TexDesc &tm, uint32_t present,
                            uint32_t w, uint32_t h, uint32_t set_flags)
{
    const uint64_t n = (set_flags & kTexelSetLinear) ? (uint64_t)(w + 2u) * (h + 2u) : (uint64_t)w * h;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255u) / 256u, 65536u);
    if (!blocks) return;
    hipLaunchKernelGGL(k_build_texel_set, dim3(blocks), dim3(256), 0, s, dst, tb, tn, tm, present, w, h, set_flags);
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
