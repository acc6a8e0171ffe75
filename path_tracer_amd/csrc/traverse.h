// traverse.h -- closest-hit query over the compressed 8-wide BVH: what traceRayEXT
// (rgen.glsl:66-78) does in the reference's driver/hardware, as per-lane software for CDNA4.
//
// The query is a small state machine (Traversal::step) so that a persistent wave can keep all
// 64 lanes busy: one step = at most one node visit (8 quantised box tests, ray-octant ordered)
// plus at most one triangle test, and a lane whose ray has finished is refilled with a new ray
// by the caller.  Short stack of (node group, hit mask) pairs.  The result is order-independent:
// closest t, ties -> lowest (instance, primitive); box culling keeps t == t_best candidates.
#pragma once

#include "bvh8.h"
#include "pt_math.h"

namespace ptx {

struct HitRec {
    float t, u, v;
    uint32_t tri;    // index into the leaf-ordered TriRec array; 0xFFFFFFFF = miss
};

struct Words80 { uint32_t w[20]; };

PT_HD Words80 load_node(const Bvh8Node *nodes, uint32_t i)
{
    Words80 r;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 *p = reinterpret_cast<const uint4 *>(nodes + i);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
    r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
    r.w[8] = c.x; r.w[9] = c.y; r.w[10] = c.z; r.w[11] = c.w;
    r.w[12] = d.x; r.w[13] = d.y; r.w[14] = d.z; r.w[15] = d.w;
    r.w[16] = e.x; r.w[17] = e.y; r.w[18] = e.z; r.w[19] = e.w;
#else
    memcpy(r.w, nodes + i, 80);
#endif
    return r;
}

PT_HD uint32_t popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}
PT_HD uint32_t msb32(uint32_t x)   // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 31u - (uint32_t)__clz((int)x);
#else
    return 31u - (uint32_t)__builtin_clz(x);
#endif
}
PT_HD uint32_t lsb32(uint32_t x)   // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__ffs((int)x) - 1u;
#else
    return (uint32_t)__builtin_ctz(x);
#endif
}
// box tests only need to be conservative, not reproducible: use the hardware FMA
PT_HD float fma_f(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_HD float byte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }

struct TraverseStats { uint32_t nodes, tris; };

// Per-ray constants of the slab tests: reciprocal direction (|d| clamped away from 0 so that no
// 0 * inf NaN can arise) and the octant word -- bits 2..0 = octinv, kRayDegenerate if the ray is
// non-finite and must miss everything.
constexpr uint32_t kRayDegenerate = 0x80000000u;
PT_HD uint32_t ray_slab_setup(f3 org, f3 dir, float &ix, float &iy, float &iz)
{
    const float tiny = 1e-18f;
    float dx = fabsf(dir.x) > tiny ? dir.x : copysignf(tiny, dir.x);
    float dy = fabsf(dir.y) > tiny ? dir.y : copysignf(tiny, dir.y);
    float dz = fabsf(dir.z) > tiny ? dir.z : copysignf(tiny, dir.z);
    ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz;
    uint32_t oct = (dx < 0.0f ? 0u : 4u) | (dy < 0.0f ? 0u : 2u) | (dz < 0.0f ? 0u : 1u);
    if (!finite3(org) || !finite3(dir)) oct |= kRayDegenerate;
    return oct;
}

#if defined(__HIPCC__)
// The same with hardware reciprocals (v_rcp_f32, 1 ulp) for the stage that fetches rays on the GPU: the slab tests only
// have to be conservative, not reproducible.
__device__ __forceinline__ uint32_t ray_slab_setup_rcp(f3 org, f3 dir, float &ix, float &iy, float &iz)
{
    const float tiny = 1e-18f;
    float dx = fabsf(dir.x) > tiny ? dir.x : copysignf(tiny, dir.x);
    float dy = fabsf(dir.y) > tiny ? dir.y : copysignf(tiny, dir.y);
    float dz = fabsf(dir.z) > tiny ? dir.z : copysignf(tiny, dir.z);
    ix = __builtin_amdgcn_rcpf(dx); iy = __builtin_amdgcn_rcpf(dy); iz = __builtin_amdgcn_rcpf(dz);
    uint32_t oct = (dx < 0.0f ? 0u : 4u) | (dy < 0.0f ? 0u : 2u) | (dz < 0.0f ? 0u : 1u);
    if (!finite3(org) || !finite3(dir)) oct |= kRayDegenerate;
    return oct;
}
#endif

// Stack: any type with push(uint32_t, uint32_t), pop(uint32_t&, uint32_t&), empty(), clear().
template <class Stack, bool kStats>
struct Traversal {
    // ray
    f3 o, d;
    float t_min, t_max;
    float idx, idy, idz;        // reciprocal direction for the slab tests
    uint32_t octinv;
    // traversal state
    uint32_t ng_base, ng_mask;  // node group: first-child index | hit bits 31..24, imask bits 7..0
    uint32_t tg_base, tg_mask;  // triangle group: first triangle | pending bits 23..0
    HitRec hit;
    uint32_t best_inst, best_prim;
    TraverseStats stats;

    // Returns false if the ray needs no traversal at all (empty scene, non-finite ray: such
    // rays hit nothing, the same rule as the oracle).
    PT_HD bool init(uint32_t n_nodes, f3 org, f3 dir, float tmin, float tmax, Stack &stack)
    {
        d = dir;
        float ix, iy, iz;
        uint32_t oct = ray_slab_setup(org, dir, ix, iy, iz);
        return init_prepared(n_nodes, org, ix, iy, iz, oct, tmin, tmax, stack);
    }

    // Same with the slab set-up already done (the GPU stages that CREATE a ray do it at full lane
    // utilisation and pass it through the ray queue; `d` is not kept).
    PT_HD bool init_prepared(uint32_t n_nodes, f3 org, float ix, float iy, float iz, uint32_t oct, float tmin, float tmax,
                             Stack &stack)
    {
        o = org; t_min = tmin; t_max = tmax;
        hit.t = tmax; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu;
        best_inst = 0xFFFFFFFFu; best_prim = 0xFFFFFFFFu;
        stats.nodes = 0; stats.tris = 0;
        stack.clear();
        tg_base = 0; tg_mask = 0;
        ng_base = 0; ng_mask = 0;
        idx = ix; idy = iy; idz = iz;
        octinv = oct & 7u;
        if (n_nodes == 0 || (oct & kRayDegenerate)) return false;
        ng_mask = 0x80000000u;   // the root as the single pending child of a virtual group (imask 0 -> index 0)
        return true;
    }

    // Takes the nearest pending child of the current node group (pushing the rest of the group)
    // and returns its node index.
    PT_HD uint32_t node_pick(Stack &stack)
    {
        uint32_t b = msb32(ng_mask);
        uint32_t imask = ng_mask & 0xFFu;
        ng_mask &= ~(1u << b);
        if (ng_mask & 0xFF000000u) stack.push(ng_base, ng_mask);
        uint32_t slot = (b - 24u) ^ octinv;
        uint32_t rel = popc32(imask & ((1u << slot) - 1u));
        return ng_base + rel;
    }

    PT_HD void node_step(const Bvh8Node *nodes, Stack &stack)
    {
        uint32_t ni = node_pick(stack);
        node_eval(load_node(nodes, ni));
    }

    // 8 quantised box tests against node `n`; the hits become the new node / triangle groups.
    PT_HD void node_eval(const Words80 &n)
    {
        if (kStats) stats.nodes++;

        float px = u2f(n.w[0]), py = u2f(n.w[1]), pz = u2f(n.w[2]);
        uint32_t ew = n.w[3];
        float sx = u2f((ew & 0xFFu) << 23), sy = u2f(((ew >> 8) & 0xFFu) << 23), sz = u2f(((ew >> 16) & 0xFFu) << 23);
        float ax = sx * idx, ay = sy * idy, az = sz * idz;            // t per quantisation step
#if defined(__HIP_DEVICE_COMPILE__)
        // plane times are measured from t_min (folded into the per-axis offset for free), see below
        float bx = fma_f(px - o.x, idx, -t_min), by = fma_f(py - o.y, idy, -t_min), bz = fma_f(pz - o.z, idz, -t_min);
#else
        float bx = (px - o.x) * idx, by = (py - o.y) * idy, bz = (pz - o.z) * idz;
#endif
        const bool negx = !(octinv & 4u), negy = !(octinv & 2u), negz = !(octinv & 1u);

        uint32_t hitmask = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        // gfx950 issues vector instructions on two pipes (profiles/r02_valu_calibration.json): FMA / add / mul, the
        // bitwise ops (v_bitop3 included), integer add and RIGHT shifts run at 2 cycles per wave on either pipe;
        // everything else (conversions, min / max, compares, v_cndmask, left shifts, v_pk_*, anything reading an SGPR)
        // takes 4 cycles on one pipe only.  The test is written for that machine: plain FMAs with VGPR operands (a
        // v_pk_fma_f32 is two half-rate slots for two FMAs: no gain), near / far planes picked by bitfield select
        // against per-axis sign masks, and the box decision as the sign of (t_far - t_near) smeared over the word and
        // folded into the hit mask with one three-input bit operation, instead of v_cmp + v_cndmask + v_or.
        // With times measured from t_min the box decision is: a <= b, a <= t_best - t_min and b >= 0, where a / b are the
        // latest entry / earliest exit over the three slabs: one v_max3, one v_min3, two subtractions and the OR of three
        // sign bits, instead of clamping a and b against t_min / t_best with two more half-rate v_max / v_min.  The limit
        // is widened by a few ulp so that the rounding of the shifted times cannot cull a candidate at t == t_best.
        // The half-rate pipe keeps the 48 byte conversions, v_max3 + v_min3 per child and one byte-selecting shift per child.
        const uint32_t mx = negx ? 0xFFFFFFFFu : 0u, my = negy ? 0xFFFFFFFFu : 0u, mz = negz ? 0xFFFFFFFFu : 0u;
        const float limit = fma_f(hit.t, 1.0f + 0x1p-21f, -t_min);
#ifdef PTX_PROBE
        // developer experiment (tools/probe_sweep.sh): 40 extra instructions per node visit of one kind -- what is the kernel sensitive to?
        {
            uint32_t p0 = n.w[0], p1 = n.w[1], p2 = n.w[2], p3 = n.w[3];
#pragma unroll
            for (int r = 0; r < 10; ++r) {
#if PTX_PROBE == 1      // full-rate vector
                asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
#elif PTX_PROBE == 2    // half-rate vector
                asm volatile("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 1, 31" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
#elif PTX_PROBE == 3    // scalar
                asm volatile("s_add_u32 s90, s90, 1\n s_add_u32 s91, s91, 1\n s_add_u32 s92, s92, 1\n s_add_u32 s93, s93, 1" : : : "s90", "s91", "s92", "s93", "scc");
#elif PTX_PROBE == 4    // idle issue slots
                asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
#endif
            }
            asm volatile("" : : "v"(p0 ^ p1 ^ p2 ^ p3));      // keeps the probe's instructions alive; the traversal's result is untouched
        }
#endif
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t meta4 = n.w[6 + half];
            uint32_t qlx = n.w[8 + half], qly = n.w[10 + half], qlz = n.w[12 + half];
            uint32_t qhx = n.w[14 + half], qhy = n.w[16 + half], qhz = n.w[18 + half];
            // (lo & ~m) | (hi & m): bitop3 table 0xD8 with sources (lo, hi, m)
            uint32_t nx4 = __builtin_amdgcn_bitop3_b32(qlx, qhx, mx, 0xD8), fx4 = __builtin_amdgcn_bitop3_b32(qhx, qlx, mx, 0xD8);
            uint32_t ny4 = __builtin_amdgcn_bitop3_b32(qly, qhy, my, 0xD8), fy4 = __builtin_amdgcn_bitop3_b32(qhy, qly, my, 0xD8);
            uint32_t nz4 = __builtin_amdgcn_bitop3_b32(qlz, qhz, mz, 0xD8), fz4 = __builtin_amdgcn_bitop3_b32(qhz, qlz, mz, 0xD8);
            uint32_t inner4 = ((meta4 & (meta4 + meta4)) >> 4) & 0x01010101u;   // 1 in each byte that is an inner child
            uint32_t bit4 = (meta4 ^ (inner4 * octinv)) & 0x1F1F1F1Fu;          // hit-mask bit position per child
            uint32_t cnt4 = (meta4 >> 5) & 0x07070707u;                         // unary triangle count (1 for inner)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float tnx = fma_f(byte_f(nx4, k), ax, bx), tfx = fma_f(byte_f(fx4, k), ax, bx);
                float tny = fma_f(byte_f(ny4, k), ay, by), tfy = fma_f(byte_f(fy4, k), ay, by);
                float tnz = fma_f(byte_f(nz4, k), az, bz), tfz = fma_f(byte_f(fz4, k), az, bz);
                float a = fmaxf(fmaxf(tnx, tny), tnz);
                float b = fminf(fminf(tfx, tfy), tfz);
                // cnt << bit with both operands picked out of their words by the instruction itself
                uint32_t bits;
                if (k == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(bits) : "v"(bit4), "v"(cnt4));
                if (k == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(bits) : "v"(bit4), "v"(cnt4));
                if (k == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(bits) : "v"(bit4), "v"(cnt4));
                if (k == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(bits) : "v"(bit4), "v"(cnt4));
                // miss = all ones iff any of (b - a), (limit - a), b is negative (the t values are finite: directions are
                // clamped away from 0 and non-finite rays never get here; x - x is +0, so a == b counts as a hit)
                float d1 = b - a, d2 = limit - a;
                uint32_t sgn = __builtin_amdgcn_bitop3_b32(__float_as_uint(d1), __float_as_uint(d2), __float_as_uint(b), 0xFE);
                uint32_t miss;
                asm("v_ashrrev_i32 %0, 31, %1" : "=v"(miss) : "v"(sgn));
                hitmask = __builtin_amdgcn_bitop3_b32(hitmask, bits, miss, 0xF4);   // hitmask | (bits & ~miss)
            }
        }
#else
        for (int half = 0; half < 2; ++half) {
            uint32_t meta4 = n.w[6 + half];
            // near / far planes by ray octant
            uint32_t qlx = n.w[8 + half], qly = n.w[10 + half], qlz = n.w[12 + half];
            uint32_t qhx = n.w[14 + half], qhy = n.w[16 + half], qhz = n.w[18 + half];
            uint32_t nx4 = negx ? qhx : qlx, fx4 = negx ? qlx : qhx;
            uint32_t ny4 = negy ? qhy : qly, fy4 = negy ? qly : qhy;
            uint32_t nz4 = negz ? qhz : qlz, fz4 = negz ? qlz : qhz;
            for (int k = 0; k < 4; ++k) {
                uint32_t meta = (meta4 >> (8 * k)) & 0xFFu;
                float tnx = fma_f(byte_f(nx4, k), ax, bx), tfx = fma_f(byte_f(fx4, k), ax, bx);
                float tny = fma_f(byte_f(ny4, k), ay, by), tfy = fma_f(byte_f(fy4, k), ay, by);
                float tnz = fma_f(byte_f(nz4, k), az, bz), tfz = fma_f(byte_f(fz4, k), az, bz);
                float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, t_min));
                float tf = fminf(fminf(tfx, tfy), fminf(tfz, hit.t));
                bool inner = (meta & 0x18u) == 0x18u;
                uint32_t bit = (inner ? (meta ^ octinv) : meta) & 31u;
                uint32_t bits = meta >> 5;
                if (tn <= tf) hitmask |= bits << bit;
            }
        }
#endif
        ng_base = n.w[4];
        ng_mask = (hitmask & 0xFF000000u) | (ew >> 24);
        tg_base = n.w[5];
        tg_mask = hitmask & 0x00FFFFFFu;
    }

    // Takes the next pending triangle of the current triangle group.
    PT_HD uint32_t tri_pick()
    {
        uint32_t i = lsb32(tg_mask);
        tg_mask &= tg_mask - 1u;
        return tg_base + i;
    }

    PT_HD void tri_eval(uint32_t ti, f3 v0, f3 v1, f3 v2, uint32_t inst, uint32_t prim)
    {
        if (kStats) stats.tris++;
        float t, u, v;
        if (tri_test(o, d, v0, v1, v2, t_min, t_max, t, u, v) && hit_better(t, inst, prim, hit.t, best_inst, best_prim)) {
            hit.t = t; hit.u = u; hit.v = v; hit.tri = ti;
            best_inst = inst; best_prim = prim;
        }
    }

    PT_HD void tri_step(const TriRec *tris)
    {
        uint32_t ti = tri_pick();
#if defined(__HIP_DEVICE_COMPILE__)
        const float4 *tp = reinterpret_cast<const float4 *>(tris + ti);
        float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
        f3 v0 = mk3(r0.x, r0.y, r0.z), v1 = mk3(r1.x, r1.y, r1.z), v2 = mk3(r2.x, r2.y, r2.z);
        uint32_t inst = __float_as_uint(r0.w), prim = __float_as_uint(r1.w);
#else
        const TriRec &tr = tris[ti];
        f3 v0 = mk3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = mk3(tr.v1[0], tr.v1[1], tr.v1[2]),
           v2 = mk3(tr.v2[0], tr.v2[1], tr.v2[2]);
        uint32_t inst = tr.instance, prim = tr.prim;
#endif
        tri_eval(ti, v0, v1, v2, inst, prim);
    }

    // Scheduling pieces for a wave that batches node visits and triangle tests separately:
    // a lane wants a node visit while it has no triangle pending.
    PT_HD bool wants_node() const { return tg_mask == 0; }
    PT_HD bool wants_tri() const { return tg_mask != 0; }
    // Makes a node group current (popping the stack if the current one is spent).  Returns false
    // when nothing is left: the query is complete (result in `hit`).
    PT_HD bool next_group(Stack &stack)
    {
        if (!(ng_mask & 0xFF000000u)) {
            if (stack.empty()) return false;
            stack.pop(ng_base, ng_mask);
        }
        return true;
    }

    // One unit of work.  Returns true when the query is complete (result in `hit`).
    PT_HD bool step(const Bvh8Node *nodes, const TriRec *tris, Stack &stack)
    {
        if (wants_node()) {
            if (!next_group(stack)) return true;
            node_step(nodes, stack);
        }
        if (wants_tri()) tri_step(tris);
        return false;
    }
};

}  // namespace ptx
