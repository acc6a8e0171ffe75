// bvh_quant.h -- the arithmetic that turns boxes into a Bvh8Node's quantisation frame and 8-bit child planes, and
// the per-node step of a refit.  Shared by the host builder / refit (bvh_build.cpp) and the device refit
// (ptx_refit.hip) so that both produce byte-identical nodes; plain fp32 / fp64, no contraction.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "bvh8.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BQ_HD __host__ __device__ inline
#else
#define BQ_HD inline
#endif

namespace ptx {

struct Box {
    float lo[3], hi[3];
    BQ_HD void reset()
    {
        for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    }
    BQ_HD void grow(const Box &b)
    {
        for (int a = 0; a < 3; ++a) { lo[a] = b.lo[a] < lo[a] ? b.lo[a] : lo[a]; hi[a] = hi[a] < b.hi[a] ? b.hi[a] : hi[a]; }
    }
    BQ_HD bool valid() const { return lo[0] <= hi[0]; }
    BQ_HD float half_area() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

BQ_HD float step_of(uint8_t e)
{
    uint32_t u = (uint32_t)e << 23; float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}

// Raw box of a triangle (9 floats); false if a coordinate is not finite.
BQ_HD bool tri_raw_box(const float *v, Box &b)
{
    bool fin = true;
    for (int k = 0; k < 9; ++k) fin = fin && isfinite(v[k]);
    if (!fin) { b.reset(); return false; }
    b.reset();
    for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 3; ++a) {
            float x = v[k * 3 + a];
            b.lo[a] = x < b.lo[a] ? x : b.lo[a]; b.hi[a] = b.hi[a] < x ? x : b.hi[a];
        }
    return true;
}

// Pads a triangle box so that box culling can never contradict the triangle test (its edge functions are exact only
// up to a few ulps of the coordinate magnitude).
BQ_HD void pad_tri_box(Box &b, float scene_ext)
{
    for (int a = 0; a < 3; ++a) {
        float fl = fabsf(b.lo[a]), fh = fabsf(b.hi[a]);
        float m = fl < fh ? fh : fl;
        float pad = 2e-5f * m + 1e-6f * scene_ext + 1e-30f;
        b.lo[a] -= pad; b.hi[a] += pad;
    }
}

// quantisation frame of a node: origin = low corner, per-axis power-of-two step covering the extent in 255 steps
BQ_HD void set_node_frame(Bvh8Node &nd, const Box &box)
{
    for (int a = 0; a < 3; ++a) {
        nd.p[a] = box.lo[a];
        float ext = box.hi[a] - box.lo[a];
        int e; (void)frexpf(ext / 255.0f, &e);           // ext/255 = m * 2^e, m in [0.5,1)
        int be = e + 127;                                // step = 2^e >= ext/255
        be = be < 1 ? 1 : (be > 254 ? 254 : be);
        while (be < 254 && !(nd.p[a] + step_of((uint8_t)be) * 255.0f >= box.hi[a])) ++be;
        nd.e[a] = (uint8_t)be;
    }
}

// conservative 8-bit planes of child box `ch` in slot s: the decoded planes (fp32: p + q*step) enclose it
BQ_HD void set_child_planes(Bvh8Node &nd, int s, const Box &ch)
{
    for (int a = 0; a < 3; ++a) {
        float st = step_of(nd.e[a]);
        int lo = (int)floor(((double)ch.lo[a] - (double)nd.p[a]) / (double)st);
        int hi = (int)ceil(((double)ch.hi[a] - (double)nd.p[a]) / (double)st);
        lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo); hi = hi < 0 ? 0 : (hi > 255 ? 255 : hi);
        while (lo > 0 && nd.p[a] + (float)lo * st > ch.lo[a]) --lo;
        while (hi < 255 && nd.p[a] + (float)hi * st < ch.hi[a]) ++hi;
        nd.qlo[a][s] = (uint8_t)lo; nd.qhi[a][s] = (uint8_t)hi;
    }
}

BQ_HD uint32_t bq_popc(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}

// One node of a refit: boxes of the used slots from the leaf triangles' padded boxes (leaf_box(k) for leaf-order
// triangle k) and the inner children's node boxes, then frame and planes.  Returns the node's box.
template <class LeafBox>
BQ_HD Box refit_one_node(Bvh8Node &nd, const Box *node_box, LeafBox leaf_box)
{
    Box child[8]; bool used[8];
    Box nb; nb.reset();
    for (int s = 0; s < 8; ++s) {
        used[s] = nd.meta[s] != 0;
        if (!used[s]) continue;
        Box cb; cb.reset();
        if ((nd.imask >> s) & 1u) {
            cb = node_box[nd.child_base + bq_popc(nd.imask & ((1u << s) - 1u))];
        } else {
            uint32_t off = nd.meta[s] & 31u, cnt = bq_popc((uint32_t)nd.meta[s] >> 5);
            for (uint32_t k = 0; k < cnt; ++k) cb.grow(leaf_box(nd.tri_base + off + k));
        }
        child[s] = cb; nb.grow(cb);
    }
    set_node_frame(nd, nb);
    for (int s = 0; s < 8; ++s) {
        if (used[s]) set_child_planes(nd, s, child[s]);
        else for (int a = 0; a < 3; ++a) { nd.qlo[a][s] = 0; nd.qhi[a][s] = 0; }
    }
    return nb;
}

}  // namespace ptx
