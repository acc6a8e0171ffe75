// ptx_refit.hip -- instance motion on the device (the reference rebuilds its TLAS on every frame,
// renderer.rs:961, and wants not to, todo.md:3): re-flatten the leaf triangles with the new instance transforms,
// refit the BVH8 bottom-up and re-quantise it, all in HBM -- no host flatten, no 48 MB upload.
//   k_refit_pack    one thread per leaf triangle: world-space vertices (and world-space normals) into TriRec and
//                   ShadeRec, raw triangle box, scene extent (wave reduce + ordered-uint atomics)
//   k_refit_nodes   one thread per node of one tree level, deepest level first: the host's refit_one_node
// The arithmetic is bvh_quant.h's, shared with the host refit: the nodes come out byte-identical (PTX_REFIT_VERIFY).
#include <hip/hip_runtime.h>

#include "bvh_quant.h"
#include "ptx_kernels.h"

namespace ptx {

namespace {

// monotone float <-> uint mapping so that atomicMin / atomicMax on uints order floats
__device__ __forceinline__ uint32_t f_to_ordered(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float ordered_to_f(uint32_t k)
{
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(b);
#else
    memcpy(&f, &b, 4);
#endif
    return f;
}

__device__ __forceinline__ void xform_point(const float *m, const float *p, float *out)
{
    for (int r = 0; r < 3; ++r) out[r] = ((m[r * 4] * p[0] + m[r * 4 + 1] * p[1]) + m[r * 4 + 2] * p[2]) + m[r * 4 + 3];
}
__device__ __forceinline__ void normal_to_world(const float *c, const float *n, float *out)
{
    for (int r = 0; r < 3; ++r) out[r] = (c[r * 3] * n[0] + c[r * 3 + 1] * n[1]) + c[r * 3 + 2] * n[2];
}

__global__ void __launch_bounds__(256) k_refit_pack(const RefitLeafSrc *__restrict__ src, const ptx_vertex *__restrict__ verts,
                                                    const RefitInstance *__restrict__ inst, uint32_t n_leaf, uint32_t world_normals,
                                                    TriRec *__restrict__ tris, ShadeRec *__restrict__ shade, Box *__restrict__ raw_box,
                                                    RefitState *__restrict__ state)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    Box b; b.reset();
    bool bad = false;
    if (k < n_leaf) {
        const RefitLeafSrc s = src[k];
        const RefitInstance &in = inst[s.instance];
        const ptx_vertex &a0 = verts[s.v[0]], &a1 = verts[s.v[1]], &a2 = verts[s.v[2]];
        float w[9];
        xform_point(in.xform, a0.position, w); xform_point(in.xform, a1.position, w + 3); xform_point(in.xform, a2.position, w + 6);
        TriRec &tr = tris[k];
        ShadeRec &sr = shade[k];
        for (int a = 0; a < 3; ++a) {
            tr.v0[a] = w[a]; tr.v1[a] = w[3 + a]; tr.v2[a] = w[6 + a];
            sr.v0[a] = w[a]; sr.v1[a] = w[3 + a]; sr.v2[a] = w[6 + a];
        }
        if (world_normals) {
            normal_to_world(in.cof, a0.normal, sr.n0); normal_to_world(in.cof, a1.normal, sr.n1); normal_to_world(in.cof, a2.normal, sr.n2);
        }
        bad = !tri_raw_box(w, b);
        raw_box[k] = b;
    }
    // scene extent over the finite triangles: wave reduction, then six atomics per wave
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = b.lo[a]; hi[a] = b.hi[a]; }
    for (int d = 32; d >= 1; d >>= 1)
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64)); }
    if ((threadIdx.x & 63u) == 0 && lo[0] <= hi[0])
        for (int a = 0; a < 3; ++a) { atomicMin(&state->ext_lo[a], f_to_ordered(lo[a])); atomicMax(&state->ext_hi[a], f_to_ordered(hi[a])); }
    if (bad) state->non_finite = 1u;
}

__global__ void __launch_bounds__(64) k_refit_nodes(Bvh8Node *__restrict__ nodes, Box *__restrict__ node_box, const Box *__restrict__ raw_box,
                                                     uint32_t first, uint32_t count, RefitState *__restrict__ state)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= count) return;
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, ordered_to_f(state->ext_hi[a]) - ordered_to_f(state->ext_lo[a]));
    Bvh8Node nd = nodes[first + i];
    Box nb = refit_one_node(nd, node_box, [&](uint32_t k) { Box t = raw_box[k]; pad_tri_box(t, ext); return t; });
    nodes[first + i] = nd;
    node_box[first + i] = nb;
    atomicAdd(&state->area_sum, (double)nb.half_area());
    if (first + i == 0) state->root_area = (double)nb.half_area();
}

}  // namespace

void refit_state_reset(RefitState *h)
{
    memset(h, 0, sizeof *h);
    for (int a = 0; a < 3; ++a) { h->ext_lo[a] = 0xFFFFFFFFu; h->ext_hi[a] = 0u; }
}

void launch_refit_pack(hipStream_t s, const RefitLeafSrc *src, const ptx_vertex *verts, const RefitInstance *inst, uint32_t n_leaf,
                       bool world_normals, TriRec *tris, ShadeRec *shade, void *raw_box, RefitState *state)
{
    if (!n_leaf) return;
    hipLaunchKernelGGL(k_refit_pack, dim3((n_leaf + 255u) / 256u), dim3(256), 0, s, src, verts, inst, n_leaf, world_normals ? 1u : 0u,
                       tris, shade, (Box *)raw_box, state);
}

void launch_refit_nodes(hipStream_t s, Bvh8Node *nodes, void *node_box, const void *raw_box, uint32_t first, uint32_t count,
                        RefitState *state)
{
    if (!count) return;
    hipLaunchKernelGGL(k_refit_nodes, dim3((count + 63u) / 64u), dim3(64), 0, s, nodes, (Box *)node_box, (const Box *)raw_box, first, count, state);
}

}  // namespace ptx
