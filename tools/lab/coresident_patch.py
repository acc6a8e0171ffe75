"""Developer experiment, kept for the record (profiles/r02_coresident_shading.json): the source patch that
ran the shading stage beside the other chunk's traversal kernel (64-VGPR / 256-thread build of k_shade, 7/8 traversal grid,
PTX_CORESIDENT=k).  It applies to the tree at commit f6afedd (before k_shade's tile loop was software-pipelined), not to HEAD:
    git stash; git checkout f6afedd -- path_tracer_amd/csrc; python tools/lab/coresident_patch.py [ntrec]; python -m path_tracer_amd.build
"""
import sys
NT_REC = len(sys.argv) > 1 and sys.argv[1] == "ntrec"
p='/root/repo/path_tracer_amd/csrc/ptx_kernels.hip'
s=open(p).read()
a=s.index('#ifdef PTX_SHADE_WAVES\n__attribute__((amdgpu_waves_per_eu(PTX_SHADE_WAVES, PTX_SHADE_WAVES)))\n#endif\n__global__ void __launch_bounds__(kShadeBlock) k_shade(')
b=s.index('// ---------------------------------------------------------------- tail: the last bounces')
body=s[a:b]
body=body.replace('''#ifdef PTX_SHADE_WAVES
__attribute__((amdgpu_waves_per_eu(PTX_SHADE_WAVES, PTX_SHADE_WAVES)))
#endif
__global__ void __launch_bounds__(kShadeBlock) k_shade(SceneDev sc, FrameDev f, ImagesDev img, uint32_t depth,
                                                        RayQueue rays_in,
                                                        const HitOut *__restrict__ hits,
                                                        uint32_t *__restrict__ counters, uint32_t capacity,
                                                        RayQueue rays_out, float4 *__restrict__ results)
{''','''// Stream accesses of the co-resident build: non-temporal, so that the 6 GB a shading launch moves do not push the
// traversal kernel's nodes and triangles out of the caches they share while the two run side by side.
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <bool kNt> __device__ __forceinline__ float4 ld4(const float4 *p)
{
    if (kNt) { v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
}
template <bool kNt> __device__ __forceinline__ float2 ld2(const float2 *p)
{
    if (kNt) { v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p)); return make_float2(v.x, v.y); }
    return *p;
}
template <bool kNt> __device__ __forceinline__ void st4(float4 *p, float4 a)
{
    if (kNt) { v4f_t v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<v4f_t *>(p)); }
    else *p = a;
}

// Two builds of the one body.  <kShadeBlock, 2>: 512-thread blocks, the compiler takes the registers it wants (84: 5 waves per
// SIMD) -- the stage alone on the chip.  <kLeanShadeBlock, 8>: 256-thread blocks = one wave per SIMD at no more than 64
// VGPRs, the build that fits into the eighth wave slot of every SIMD while the other chunk's persistent traversal kernel
// holds seven (launch_extend / ptx_render_batch): the shading stage streams from HBM, the traversal kernel issues vector
// instructions out of L2 -- side by side they hide each other.
template <int kBlock, int kMinWaves>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kMinWaves)))
k_shade(SceneDev sc, FrameDev f, ImagesDev img, uint32_t depth, RayQueue rays_in, const HitOut *__restrict__ hits,
        uint32_t *__restrict__ counters, uint32_t capacity, RayQueue rays_out, float4 *__restrict__ results)
{''')
body=body.replace('kShadeBlock','kBlock').replace('<kBlock, 2>: 512-thread','<kShadeBlock, 2>: 512-thread')
rep=[
('''    __shared__ float s_lut[256];''','''    constexpr bool kStaged = kMinWaves >= 8;      // the 64-VGPR build that runs beside a traversal kernel
    constexpr bool kNtRec = %s;
    __shared__ float s_lut[256];''' % ("true" if NT_REC else "false")),
('''    const uint32_t n_live = counters[depth];
    if ((uint64_t)blockIdx.x * kBlock >= n_live) return;''','''    // Beside a traversal kernel these waves are the youngest on their SIMD, and vector issue goes by priority, then age
    // (MI355X_MICROARCH.md "Two waves per SIMD").  Their instruction count is 0.4 % of the traversal's: they go first.
    __builtin_amdgcn_s_setprio(3);
    const uint32_t n_live = counters[depth];
    if ((uint64_t)blockIdx.x * kBlock >= n_live) return;'''),
('''    for (uint32_t tile = blockIdx.x; (uint64_t)tile * kBlock < n_live; tile += gridDim.x) {''','''    __syncthreads();      // the tables are in LDS (once per block: no barrier sits between a tile's fetches and their use, so
                          // the compiler is free to place the record's loads where the register budget allows)
    for (uint32_t tile = blockIdx.x; (uint64_t)tile * kBlock < n_live; tile += gridDim.x) {'''),
('''        r1 = rays_in.dir[i];''','''        r1 = ld4<kStaged>(rays_in.dir + i);'''),
('''make_float4(f.cam_origin[0], f.cam_origin[1], f.cam_origin[2], __uint_as_float(i)) : rays_in.org[i];''','''make_float4(f.cam_origin[0], f.cam_origin[1], f.cam_origin[2], __uint_as_float(i)) : ld4<kStaged>(rays_in.org + i);'''),
('''make_float4(1.0f, 1.0f, 1.0f, 0.0f) : rays_in.thr[i];''','''make_float4(1.0f, 1.0f, 1.0f, 0.0f) : ld4<kStaged>(rays_in.thr + i);'''),
('''        hr = *reinterpret_cast<const float2 *>(hits + i);''','''        hr = ld2<kStaged>(reinterpret_cast<const float2 *>(hits + i));'''),
('''    float4 t0 = r0, t1 = r0, t2 = r0, s0 = r0, s1 = r0, s2 = r0, s3 = r0, s4 = r0;
    if (tri != 0xFFFFFFFFu) {
        const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
        t0 = sp[0]; t1 = sp[1]; t2 = sp[2]; s0 = sp[3]; s1 = sp[4]; s2 = sp[5]; s3 = sp[6]; s4 = sp[7];
    }
    __syncthreads();
''','''    // The 512-thread build fetches the whole 128-byte record here, ahead of everything that waits.  The 64-VGPR build
    // fetches it inside the hit branch, a part where it is first used (the parts after the first come out of the
    // line the first one brought in): 32 registers of record never have to be live at once.
    float4 t0 = r0, t1 = r0, t2 = r0, s0 = r0, s1 = r0, s2 = r0, s3 = r0, s4 = r0;
    if (!kStaged && tri != 0xFFFFFFFFu) {
        const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
        t0 = sp[0]; t1 = sp[1]; t2 = sp[2]; s0 = sp[3]; s1 = sp[4]; s2 = sp[5]; s3 = sp[6]; s4 = sp[7];
    }
'''),
('''            was_hit = true;
            // rchit.glsl:160-183
            f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
            // barycentrics of the hit, recomputed from the triangle (tri_uv == tri_test's u, v)
            float hu, hv;
            tri_uv(ro, rd, mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), hu, hv);
            float bx = 1.0f - hu - hv, by = hu, bz = hv;
            normal = normalize((n0 * bx + n1 * by) + n2 * bz);
''','''            was_hit = true;
            const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
            if (kStaged) { t0 = ld4<kNtRec>(sp); t1 = ld4<kNtRec>(sp + 1); t2 = ld4<kNtRec>(sp + 2); }
            // barycentrics of the hit, recomputed from the triangle (tri_uv == tri_test's u, v)
            float hu, hv;
            tri_uv(ro, rd, mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), hu, hv);
            float bx = 1.0f - hu - hv, by = hu, bz = hv;
            if (kStaged) { s0 = ld4<kNtRec>(sp + 3); s1 = ld4<kNtRec>(sp + 4); s2 = ld4<kNtRec>(sp + 5); }
            // rchit.glsl:160-183
            f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
            normal = normalize((n0 * bx + n1 * by) + n2 * bz);
            if (kStaged) { s3 = ld4<kNtRec>(sp + 6); s4 = ld4<kNtRec>(sp + 7); }
'''),
('''            results[path] = make_float4(thr.x, thr.y, thr.z, 1.0f);''','''            st4<kStaged>(results + path, make_float4(thr.x, thr.y, thr.z, 1.0f));'''),
('''        rays_out.org[slot] = o0; rays_out.dir[slot] = o1; rays_out.thr[slot] = o2;''','''        st4<kStaged>(rays_out.org + slot, o0); st4<kStaged>(rays_out.dir + slot, o1); st4<kStaged>(rays_out.thr + slot, o2);'''),
]
for o,n in rep:
    assert o in body, o
    body=body.replace(o,n)
s=s[:a]+body+s[b:]
old3='''    blocks = std::min(blocks, shade_grid_cap());
    hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(kShadeBlock), 0, s, sc, f, img, depth, rays_in, hits, counters,
                       capacity, rays_out, results);
}'''
new3='''    if (lean) {
        blocks = std::min((capacity + kLeanShadeBlock - 1u) / kLeanShadeBlock, shade_grid_cap() * (uint32_t)(kShadeBlock / kLeanShadeBlock));
        hipLaunchKernelGGL((k_shade<kLeanShadeBlock, 8>), dim3(blocks), dim3(kLeanShadeBlock), 0, s, sc, f, img, depth, rays_in, hits,
                           counters, capacity, rays_out, results);
        return;
    }
    blocks = std::min(blocks, shade_grid_cap());
    hipLaunchKernelGGL((k_shade<kShadeBlock, 2>), dim3(blocks), dim3(kShadeBlock), 0, s, sc, f, img, depth, rays_in, hits, counters,
                       capacity, rays_out, results);
}'''
assert old3 in s
s=s.replace(old3,new3)
s=s.replace('''                  RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                  float4 *results)
{
    uint32_t blocks = (capacity + kShadeBlock - 1u) / kShadeBlock;''','''                  RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                  float4 *results, bool lean)
{
    uint32_t blocks = (capacity + kShadeBlock - 1u) / kShadeBlock;''')
s=s.replace('''constexpr int kShadeBlock = PTX_SHADE_BLOCK;''','''constexpr int kLeanShadeBlock = 256;        // k_shade's co-resident build: one wave per SIMD (see k_shade)
constexpr int kShadeBlock = PTX_SHADE_BLOCK;''')
for old,new in [('''    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0) {      // the chunk's counters''','''    __builtin_amdgcn_s_setprio(3);      // streams beside the other chunk's traversal kernel (see k_shade)
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0) {      // the chunk's counters'''),('''    uint32_t pix = blockIdx.x * 256u + threadIdx.x;
    if (pix >= f.n_pixels) return;
    float4 cur = img.albedo[pix];''','''    __builtin_amdgcn_s_setprio(3);      // streams beside the other chunk's traversal kernel (see k_shade)
    uint32_t pix = blockIdx.x * 256u + threadIdx.x;
    if (pix >= f.n_pixels) return;
    float4 cur = img.albedo[pix];''')]:
    assert old in s
    s=s.replace(old,new)
open(p,'w').write(s)
p='/root/repo/path_tracer_amd/csrc/ptx_kernels.h'
s=open(p).read()
s=s.replace('''                  RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                  float4 *results);
// Lit integrator''','''                  RayQueue rays_in, const HitOut *hits, uint32_t *counters, uint32_t capacity, RayQueue rays_out,
                  float4 *results, bool lean = false /* the 64-VGPR, one-wave-per-SIMD build that runs beside a traversal kernel */);
// Lit integrator''')
open(p,'w').write(s)
p='/root/repo/path_tracer_amd/csrc/ptx_api.cpp'
s=open(p).read()
old='''        const uint32_t waves = (c->n_sets >= 2 && n_paths <= (4u << 20)) ? std::max(256u, c->extend_waves / 2u) : c->extend_waves;'''
new='''        // Larger chunks with another chunk in flight (16-spp batches): the traversal kernel leaves `coresident` of the eight
        // wave slots of every SIMD free and this chunk's shading launches use the 64-VGPR build of k_shade, which fits into
        // them beside the OTHER chunk's traversal kernel: the HBM-bound stage and the issue-bound stage overlap.
        const bool coresident = c->coresident > 0 && c->n_sets >= 2 && !c->count_traversal && !lit && n_paths > (4u << 20);
        const uint32_t waves = (c->n_sets >= 2 && n_paths <= (4u << 20)) ? std::max(256u, c->extend_waves / 2u)
                               : coresident ? std::max(256u, c->extend_waves - (uint32_t)c->coresident * (c->extend_waves / 8u)) : c->extend_waves;'''
assert old in s
s=s.replace(old,new)
old='''                launch_shade(st, sc, f, img, b, rin, ps.d_hits, ps.d_counters, n_paths, rout, ps.d_results);
            } else {'''
new='''                launch_shade(st, sc, f, img, b, rin, ps.d_hits, ps.d_counters, n_paths, rout, ps.d_results, coresident);
            } else {'''
assert old in s
s=s.replace(old,new)
old='''    uint32_t extend_waves = 4096;           // persistent grid of the traversal kernel'''
new='''    uint32_t extend_waves = 4096;           // persistent grid of the traversal kernel
    int coresident = 0;                     // PTX_CORESIDENT=k: wave slots per SIMD left to the other chunk's shading'''
assert old in s
s=s.replace(old,new)
old='''    if (const char *ew = getenv("PTX_EXTEND_WAVES"))'''
new='''    if (const char *cr = getenv("PTX_CORESIDENT")) c->coresident = std::max(0, std::min(4, atoi(cr)));
    if (const char *ew = getenv("PTX_EXTEND_WAVES"))'''
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
