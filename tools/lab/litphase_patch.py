"""Developer experiment: where a tile of k_shade_lit spends its time.  Patches csrc/ptx_kernels.hip IN PLACE with s_memtime
stamps at the joints of one tile (record arrived, taps issued / arrived, light record issued / arrived, end of the hit's
shading, the queue reservation, the stores), summed over the tiles a block walks, and ONE device printf per sampled wave after the
loop of a bounce-1 launch (a printf inside the loop stalls the wave for a host round trip).  Stamps 1, 3 and 5 drain the wave's vector-memory counter first, so that the waits show up where they are paid.
    python tools/lab/litphase_patch.py && python -m path_tracer_amd.build --variant litphase -DPTX_LAB_LITPHASE
    git checkout path_tracer_amd/csrc/ptx_kernels.hip          # the product sources never carry the patch
    gpurun -- 'PTX_LIB_DIR=$PWD/probe_libs/litphase python3 bench.py --integrator lit --steps 1 --warmup 0 --cpu-baseline-spp 0'
Result of round 3: DESIGN.md section 9."""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "path_tracer_amd/csrc/ptx_kernels.hip")
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old)
    s = s.replace(old, new)


rep("    float4 q0, q1, q2;\n};\n__device__ __forceinline__ void lit_shade_hit(",
    "    float4 q0, q1, q2;\n#ifdef PTX_LAB_LITPHASE\n    unsigned long long st[12];\n#endif\n};\n#ifdef PTX_LAB_LITPHASE\n"
    "#define LIT_STAMP(i, drain) do { if (drain) asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); S.st[i] = __builtin_amdgcn_s_memtime(); "
    "asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); } while (0)\n#else\n#define LIT_STAMP(i, drain) do { } while (0)\n#endif\n"
    "__device__ __forceinline__ void lit_shade_hit(")
rep("    was_hit = true;\n    const f3 va = mk3(t0.x, t0.y, t0.z)", "    was_hit = true;\n    LIT_STAMP(1, 1);\n    const f3 va = mk3(t0.x, t0.y, t0.z)")
rep("    if (tex != 0xFFFFFFFFu) {\n        f3 tx = tap_resolve(tap_b, s_lut);",
    "    LIT_STAMP(2, 0);\n    LIT_STAMP(3, 1);\n    if (tex != 0xFFFFFFFFu) {\n        f3 tx = tap_resolve(tap_b, s_lut);")
rep("            const float4 *lp = reinterpret_cast<const float4 *>(sc.lights + lo);\n            const float4 l0 = lp[0], l1 = lp[1], l2 = lp[2], l3 = lp[3];\n",
    "            LIT_STAMP(4, 0);\n            const float4 *lp = reinterpret_cast<const float4 *>(sc.lights + lo);\n"
    "            const float4 l0 = lp[0], l1 = lp[1], l2 = lp[2], l3 = lp[3];\n            LIT_STAMP(5, 1);\n")
rep("    if (!done && !(finite3(thr) && finite3(new_d))) done = true;\n}\n",
    "    if (!done && !(finite3(thr) && finite3(new_d))) done = true;\n    LIT_STAMP(6, 0);\n}\n")
rep("    TexTap tap_b, tap_n, tap_m;\n", "    LIT_STAMP(11, 0);\n    TexTap tap_b, tap_n, tap_m;\n")
# kernel side
rep("    uint32_t buf = 0;\n    for (uint32_t tile = blockIdx.x; (uint64_t)tile * kLitBlock < n_live; tile += gridDim.x, buf ^= 1u) {\n",
    "    uint32_t buf = 0;\n#ifdef PTX_LAB_LITPHASE\n    unsigned long long lab_sum[12], lab_prev_end = 0, lab_gap = 0; uint32_t lab_tiles = 0;\n    for (int i_ = 0; i_ < 12; ++i_) lab_sum[i_] = 0;\n#endif\n"
    "    for (uint32_t tile = blockIdx.x; (uint64_t)tile * kLitBlock < n_live; tile += gridDim.x, buf ^= 1u) {\n")
rep("    const uint32_t slot_in = tile * kLitBlock + threadIdx.x;\n    const bool valid = slot_in < capacity && slot_in < n_live;\n    const uint32_t tri = valid ? __float_as_uint(hr.y) : 0xFFFFFFFFu;\n",
    "#ifdef PTX_LAB_LITPHASE\n    unsigned long long lab_st[12];\n    for (int i_ = 0; i_ < 12; ++i_) lab_st[i_] = 0;\n    lab_st[10] = __builtin_amdgcn_s_memtime();\n#endif\n"
    "    const uint32_t slot_in = tile * kLitBlock + threadIdx.x;\n    const bool valid = slot_in < capacity && slot_in < n_live;\n    const uint32_t tri = valid ? __float_as_uint(hr.y) : 0xFFFFFFFFu;\n")
rep("    bool alive = false, was_hit = false, want_shadow = false;\n    float4 o0, o1, o2, q0, q1, q2;\n    if (valid) {\n        const f3 o = mk3(r0.x, r0.y, r0.z), dir = mk3(r1.x, r1.y, r1.z);\n        const uint32_t path = __float_as_uint(r0.w), uv_state = __float_as_uint(r1.w);\n        const uint32_t sample = path / f.n_pixels, pix = path - sample * f.n_pixels;\n        const uint32_t acc = f.mean_base + sample;\n        const uint32_t rng_acc",
    "#ifdef PTX_LAB_LITPHASE\n    lab_st[0] = __builtin_amdgcn_s_memtime();\n#endif\n    bool alive = false, was_hit = false, want_shadow = false;\n    float4 o0, o1, o2, q0, q1, q2;\n    if (valid) {\n        const f3 o = mk3(r0.x, r0.y, r0.z), dir = mk3(r1.x, r1.y, r1.z);\n        const uint32_t path = __float_as_uint(r0.w), uv_state = __float_as_uint(r1.w);\n        const uint32_t sample = path / f.n_pixels, pix = path - sample * f.n_pixels;\n        const uint32_t acc = f.mean_base + sample;\n        const uint32_t rng_acc")
rep("        LitState S;\n        S.thr = mk3(r2.x, r2.y, r2.z);",
    "        LitState S;\n#ifdef PTX_LAB_LITPHASE\n        for (int i_ = 0; i_ < 12; ++i_) S.st[i_] = 0;\n#endif\n        S.thr = mk3(r2.x, r2.y, r2.z);")
rep("        q0 = S.q0; q1 = S.q1; q2 = S.q2;\n        // radiance so far",
    "        q0 = S.q0; q1 = S.q1; q2 = S.q2;\n#ifdef PTX_LAB_LITPHASE\n        for (int i_ = 1; i_ <= 6; ++i_) lab_st[i_] = S.st[i_];\n        lab_st[11] = S.st[11];\n#endif\n        // radiance so far")
rep("    fetch_ray(tile + gridDim.x, r0, r1, r2, hr);      // this tile's ray registers are dead",
    "#ifdef PTX_LAB_LITPHASE\n    lab_st[7] = __builtin_amdgcn_s_memtime();\n#endif\n    fetch_ray(tile + gridDim.x, r0, r1, r2, hr);      // this tile's ray registers are dead")
rep("    if (alive) { rays_out.org[slot] = o0; rays_out.dir[slot] = o1; rays_out.thr[slot] = o2; }\n"
    "    if (want_shadow) { shadow.org[sslot] = q0; shadow.dir[sslot] = q1; shadow.thr[sslot] = q2; }\n    }\n",
    "#ifdef PTX_LAB_LITPHASE\n    lab_st[8] = __builtin_amdgcn_s_memtime();\n#endif\n"
    "    if (alive) { rays_out.org[slot] = o0; rays_out.dir[slot] = o1; rays_out.thr[slot] = o2; }\n"
    "    if (want_shadow) { shadow.org[sslot] = q0; shadow.dir[sslot] = q1; shadow.thr[sslot] = q2; }\n"
    "#ifdef PTX_LAB_LITPHASE\n    lab_st[9] = __builtin_amdgcn_s_memtime();\n"
    "    if (lab_prev_end) lab_gap += lab_st[10] - lab_prev_end;\n    lab_prev_end = lab_st[9]; ++lab_tiles;\n"
    "    lab_sum[0] += lab_st[0] - lab_st[10]; lab_sum[9] += lab_st[9] - lab_st[8]; lab_sum[8] += lab_st[8] - lab_st[7];\n"
    "    if (lab_st[1] && lab_st[2] && lab_st[3] && lab_st[4] && lab_st[5] && lab_st[6]) { for (int i_ = 1; i_ <= 6; ++i_) lab_sum[i_] += lab_st[i_] - lab_st[i_ - 1]; lab_sum[7] += lab_st[7] - lab_st[6]; lab_sum[11] += lab_st[11] - lab_st[1]; } else lab_sum[7] += lab_st[7] - lab_st[0];\n"
    "#endif\n    }\n"
    "#ifdef PTX_LAB_LITPHASE\n"
    "    if (depth == 1 && (blockIdx.x == 37 || blockIdx.x == 300 || blockIdx.x == 5000) && (threadIdx.x == 0 || threadIdx.x == 192) && n_live > 10000000u)\n"
    "        printf(\"LITPHASE blk %u thr %u tiles %u: gap %llu top %llu rec %llu uv %llu uv+issue %llu taps %llu mid %llu light %llu rest %llu post %llu reserve %llu stores %llu\\n\", blockIdx.x, threadIdx.x, lab_tiles,\n"
    "               lab_gap, lab_sum[0], lab_sum[1], lab_sum[11], lab_sum[2], lab_sum[3], lab_sum[4], lab_sum[5], lab_sum[6], lab_sum[7], lab_sum[8], lab_sum[9]);\n#endif\n")
open(p, "w").write(s)
print("patched", p)
