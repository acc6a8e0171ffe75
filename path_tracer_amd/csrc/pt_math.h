// pt_math.h -- fp32 vector helpers and the reference's shader arithmetic, written so that
// every expression is evaluated in source order without FMA contraction (the library is
// built with -ffp-contract=off).  Device code; PT_HD lets tests/native compile the same
// functions for the host to check the BVH builder and traversal without a GPU.
//
// Reference (all under /root/reference/src/shaders): shared.glsl:156-168,194-207 (hash, rand),
// :236-271 (create_basis, cosine_hemisphere_sample), rmiss.glsl:8-15 (sky_color).
#pragma once

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace ptx {

struct f3 { float x, y, z; };

PT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_HD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// GLSL normalize(): v * (1 / sqrt(dot(v, v))) with IEEE sqrt and divide
PT_HD f3 normalize(f3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
PT_HD float length(f3 a) { return sqrtf(dot(a, a)); }

PT_HD uint32_t f2u(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
PT_HD float u2f(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

PT_HD bool finite3(f3 a)
{
    // exponent field all ones <=> inf or NaN
    return ((f2u(a.x) & 0x7F800000u) != 0x7F800000u) && ((f2u(a.y) & 0x7F800000u) != 0x7F800000u) &&
           ((f2u(a.z) & 0x7F800000u) != 0x7F800000u);
}

// shared.glsl:156-161
PT_HD uint32_t hash(uint32_t state)
{
    state ^= state >> 16; state *= 0x7feb352dU;
    state ^= state >> 15; state *= 0x846ca68bU;
    state ^= state >> 16;
    return state;
}

// shared.glsl:164-168
PT_HD uint32_t hash2(uint32_t x, uint32_t y)
{
    x = hash(x);
    y = hash(y);
    return hash(x ^ (y * 7919u));
}

// shared.glsl:194-207 with the uv hash hoisted: `uv_state` = hash2(floatBitsToUint(uv)) is a
// per-pixel constant, so callers compute it once per path instead of once per rand().
PT_HD float rand_from(uint32_t uv_state, uint32_t depth, uint32_t accumulated_count, uint32_t seed)
{
    uint32_t state = hash(uv_state ^ depth);
    state = hash(state ^ accumulated_count);
    state = hash(state ^ seed);
    return (float)state / 4294967296.0f;   // RTE convert, exact power-of-two divide; 1.0 reachable
}

// shared.glsl:236-244 (sign(0) := 1)
PT_HD void create_basis(f3 n, f3 &tangent, f3 &bitangent)
{
    float sign_val = (n.z > 0.0f) ? 1.0f : ((n.z < 0.0f) ? -1.0f : 0.0f);
    if (n.z == 0.0f) sign_val = 1.0f;
    float a = -1.0f / (sign_val + n.z);
    float b = n.x * n.y * a;
    tangent = mk3(1.0f + sign_val * n.x * n.x * a, sign_val * b, -sign_val * n.x);
    bitangent = mk3(b, sign_val + n.y * n.y * a, -n.y);
}

// sin/cos of phi in [0, 2*pi] for shared.glsl:262-263.  GLSL leaves their precision to the
// implementation; the build pins one definition (3-term Cody-Waite reduction + Cephes minimax
// polynomials, no FMA) so that a path is bit-reproducible on any host or device.
PT_HD void sincos_2pi(float phi, float &s, float &c)
{
    int k = (int)(phi * 0.636619772f + 0.5f);
    float fk = (float)k;
    float r = ((phi - fk * 1.5703125f) - fk * 4.83751296997e-4f) - fk * 7.549789954e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    int q = k & 3;
    float s0 = (q & 1) ? cp : sp;
    float c0 = (q & 1) ? sp : cp;
    s = (q & 2) ? -s0 : s0;
    c = (q == 1 || q == 2) ? -c0 : c0;
}

// shared.glsl:251-271
PT_HD f3 cosine_hemisphere_sample(uint32_t uv_state, uint32_t depth, uint32_t acc, uint32_t seed, f3 n)
{
    float u1 = rand_from(uv_state, depth, acc, seed + 7919u);
    float u2 = rand_from(uv_state, depth, acc, seed + 16381u);
    float r = sqrtf(u1);
    float phi = 6.283185307f * u2;
    float sn, cs;
    sincos_2pi(phi, sn, cs);
    float x = r * cs;
    float y = r * sn;
    float om = 1.0f - u1;
    float z = sqrtf(0.0f < om ? om : 0.0f);
    f3 t, b;
    create_basis(n, t, b);
    return (t * x + b * y) + n * z;
}

// rmiss.glsl:8-15; direction is not normalised
PT_HD f3 sky_color(f3 d)
{
    if (d.y > 0.0f) {
        float t = d.y, om = 1.0f - t;
        return mk3(1.0f * om + 0.25f * t, 1.0f * om + 0.5f * t, 1.0f * om + 1.0f * t);
    }
    return mk3(0.03f, 0.03f, 0.03f);
}

// Ray/triangle test.  The reference's lives in the Vulkan driver (rgen.glsl:66-78), so the
// build defines it: edge functions are scalar triple products of origin-relative vertices
// (a shared edge evaluates to exact negatives in its two triangles => watertight), both
// faces accepted (no cull flags, renderer.rs:944-954), t_min < t < t_max, (u,v) = weights
// of vertex 1 and 2.  The oracle evaluates the identical expression tree.
PT_HD bool tri_test(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float t_min, float t_max, float &t_out, float &u_out,
                    float &v_out)
{
    f3 A = v0 - o, B = v1 - o, C = v2 - o;
    float U = dot(d, cross(C - B, C + B));
    float V = dot(d, cross(A - C, A + C));
    float W = dot(d, cross(B - A, B + A));
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    f3 Ng = cross(B - A, C - A);
    float den = dot(d, Ng);
    float t = dot(A, Ng) / den;
    if (!(t > t_min && t < t_max)) return false;
    t_out = t; u_out = V / det; v_out = W / det;
    return true;
}

// (u, v) of an accepted hit, by the same expression tree as tri_test: the traversal kernel keeps
// only (t, triangle) per ray and the shading stage recomputes the barycentrics once per ray
// instead of dividing in every triangle test.
PT_HD void tri_uv(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float &u_out, float &v_out)
{
    f3 A = v0 - o, B = v1 - o, C = v2 - o;
    float U = dot(d, cross(C - B, C + B));
    float V = dot(d, cross(A - C, A + C));
    float W = dot(d, cross(B - A, B + A));
    float det = (U + V) + W;
    u_out = V / det; v_out = W / det;
}

// tri_test without the barycentrics (same U, V, W, det, t and the same accept / reject decisions)
PT_HD bool tri_test_t(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float t_min, float t_max, float &t_out)
{
    f3 A = v0 - o, B = v1 - o, C = v2 - o;
    float U = dot(d, cross(C - B, C + B));
    float V = dot(d, cross(A - C, A + C));
    float W = dot(d, cross(B - A, B + A));
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    f3 Ng = cross(B - A, C - A);
    float den = dot(d, Ng);
    float t = dot(A, Ng) / den;
    if (!(t > t_min && t < t_max)) return false;
    t_out = t;
    return true;
}

// closest hit; equal t -> lowest (instance, primitive)
PT_HD bool hit_better(float t, uint32_t inst, uint32_t prim, float bt, uint32_t bi, uint32_t bp)
{
    if (t < bt) return true;
    if (t > bt) return false;
    return (inst < bi) || (inst == bi && prim < bp);
}

// ---------------------------------------------------------------------------------------------
// Lit integrator (build extension): BSDF pieces.  Specified by
// oracle/pt_oracle.c (raygen_pixel_lit and its helpers); same fp32 expression trees here.
constexpr uint32_t kLitSeedLobe = 101u, kLitSeedMix = 151u, kLitSeedLight = 211u, kLitSeedLu = 307u,
                   kLitSeedLv = 401u, kLitSeedRr = 503u, kLitSeedU1 = 7919u, kLitSeedU2 = 16381u;
constexpr float kLitInvPi = 0.318309886f;
constexpr float kLitShadowTmax = 0.999f;

PT_HD float ggx_d(float noh, float a2)
{
    float d = (noh * noh) * (a2 - 1.0f) + 1.0f;
    return a2 / ((3.14159265f * d) * d);
}
PT_HD float smith_g1(float nox, float a2)
{
    return (2.0f * nox) / (nox + sqrtf(a2 + (1.0f - a2) * (nox * nox)));
}
PT_HD float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// power heuristic (beta = 2) of the strategy with density a against the one with density b
PT_HD float lit_mis_weight(float a, float b)
{
    float a2 = a * a, b2 = b * b;
    float w = a2 / (a2 + b2);
    if (!(w >= 0.0f)) w = 0.0f;
    return w;
}

// f (rgb) and the mixture pdf of the opaque BSDF (Lambert + GGX) for unit wo, wi on the side of n
PT_HD bool bsdf_eval(f3 n, f3 wo, f3 wi, f3 diff, f3 f0, float a2, float p_spec, f3 &f, float &pdf)
{
    float nol = dot(n, wi), nov = dot(n, wo);
    if (!(nol > 0.0f && nov > 0.0f)) return false;
    f3 h = normalize(wo + wi);
    float noh = dot(n, h), voh = dot(wo, h);
    float dd = ggx_d(noh, a2), g1v = smith_g1(nov, a2), g1l = smith_g1(nol, a2);
    float x = 1.0f - voh;
    if (x < 0.0f) x = 0.0f;
    float fr = pow5(x);
    float spec = (dd * (g1v * g1l)) / ((4.0f * nov) * nol);
    f.x = diff.x * kLitInvPi + (f0.x + (1.0f - f0.x) * fr) * spec;
    f.y = diff.y * kLitInvPi + (f0.y + (1.0f - f0.y) * fr) * spec;
    f.z = diff.z * kLitInvPi + (f0.z + (1.0f - f0.z) * fr) * spec;
    pdf = p_spec * ((dd * g1v) / (4.0f * nov)) + (1.0f - p_spec) * (nol * kLitInvPi);
    return true;
}

// Heitz 2018 visible-normal sampling of GGX: half vector in world space
PT_HD f3 ggx_sample_vndf(f3 n, f3 wo, float alpha, float u1, float u2)
{
    f3 t, b;
    create_basis(n, t, b);
    f3 vl = mk3(dot(wo, t), dot(wo, b), dot(wo, n));
    f3 vh = normalize(mk3(alpha * vl.x, alpha * vl.y, vl.z));
    float lensq = vh.x * vh.x + vh.y * vh.y;
    f3 t1 = mk3(1.0f, 0.0f, 0.0f);
    if (lensq > 0.0f) { float il = 1.0f / sqrtf(lensq); t1 = mk3(-vh.y * il, vh.x * il, 0.0f); }
    f3 t2 = cross(vh, t1);
    float r = sqrtf(u1), phi = 6.283185307f * u2, sn, cs;
    sincos_2pi(phi, sn, cs);
    float p1 = r * cs, p2 = r * sn;
    float sw = 0.5f * (1.0f + vh.z);
    float om = 1.0f - p1 * p1;
    p2 = (1.0f - sw) * sqrtf(0.0f < om ? om : 0.0f) + sw * p2;
    float oz = (1.0f - p1 * p1) - p2 * p2;
    float pz = sqrtf(0.0f < oz ? oz : 0.0f);
    f3 nh = (t1 * p1 + t2 * p2) + vh * pz;
    f3 ne = normalize(mk3(alpha * nh.x, alpha * nh.y, 0.0f < nh.z ? nh.z : 0.0f));
    return (t * ne.x + b * ne.y) + n * ne.z;
}

}  // namespace ptx
