/*
 * pt_oracle.c -- CPU restatement of AHL00/path_tracer's render hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  PARITY STATUS: "parity
 * unpinned" at every third-party boundary (driver traversal, glam, GLSL
 * sin/cos, sampler defaults); the reference ships no golden vectors.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC -pthread
 * Every fp32 expression here is evaluated in source order with no FMA
 * contraction; the HIP kernels evaluate the same expressions in the same
 * order, which is what makes whole-image bit-exact comparison possible.
 *
 * Reference files restated (all under /root/reference/src):
 *   shaders/shared.glsl:156-168,194-207,236-271   hash / rand / sampling
 *   shaders/rgen.glsl:20-141                      ray generation, path loop, accumulate
 *   shaders/rchit.glsl:158-254                    closest-hit shading
 *   shaders/rmiss.glsl:8-34                       sky
 *   renderer.rs:912-1031,1416-1423                instances, camera, uniforms
 *   scene/mod.rs:557-580                          Transform
 *   graphics.rs:492-516                           texel format
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* GLSL normalize(): v * (1/sqrt(dot(v,v))) -- one correctly rounded sqrt and divide */
static inline v3 v3_normalize(v3 a)
{
    float inv = 1.0f / sqrtf(v3_dot(a, a));
    return v3_scale(a, inv);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }

/* ------------------------------------------------------------------ */
/* shared.glsl:156-161                                                */
uint32_t orc_hash(uint32_t state)
{
    state ^= state >> 16; state *= 0x7feb352dU;
    state ^= state >> 15; state *= 0x846ca68bU;
    state ^= state >> 16;
    return state;
}

/* shared.glsl:164-168 */
uint32_t orc_hash2(uint32_t x, uint32_t y)
{
    x = orc_hash(x);
    y = orc_hash(y);
    return orc_hash(x ^ (y * 7919u));
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* shared.glsl:194-207.  float(state)/4294967296.0: RTE conversion, exact
 * power-of-two divide; 1.0 is reachable for state >= 0xFFFFFF80. */
float orc_rand(float uvx, float uvy, uint32_t depth, uint32_t accumulated_count, uint32_t seed)
{
    uint32_t state = orc_hash2(f2u(uvx), f2u(uvy));
    state = orc_hash(state ^ depth);
    state = orc_hash(state ^ accumulated_count);
    state = orc_hash(state ^ seed);
    return (float)state / 4294967296.0f;
}

/* shared.glsl:236-244 (Duff et al. ONB, sign(0) := 1) */
static void create_basis(v3 n, v3 *tangent, v3 *bitangent)
{
    float sign_val = (n.z > 0.0f) ? 1.0f : ((n.z < 0.0f) ? -1.0f : 0.0f);
    if (n.z == 0.0f) sign_val = 1.0f;
    float a = -1.0f / (sign_val + n.z);
    float b = n.x * n.y * a;
    *tangent = v3_make(1.0f + sign_val * n.x * n.x * a, sign_val * b, -sign_val * n.x);
    *bitangent = v3_make(b, sign_val + n.y * n.y * a, -n.y);
}

void orc_create_basis(const float n[3], float t[3], float b[3])
{
    v3 tt, bb;
    create_basis(v3_make(n[0], n[1], n[2]), &tt, &bb);
    t[0] = tt.x; t[1] = tt.y; t[2] = tt.z;
    b[0] = bb.x; b[1] = bb.y; b[2] = bb.z;
}

/*
 * sin/cos for shared.glsl:262-263.  GLSL leaves sin()/cos() precision to the
 * implementation (Vulkan: abs error 2^-11 on [-pi,pi]), so the build fixes one
 * definition used by the oracle and the HIP kernels alike: 3-term Cody-Waite
 * reduction to [-pi/4,pi/4] and the Cephes single-precision minimax
 * polynomials, evaluated without FMA.  Valid for 0 <= phi <= ~8 (phi = 2*pi*u2).
 */
void orc_sincos(float phi, float *s_out, float *c_out)
{
    int k = (int)(phi * 0.636619772f + 0.5f);      /* round(phi / (pi/2)) */
    float fk = (float)k;
    float r = ((phi - fk * 1.5703125f) - fk * 4.83751296997e-4f) - fk * 7.549789954e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    float s, c;
    switch (k & 3) {
    case 0: s = sp; c = cp; break;
    case 1: s = cp; c = -sp; break;
    case 2: s = -sp; c = -cp; break;
    default: s = -cp; c = sp; break;
    }
    *s_out = s; *c_out = c;
}

/* shared.glsl:251-271 */
static v3 cosine_hemisphere_sample(float uvx, float uvy, uint32_t depth, uint32_t acc, uint32_t seed, v3 n)
{
    float u1 = orc_rand(uvx, uvy, depth, acc, seed + 7919u);
    float u2 = orc_rand(uvx, uvy, depth, acc, seed + 16381u);
    float r = sqrtf(u1);
    float phi = 6.283185307f * u2;               /* 2.0 * 3.1415926535 folded to fp32 */
    float sn, cs;
    orc_sincos(phi, &sn, &cs);
    float x = r * cs;
    float y = r * sn;
    float om = 1.0f - u1;
    float z = sqrtf(0.0f < om ? om : 0.0f);      /* max(0.0, 1.0 - u1) */
    v3 t, b;
    create_basis(n, &t, &b);
    /* mat3(t,b,n) * local = t*x + b*y + n*z */
    return v3_add(v3_add(v3_scale(t, x), v3_scale(b, y)), v3_scale(n, z));
}

void orc_cosine_hemisphere_sample(float uvx, float uvy, uint32_t depth, uint32_t acc, uint32_t seed,
                                  const float n[3], float out[3])
{
    v3 r = cosine_hemisphere_sample(uvx, uvy, depth, acc, seed, v3_make(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* rmiss.glsl:8-15 ; mix(a,b,t) = a*(1-t) + b*t, direction NOT normalised */
static v3 sky_color(v3 d)
{
    if (d.y > 0.0f) {
        float t = d.y, om = 1.0f - t;
        return v3_make(1.0f * om + 0.25f * t, 1.0f * om + 0.5f * t, 1.0f * om + 1.0f * t);
    }
    return v3_make(0.03f, 0.03f, 0.03f);
}

void orc_sky_color(const float dir[3], float out[3])
{
    v3 c = sky_color(v3_make(dir[0], dir[1], dir[2]));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

/* IEC 61966-2-1 decode as done by sampling an R8G8B8A8_SRGB image (graphics.rs:494-495) */
static float g_srgb_lut[256];
static pthread_once_t g_lut_once = PTHREAD_ONCE_INIT;
static void init_lut(void)
{
    for (int i = 0; i < 256; ++i) {
        double c = i / 255.0;
        double l = (c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
        g_srgb_lut[i] = (float)l;
    }
}
float orc_srgb_to_linear(uint32_t c)
{
    pthread_once(&g_lut_once, init_lut);
    return g_srgb_lut[c & 255u];
}

/* ------------------------------------------------------------------ */
/* Scene: instances flattened to world space (stands in for the TLAS/BLAS
 * the reference builds in the driver, renderer.rs:1196-1388).            */

typedef struct { v3 a, b, c; uint32_t instance, prim; } wtri;
typedef struct { float lo[3], hi[3]; uint32_t left_first, count; } bnode;   /* count==0: inner, children left_first, left_first+1 */

/* lit integrator: one emissive triangle of the light table */
typedef struct { v3 a, b, c, n, le; float cdf, pdf_area; float uv[3][2]; uint32_t etex; /* emissive texture + 1 */ } light_t;

struct orc_scene {
    orc_scene_desc d;
    wtri *tris;  uint64_t n_tris;
    uint32_t *order;             /* BVH leaf order -> tri index */
    bnode *nodes; uint32_t n_nodes;
    uint64_t *inst_first;        /* first flattened triangle of each instance */
    float (*cof)[9];             /* per instance: cofactor matrix of the 3x3 transform, sign of det folded in */
    light_t *lights; uint64_t n_lights;
    float *mat_pdf_area;      /* per material: area density of the light pick for a point on a triangle of it */
};

void orc_default_params(orc_params *p)
{
    memset(p, 0, sizeof *p);
    p->max_depth = 16; p->t_min = 0.001f; p->t_max = 10000.0f; p->normal_eps = 0.001f;
}

static inline v3 xform_point(const float m[3][4], v3 p)
{
    return v3_make(((m[0][0] * p.x + m[0][1] * p.y) + m[0][2] * p.z) + m[0][3],
                   ((m[1][0] * p.x + m[1][1] * p.y) + m[1][2] * p.z) + m[1][3],
                   ((m[2][0] * p.x + m[2][1] * p.y) + m[2][2] * p.z) + m[2][3]);
}

/*
 * Ray/triangle test.  The reference's is inside the Vulkan driver (rgen.glsl:66-78),
 * so the build defines one: edge functions as scalar triple products of the
 * origin-relative vertices (each shared edge evaluates to exact negatives in
 * the two triangles that share it => watertight), both faces accepted (no
 * cull flags, renderer.rs:944-954), t_min < t < t_max, barycentrics
 * (u,v) = weights of vertex 1 and 2 as in VK_KHR_ray_tracing hit attributes.
 */
static inline int tri_test(v3 o, v3 d, const wtri *tr, float t_min, float t_max,
                           float *t_out, float *u_out, float *v_out)
{
    v3 A = v3_sub(tr->a, o), B = v3_sub(tr->b, o), C = v3_sub(tr->c, o);
    float U = v3_dot(d, v3_cross(v3_sub(C, B), v3_add(C, B)));
    float V = v3_dot(d, v3_cross(v3_sub(A, C), v3_add(A, C)));
    float W = v3_dot(d, v3_cross(v3_sub(B, A), v3_add(B, A)));
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    float det = (U + V) + W;
    if (det == 0.0f) return 0;
    v3 Ng = v3_cross(v3_sub(B, A), v3_sub(C, A));
    float den = v3_dot(d, Ng);
    float t = v3_dot(A, Ng) / den;
    if (!(t > t_min && t < t_max)) return 0;
    *t_out = t; *u_out = V / det; *v_out = W / det;
    return 1;
}

/* closest hit, ties on t -> lowest (instance, prim) */
static inline int hit_better(float t, uint32_t inst, uint32_t prim, float bt, uint32_t bi, uint32_t bp)
{
    if (t < bt) return 1;
    if (t > bt) return 0;
    return (inst < bi) || (inst == bi && prim < bp);
}

static inline int finite3(v3 a)
{
    return isfinite(a.x) && isfinite(a.y) && isfinite(a.z);
}

typedef struct { float t, u, v; uint32_t instance, prim; } hit_rec;

static int trace_brute(const orc_scene *s, v3 o, v3 d, float t_min, float t_max, hit_rec *h, orc_stats *st)
{
    int found = 0;
    float bt = t_max; uint32_t bi = 0xFFFFFFFFu, bp = 0xFFFFFFFFu; float bu = 0, bv = 0;
    for (uint64_t i = 0; i < s->n_tris; ++i) {
        float t, u, v;
        const wtri *tr = &s->tris[i];
        if (tri_test(o, d, tr, t_min, t_max, &t, &u, &v) && hit_better(t, tr->instance, tr->prim, bt, bi, bp)) {
            bt = t; bi = tr->instance; bp = tr->prim; bu = u; bv = v; found = 1;
        }
    }
    if (st) st->tris_tested += s->n_tris;
    if (found) { h->t = bt; h->u = bu; h->v = bv; h->instance = bi; h->prim = bp; }
    return found;
}

/* conservative slab test (NaN slabs are ignored by fminf/fmaxf); returns entry distance or -1 */
static inline int slab(const bnode *n, const float org[3], const float inv[3], float t_min, float bt, float *tnear)
{
    float tn = t_min, tf = bt;
    for (int a = 0; a < 3; ++a) {
        float t0 = (n->lo[a] - org[a]) * inv[a], t1 = (n->hi[a] - org[a]) * inv[a];
        float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
        lo -= fabsf(lo) * 1e-6f; hi += fabsf(hi) * 1e-6f;
        tn = fmaxf(tn, lo); tf = fminf(tf, hi);
    }
    *tnear = tn;
    return tn <= tf;
}

static int trace_bvh(const orc_scene *s, v3 o, v3 d, float t_min, float t_max, hit_rec *h, orc_stats *st)
{
    int found = 0;
    float bt = t_max; uint32_t bi = 0xFFFFFFFFu, bp = 0xFFFFFFFFu; float bu = 0, bv = 0;
    float inv[3] = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    float org[3] = { o.x, o.y, o.z };
    uint32_t stack[256]; float stack_t[256]; int sp = 0;
    float tn0;
    if (!slab(&s->nodes[0], org, inv, t_min, bt, &tn0)) return 0;
    stack[sp] = 0; stack_t[sp++] = tn0;
    while (sp) {
        --sp;
        if (stack_t[sp] > bt) continue;      /* entry beyond the best hit (ties still visited) */
        const bnode *n = &s->nodes[stack[sp]];
        if (st) st->nodes_visited++;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; ++i) {
                const wtri *tr = &s->tris[s->order[n->left_first + i]];
                float t, u, v;
                if (tri_test(o, d, tr, t_min, t_max, &t, &u, &v) && hit_better(t, tr->instance, tr->prim, bt, bi, bp)) {
                    bt = t; bi = tr->instance; bp = tr->prim; bu = u; bv = v; found = 1;
                }
            }
            if (st) st->tris_tested += n->count;
        } else {
            float ta, tb;
            int ha = slab(&s->nodes[n->left_first], org, inv, t_min, bt, &ta);
            int hb = slab(&s->nodes[n->left_first + 1], org, inv, t_min, bt, &tb);
            if (sp + 2 > 256) abort();
            if (ha && hb) {
                uint32_t near_i = n->left_first, far_i = n->left_first + 1;
                if (tb < ta) { near_i = n->left_first + 1; far_i = n->left_first; float tt = ta; ta = tb; tb = tt; }
                stack[sp] = far_i; stack_t[sp++] = tb;
                stack[sp] = near_i; stack_t[sp++] = ta;
            } else if (ha) { stack[sp] = n->left_first; stack_t[sp++] = ta; }
            else if (hb) { stack[sp] = n->left_first + 1; stack_t[sp++] = tb; }
        }
    }
    if (found) { h->t = bt; h->u = bu; h->v = bv; h->instance = bi; h->prim = bp; }
    return found;
}

static int trace(const orc_scene *s, v3 o, v3 d, float t_min, float t_max, uint32_t flags, hit_rec *h, orc_stats *st)
{
    /* rays with a non-finite origin or direction hit nothing (defined by the build) */
    if (!finite3(o) || !finite3(d)) return 0;
    if ((flags & 1u) || !s->nodes) return trace_brute(s, o, d, t_min, t_max, h, st);
    return trace_bvh(s, o, d, t_min, t_max, h, st);
}

int orc_trace(const orc_scene *s, const float org[3], const float dir[3], float t_min, float t_max,
              uint32_t flags, float *t, float *u, float *v, uint32_t *instance, uint32_t *prim)
{
    hit_rec h;
    if (!trace(s, v3_make(org[0], org[1], org[2]), v3_make(dir[0], dir[1], dir[2]), t_min, t_max, flags, &h, NULL))
        return 0;
    *t = h.t; *u = h.u; *v = h.v; *instance = h.instance; *prim = h.prim;
    return 1;
}

/* --- oracle BVH2: median split on the longest centroid axis, leaves <= 4 --- */
typedef struct { float lo[3], hi[3], c[3]; } tbox;

static void box_of(const wtri *t, tbox *b)
{
    const float *p[3] = { &t->a.x, &t->b.x, &t->c.x };
    for (int a = 0; a < 3; ++a) {
        float lo = p[0][a], hi = p[0][a];
        for (int k = 1; k < 3; ++k) { if (p[k][a] < lo) lo = p[k][a]; if (p[k][a] > hi) hi = p[k][a]; }
        b->lo[a] = lo; b->hi[a] = hi; b->c[a] = 0.5f * lo + 0.5f * hi;
    }
}

typedef struct { orc_scene *s; tbox *boxes; } bctx;

static void build_rec(bctx *c, uint32_t node, uint32_t first, uint32_t count)
{
    orc_scene *s = c->s;
    bnode *n = &s->nodes[node];
    float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int a = 0; a < 3; ++a) { n->lo[a] = INFINITY; n->hi[a] = -INFINITY; }
    for (uint32_t i = 0; i < count; ++i) {
        const tbox *b = &c->boxes[s->order[first + i]];
        for (int a = 0; a < 3; ++a) {
            if (b->lo[a] < n->lo[a]) n->lo[a] = b->lo[a];
            if (b->hi[a] > n->hi[a]) n->hi[a] = b->hi[a];
            if (b->c[a] < clo[a]) clo[a] = b->c[a];
            if (b->c[a] > chi[a]) chi[a] = b->c[a];
        }
    }
    /* pad: the slab test must never reject a box whose triangle the test above accepts */
    for (int a = 0; a < 3; ++a) {
        float m = fmaxf(fabsf(n->lo[a]), fabsf(n->hi[a]));
        float pad = m * 1e-5f + 1e-6f;
        n->lo[a] -= pad; n->hi[a] += pad;
    }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    if (count <= 4 || !(chi[axis] > clo[axis])) {
        if (count <= 64) { n->left_first = first; n->count = count; return; }
        /* many coincident centroids: split the range in half */
        uint32_t half = count / 2, l = s->n_nodes; s->n_nodes += 2;
        n->left_first = l; n->count = 0;
        build_rec(c, l, first, half); build_rec(c, l + 1, first + half, count - half);
        return;
    }
    float split = 0.5f * clo[axis] + 0.5f * chi[axis];
    uint32_t i = first, j = first + count;
    while (i < j) {
        if (c->boxes[s->order[i]].c[axis] < split) ++i;
        else { --j; uint32_t t = s->order[i]; s->order[i] = s->order[j]; s->order[j] = t; }
    }
    uint32_t nl = i - first;
    if (nl == 0 || nl == count) nl = count / 2;
    uint32_t l = s->n_nodes; s->n_nodes += 2;
    n->left_first = l; n->count = 0;
    build_rec(c, l, first, nl);
    build_rec(c, l + 1, first + nl, count - nl);
}

/* Build extension (ORC_SCENE_WORLD_NORMALS): n_world ~ cofactor(M3) * n, with the sign of det(M3)
 * folded in (= det * inverse transpose).  Rows of the cofactor matrix are cross products of M3's rows. */
static void normal_matrix(const float m[3][4], float c[9])
{
    v3 r0 = v3_make(m[0][0], m[0][1], m[0][2]), r1 = v3_make(m[1][0], m[1][1], m[1][2]), r2 = v3_make(m[2][0], m[2][1], m[2][2]);
    v3 c0 = v3_cross(r1, r2), c1 = v3_cross(r2, r0), c2 = v3_cross(r0, r1);
    float det = v3_dot(r0, c0);
    float sg = det < 0.0f ? -1.0f : 1.0f;
    c[0] = c0.x * sg; c[1] = c0.y * sg; c[2] = c0.z * sg;
    c[3] = c1.x * sg; c[4] = c1.y * sg; c[5] = c1.z * sg;
    c[6] = c2.x * sg; c[7] = c2.y * sg; c[8] = c2.z * sg;
}

static inline v3 normal_to_world(const float c[9], v3 n)
{
    return v3_make((c[0] * n.x + c[1] * n.y) + c[2] * n.z, (c[3] * n.x + c[4] * n.y) + c[5] * n.z,
                   (c[6] * n.x + c[7] * n.y) + c[8] * n.z);
}

static const orc_material_ext g_no_ext = { { 0.0f, 0.0f, 0.0f }, 0.0f, 1.5f, 0, 0, 0 };
static inline const orc_material_ext *material_ext(const orc_scene *s, uint32_t m)
{
    return s->d.material_ext ? &s->d.material_ext[m] : &g_no_ext;
}

/* Lit integrator: the light table = every triangle whose material emits, in (instance, primitive)
 * order; picked with probability proportional to area x luminance. */
static int light_candidate(const orc_scene *s, const orc_material_ext *e, uint64_t k, v3 *cr, float *len, float *area, float *power)
{
    if (!(e->emissive[0] > 0.0f || e->emissive[1] > 0.0f || e->emissive[2] > 0.0f)) return 0;
    const wtri *t = &s->tris[k];
    *cr = v3_cross(v3_sub(t->b, t->a), v3_sub(t->c, t->a));
    *len = sqrtf(v3_dot(*cr, *cr));
    *area = 0.5f * *len;
    if (!(*area > 0.0f) || !isfinite(*area)) return 0;
    float lum = (0.2126f * e->emissive[0] + 0.7152f * e->emissive[1]) + 0.0722f * e->emissive[2];
    *power = *area * lum;
    return 1;
}

static void build_lights(orc_scene *s)
{
    s->lights = NULL; s->n_lights = 0; s->mat_pdf_area = NULL;
    if (!s->d.material_ext) return;
    /* first pass: count and total power (double); second pass: fill */
    uint64_t n = 0, k = 0;
    double total = 0.0;
    v3 cr; float len, area, power;
    for (uint64_t i = 0; i < s->d.n_instances; ++i) {
        const orc_instance *in = &s->d.instances[i];
        const orc_material_ext *e = material_ext(s, in->off.material_offset);
        uint32_t np = in->off.index_count / 3;
        for (uint32_t p = 0; p < np; ++p, ++k)
            if (light_candidate(s, e, k, &cr, &len, &area, &power)) { total += (double)power; ++n; }
    }
    if (!n || !(total > 0.0)) return;
    s->lights = (light_t *)malloc(sizeof(light_t) * n);
    double run = 0.0;
    uint64_t li = 0;
    k = 0;
    for (uint64_t i = 0; i < s->d.n_instances; ++i) {
        const orc_instance *in = &s->d.instances[i];
        const orc_material_ext *e = material_ext(s, in->off.material_offset);
        uint32_t np = in->off.index_count / 3;
        for (uint32_t p = 0; p < np; ++p, ++k) {
            if (!light_candidate(s, e, k, &cr, &len, &area, &power)) continue;
            const wtri *t = &s->tris[k];
            run += (double)power;
            light_t *l = &s->lights[li++];
            l->a = t->a; l->b = t->b; l->c = t->c;
            l->n = v3_scale(cr, 1.0f / len);
            l->le = v3_make(e->emissive[0], e->emissive[1], e->emissive[2]);
            l->etex = e->emissive_texture;
            {
                uint64_t ib = (uint64_t)in->off.index_offset + (uint64_t)p * 3;
                for (int c3 = 0; c3 < 3; ++c3) {
                    const orc_vertex *vx = &s->d.vertices[in->off.vertex_offset + s->d.indices[ib + c3]];
                    l->uv[c3][0] = vx->uv[0]; l->uv[c3][1] = vx->uv[1];
                }
            }
            l->cdf = (float)(run / total);
            l->pdf_area = (float)((double)power / total) / area;
        }
    }
    s->lights[li - 1].cdf = 1.0f;
    s->n_lights = li;
    /* the same density seen from the other side (a BSDF-sampled ray that lands on an emitter): pick probability /
     * area = luminance / total power, one value per material */
    s->mat_pdf_area = (float *)calloc(s->d.n_materials ? s->d.n_materials : 1, sizeof(float));
    for (uint64_t m = 0; m < s->d.n_materials; ++m) {
        const orc_material_ext *e = &s->d.material_ext[m];
        if (!(e->emissive[0] > 0.0f || e->emissive[1] > 0.0f || e->emissive[2] > 0.0f)) continue;
        float lum = (0.2126f * e->emissive[0] + 0.7152f * e->emissive[1]) + 0.0722f * e->emissive[2];
        s->mat_pdf_area[m] = (float)((double)lum / total);
    }
}

uint64_t orc_scene_light_count(const orc_scene *s) { return s->n_lights; }
void orc_scene_light(const orc_scene *s, uint64_t i, float verts[9], float le[3], float *cdf, float *pdf_area)
{
    const light_t *l = &s->lights[i];
    verts[0] = l->a.x; verts[1] = l->a.y; verts[2] = l->a.z; verts[3] = l->b.x; verts[4] = l->b.y; verts[5] = l->b.z;
    verts[6] = l->c.x; verts[7] = l->c.y; verts[8] = l->c.z;
    le[0] = l->le.x; le[1] = l->le.y; le[2] = l->le.z; *cdf = l->cdf; *pdf_area = l->pdf_area;
}

orc_scene *orc_scene_create(const orc_scene_desc *desc)
{
    pthread_once(&g_lut_once, init_lut);
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    s->d = *desc;
    uint64_t nt = 0;
    for (uint64_t i = 0; i < desc->n_instances; ++i) nt += desc->instances[i].off.index_count / 3;
    s->n_tris = nt;
    s->tris = (wtri *)malloc(sizeof(wtri) * (nt ? nt : 1));
    s->inst_first = (uint64_t *)malloc(sizeof(uint64_t) * (desc->n_instances ? desc->n_instances : 1));
    s->cof = (float (*)[9])malloc(sizeof(float[9]) * (desc->n_instances ? desc->n_instances : 1));
    uint64_t k = 0;
    for (uint64_t i = 0; i < desc->n_instances; ++i) {
        const orc_instance *in = &desc->instances[i];
        uint32_t np = in->off.index_count / 3;
        s->inst_first[i] = k;
        normal_matrix(in->xform, s->cof[i]);
        for (uint32_t p = 0; p < np; ++p) {
            const uint32_t *ix = &desc->indices[in->off.index_offset + (uint64_t)p * 3];
            const orc_vertex *v0 = &desc->vertices[in->off.vertex_offset + ix[0]];
            const orc_vertex *v1 = &desc->vertices[in->off.vertex_offset + ix[1]];
            const orc_vertex *v2 = &desc->vertices[in->off.vertex_offset + ix[2]];
            wtri *t = &s->tris[k++];
            t->a = xform_point(in->xform, v3_make(v0->position[0], v0->position[1], v0->position[2]));
            t->b = xform_point(in->xform, v3_make(v1->position[0], v1->position[1], v1->position[2]));
            t->c = xform_point(in->xform, v3_make(v2->position[0], v2->position[1], v2->position[2]));
            t->instance = (uint32_t)i; t->prim = p;
        }
    }
    build_lights(s);
    if (nt > 64) {
        s->order = (uint32_t *)malloc(sizeof(uint32_t) * nt);
        for (uint64_t i = 0; i < nt; ++i) s->order[i] = (uint32_t)i;
        s->nodes = (bnode *)malloc(sizeof(bnode) * (2 * nt + 2));
        tbox *boxes = (tbox *)malloc(sizeof(tbox) * nt);
        for (uint64_t i = 0; i < nt; ++i) box_of(&s->tris[i], &boxes[i]);
        bctx c = { s, boxes };
        s->n_nodes = 1;
        build_rec(&c, 0, 0, (uint32_t)nt);
        free(boxes);
    }
    return s;
}

void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    free(s->tris); free(s->order); free(s->nodes); free(s->inst_first); free(s->cof); free(s->lights); free(s->mat_pdf_area); free(s);
}

uint64_t orc_scene_triangle_count(const orc_scene *s) { return s->n_tris; }

/* ------------------------------------------------------------------ */
/* shared.glsl:38-49 */
typedef struct {
    v3 attenuation;
    uint32_t depth;
    int done;
    v3 origin, direction;
    float in_uv[2];
    v3 normal;
    float dist;
} payload_t;

/* rmiss.glsl:18-34 */
static void miss_shader(payload_t *pl, v3 ray_dir)
{
    pl->attenuation = sky_color(ray_dir);
    pl->done = 1;
    pl->dist = 3.402823466e+38f;
    pl->normal = v3_make(0.0f, 0.0f, 0.0f);
}

/* one texel, decoded the way the hardware's R8G8B8A8_SRGB / _UNORM fetch does */
static inline v3 texel_at(const orc_texture *tx, uint32_t x, uint32_t y)
{
    const uint8_t *px = tx->rgba8 + ((uint64_t)y * tx->width + x) * 4;
    if (tx->srgb & 1u) return v3_make(g_srgb_lut[px[0]], g_srgb_lut[px[1]], g_srgb_lut[px[2]]);
    return v3_make((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

/* build extension (flag bit 2, glTF magFilter LINEAR, graphics.rs:446-451 reads it and drops it): bilinear
 * filter at LOD 0 in the Vulkan convention -- texel centres at i + 0.5, weights from the fraction of
 * u*W - 0.5, the four texels decoded to linear before the blend; neighbours wrap or clamp like the texel itself */
static inline void linear_axis(float u, uint32_t size, int repeat, uint32_t *i0, uint32_t *i1, float *a)
{
    float x = u * (float)size - 0.5f;
    float xf = floorf(x);
    float fr = x - xf;
    if (!(fr >= 0.0f)) fr = 0.0f;
    if (fr > 1.0f) fr = 1.0f;
    float hi = (float)size;
    if (!(xf >= -1.0f)) xf = -1.0f;
    if (xf > hi) xf = hi;
    int32_t k0 = (int32_t)xf, k1 = k0 + 1, n = (int32_t)size;
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
    *i0 = (uint32_t)k0; *i1 = (uint32_t)k1; *a = fr;
}

/* texture(textures[i], uv): vulkano SamplerCreateInfo::default() = nearest, clamp-to-edge,
 * LOD 0 (graphics.rs:453-459, parity unpinned); texel = clamp(floor(u*W),0,W-1) */
static v3 sample_texture(const orc_texture *tx, float u, float v)
{
    if (tx->srgb & 2u) { u = u - floorf(u); v = v - floorf(v); }   /* build extension: wrap = repeat */
    if (tx->srgb & 4u) {
        uint32_t x0, x1, y0, y1; float ax, ay;
        linear_axis(u, tx->width, (tx->srgb & 2u) != 0, &x0, &x1, &ax);
        linear_axis(v, tx->height, (tx->srgb & 2u) != 0, &y0, &y1, &ay);
        v3 c00 = texel_at(tx, x0, y0), c10 = texel_at(tx, x1, y0), c01 = texel_at(tx, x0, y1), c11 = texel_at(tx, x1, y1);
        float bx = 1.0f - ax, by = 1.0f - ay;
        v3 top = v3_add(v3_scale(c00, bx), v3_scale(c10, ax));
        v3 bot = v3_add(v3_scale(c01, bx), v3_scale(c11, ax));
        return v3_add(v3_scale(top, by), v3_scale(bot, ay));
    }
    float fu = floorf(u * (float)tx->width), fv = floorf(v * (float)tx->height);
    float mu = (float)(tx->width - 1), mv = (float)(tx->height - 1);
    if (!(fu >= 0.0f)) fu = 0.0f;
    if (fu > mu) fu = mu;
    if (!(fv >= 0.0f)) fv = 0.0f;
    if (fv > mv) fv = mv;
    return texel_at(tx, (uint32_t)fu, (uint32_t)fv);
}

/* rchit.glsl:158-254 */
static void closest_hit_shader(const orc_scene *s, payload_t *pl, const hit_rec *h, v3 ray_o, v3 ray_d,
                               uint32_t rng_acc, uint32_t seed, float normal_eps, orc_stats *st)
{
    const orc_scene_desc *d = &s->d;
    const orc_offsets *off = &d->instances[h->instance].off;            /* :160 */
    uint64_t ib = (uint64_t)off->index_offset + (uint64_t)h->prim * 3;  /* :163-166 */
    uint32_t i0 = d->indices[ib], i1 = d->indices[ib + 1], i2 = d->indices[ib + 2];
    const orc_vertex *v0 = &d->vertices[off->vertex_offset + i0];       /* :169-171 */
    const orc_vertex *v1 = &d->vertices[off->vertex_offset + i1];
    const orc_vertex *v2 = &d->vertices[off->vertex_offset + i2];

    float bx = 1.0f - h->u - h->v, by = h->u, bz = h->v;                /* :174 */
    v3 n0 = v3_make(v0->normal[0], v0->normal[1], v0->normal[2]);
    v3 n1 = v3_make(v1->normal[0], v1->normal[1], v1->normal[2]);
    v3 n2 = v3_make(v2->normal[0], v2->normal[1], v2->normal[2]);
    if (d->options & ORC_SCENE_WORLD_NORMALS) {     /* build extension; the reference keeps object space */
        n0 = normal_to_world(s->cof[h->instance], n0);
        n1 = normal_to_world(s->cof[h->instance], n1);
        n2 = normal_to_world(s->cof[h->instance], n2);
    }
    /* :175-179  object-space normal, never transformed, never flipped */
    v3 normal = v3_normalize(v3_add(v3_add(v3_scale(n0, bx), v3_scale(n1, by)), v3_scale(n2, bz)));
    float tu = (v0->uv[0] * bx + v1->uv[0] * by) + v2->uv[0] * bz;      /* :180-183 */
    float tv = (v0->uv[1] * bx + v1->uv[1] * by) + v2->uv[1] * bz;

    v3 world_pos = v3_add(ray_o, v3_scale(ray_d, h->t));                /* :186 */

    const orc_material *m = &d->materials[off->material_offset];        /* :190 */
    v3 albedo;
    if (m->has_base_texture) {                                          /* :193-198 */
        albedo = sample_texture(&d->textures[m->base_texture_indice], tu, tv);
        if (d->options & ORC_SCENE_TEXTURE_TIMES_FACTOR)
            albedo = v3_mul(albedo, v3_make(m->base_color[0], m->base_color[1], m->base_color[2]));
        if (st) st->textured_hits++;
    } else {
        albedo = v3_make(m->base_color[0], m->base_color[1], m->base_color[2]);
    }
    float roughness = m->roughness;                                     /* :200 */

    /* :209  reflect(I,N) = I - 2*dot(N,I)*N */
    float k2 = 2.0f * v3_dot(normal, ray_d);
    v3 reflected = v3_sub(ray_d, v3_scale(normal, k2));
    /* :212-218 */
    v3 rr = v3_normalize(cosine_hemisphere_sample(pl->in_uv[0], pl->in_uv[1], pl->depth, rng_acc, seed, normal));
    /* :220  mix(reflected, rr, roughness), left unnormalised */
    float om = 1.0f - roughness;
    v3 scatter = v3_add(v3_scale(reflected, om), v3_scale(rr, roughness));

    int should_scatter = 1;
    v3 att;
    if (v3_dot(scatter, normal) <= 0.0f) {                              /* :223-230 */
        should_scatter = 0;
        att = v3_make(0.0f, 0.0f, 0.0f);
    } else {
        att = albedo;
    }
    v3 scatter_origin = v3_add(world_pos, v3_scale(normal, normal_eps)); /* :242 */
    pl->origin = scatter_origin;                                        /* :245-249 */
    pl->direction = scatter;
    pl->attenuation = att;
    pl->done = should_scatter ? 0 : 1;
    pl->normal = normal;                                                /* :253 */
    pl->dist = v3_length(v3_sub(pl->origin, ray_o));                    /* :254 */
}

static inline v3 mat4_mul_dir4(const float m[16], float x, float y, float z, float w, float *w_out)
{
    /* GLSL mat4 * vec4 = c0*x + c1*y + c2*z + c3*w, summed left to right */
    v3 r;
    r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
    r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
    if (w_out) *w_out = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
    return r;
}

typedef struct {
    const orc_scene *s; const orc_camera *cam; orc_uniforms u; orc_params p;
    uint32_t W, H, acc;
    float *albedo, *normal, *depth; uint32_t *ids;
    uint32_t y0, y1;
    orc_stats st;
} job_t;

/* rgen.glsl:20-141 for one launch id */
static void raygen_pixel(job_t *j, uint32_t x, uint32_t y)
{
    const orc_params *p = &j->p;
    uint32_t W = j->W, H = j->H;
    uint32_t acc = j->acc;                       /* uniforms.accumulated_count of this frame */
    uint32_t rng_acc = acc + p->rng_sample_offset;
    float pcx = (float)x + 0.5f, pcy = (float)(H - y - 1) + 0.5f;      /* :25 */
    float uvx = pcx / (float)W, uvy = pcy / (float)H;                   /* :26 */
    float dx = uvx * 2.0f - 1.0f, dy = uvy * 2.0f - 1.0f;               /* :28 */

    v3 origin = mat4_mul_dir4(j->cam->view_inverse, 0.0f, 0.0f, 0.0f, 1.0f, NULL);   /* :30 */
    v3 target = mat4_mul_dir4(j->cam->proj_inverse, dx, dy, 1.0f, 1.0f, NULL);       /* :31 */
    v3 tn = v3_normalize(target);
    v3 direction = v3_normalize(mat4_mul_dir4(j->cam->view_inverse, tn.x, tn.y, tn.z, 0.0f, NULL)); /* :32-34 */

    payload_t pl;                                                       /* :41-46,60 */
    pl.depth = 0;
    pl.attenuation = v3_make(1.0f, 1.0f, 1.0f);
    pl.done = 1;
    pl.origin = origin;
    pl.direction = direction;
    pl.in_uv[0] = uvx; pl.in_uv[1] = uvy;
    pl.normal = v3_make(0, 0, 0); pl.dist = 0;

    v3 attenuation = v3_make(1.0f, 1.0f, 1.0f);                         /* :62 */
    uint64_t pix = (uint64_t)y * W + x;

    for (;;) {                                                          /* :64-108 */
        hit_rec h;
        v3 ro = pl.origin, rd = pl.direction;
        int hit = trace(j->s, ro, rd, p->t_min, p->t_max, p->flags, &h, &j->st);   /* :66-78 */
        j->st.rays++;
        if (hit) {
            j->st.hits++;
            closest_hit_shader(j->s, &pl, &h, ro, rd, rng_acc, j->u.seed, p->normal_eps, &j->st);
        } else {
            miss_shader(&pl, rd);
        }
        attenuation = v3_mul(attenuation, pl.attenuation);              /* :80 */

        if (pl.depth == 0) {                                            /* :83-99 */
            if (j->ids && acc == j->u.accumulated_count) {
                j->ids[pix * 2 + 0] = hit ? h.instance : 0xFFFFFFFFu;
                j->ids[pix * 2 + 1] = hit ? h.prim : 0xFFFFFFFFu;
            }
            if (acc == 0) {
                float z_range = j->u.z_far - j->u.z_near;
                float zn = (pl.dist - j->u.z_near) / z_range;
                zn = fminf(fmaxf(zn, 0.0f), 1.0f);
                if (j->depth) j->depth[pix] = zn;
                if (j->normal) {
                    j->normal[pix * 4 + 0] = pl.normal.x; j->normal[pix * 4 + 1] = pl.normal.y;
                    j->normal[pix * 4 + 2] = pl.normal.z; j->normal[pix * 4 + 3] = 1.0f;
                }
            }
        }
        pl.depth++;                                                     /* :101-103 */
        if (pl.done == 1 || pl.depth >= p->max_depth) break;
        pl.done = 1;                                                    /* :107 */
    }

    /* :124-141  running mean */
    float n = (float)acc;
    float *px = &j->albedo[pix * 4];
    v3 cur = v3_make(px[0], px[1], px[2]);
    v3 scaled_cur = v3_make(cur.x * n / (n + 1.0f), cur.y * n / (n + 1.0f), cur.z * n / (n + 1.0f));
    v3 scaled_att = v3_make(attenuation.x / (n + 1.0f), attenuation.y / (n + 1.0f), attenuation.z / (n + 1.0f));
    px[0] = scaled_att.x + scaled_cur.x;
    px[1] = scaled_att.y + scaled_cur.y;
    px[2] = scaled_att.z + scaled_cur.z;
    px[3] = 1.0f;
    j->st.samples++;
}

/* ------------------------------------------------------------------ */
/* Lit integrator (build extension; todo.md:4 "Proper lighting", the Russian-roulette
 * sketch of rchit.glsl:256-265).  NOT a restatement of reference code: the reference has no light
 * sources beyond its sky.  This is the specification the HIP stages k_shade_lit / k_shadow_resolve
 * implement, expression for expression:
 *   - radiance L accumulated per path: sky on a miss, emission in full when a camera ray or a
 *     specular bounce hits an emitter;
 *   - next-event estimation at every opaque hit: one emissive triangle picked by area x luminance,
 *     one uniformly sampled point, one occlusion query;
 *   - multiple importance sampling (power heuristic) between that light sample and the BSDF sample of
 *     the same vertex: the light sample is weighted by p_l^2 / (p_l^2 + p_b^2), and an emitter reached
 *     by the sampled diffuse / glossy bounce by p_b^2 / (p_b^2 + p_l^2), both densities per solid angle;
 *   - BSDF: glTF metallic-roughness = Lambert + GGX (VNDF sampling, Smith separable G, Schlick F),
 *     one-sample lobe choice weighted with the mixture pdf; KHR_materials_transmission = smooth
 *     dielectric (Schlick Fresnel, Snell refraction);
 *   - Russian roulette from the fourth bounce on.
 * Random numbers: rand() of shared.glsl:194-207 with per-dimension seed offsets. */
#define LIT_SEED_LOBE 101u
#define LIT_SEED_MIX 151u
#define LIT_SEED_LIGHT 211u
#define LIT_SEED_LU 307u
#define LIT_SEED_LV 401u
#define LIT_SEED_RR 503u
#define LIT_SEED_U1 7919u
#define LIT_SEED_U2 16381u
#define LIT_INV_PI 0.318309886f
#define LIT_SHADOW_TMAX 0.999f

static inline float ggx_d(float noh, float a2)
{
    float d = (noh * noh) * (a2 - 1.0f) + 1.0f;
    return a2 / ((3.14159265f * d) * d);
}
static inline float smith_g1(float nox, float a2)
{
    return (2.0f * nox) / (nox + sqrtf(a2 + (1.0f - a2) * (nox * nox)));
}
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

/* f (rgb) and the mixture pdf of the opaque BSDF for unit wo, wi on the side of N */
static int bsdf_eval(v3 n, v3 wo, v3 wi, v3 diff, v3 f0, float a2, float p_spec, v3 *f, float *pdf)
{
    float nol = v3_dot(n, wi), nov = v3_dot(n, wo);
    if (!(nol > 0.0f && nov > 0.0f)) return 0;
    v3 h = v3_normalize(v3_add(wo, wi));
    float noh = v3_dot(n, h), voh = v3_dot(wo, h);
    float dd = ggx_d(noh, a2), g1v = smith_g1(nov, a2), g1l = smith_g1(nol, a2);
    float x = 1.0f - voh;
    if (x < 0.0f) x = 0.0f;
    float fr = pow5(x);
    float spec = (dd * (g1v * g1l)) / ((4.0f * nov) * nol);
    f->x = diff.x * LIT_INV_PI + (f0.x + (1.0f - f0.x) * fr) * spec;
    f->y = diff.y * LIT_INV_PI + (f0.y + (1.0f - f0.y) * fr) * spec;
    f->z = diff.z * LIT_INV_PI + (f0.z + (1.0f - f0.z) * fr) * spec;
    *pdf = p_spec * ((dd * g1v) / (4.0f * nov)) + (1.0f - p_spec) * (nol * LIT_INV_PI);
    return 1;
}

/* Heitz 2018, "Sampling the GGX Distribution of Visible Normals": half vector in world space */
static v3 ggx_sample_vndf(v3 n, v3 wo, float alpha, float u1, float u2)
{
    v3 t, b;
    create_basis(n, &t, &b);
    v3 vl = v3_make(v3_dot(wo, t), v3_dot(wo, b), v3_dot(wo, n));
    v3 vh = v3_normalize(v3_make(alpha * vl.x, alpha * vl.y, vl.z));
    float lensq = vh.x * vh.x + vh.y * vh.y;
    v3 t1 = v3_make(1.0f, 0.0f, 0.0f);
    if (lensq > 0.0f) { float il = 1.0f / sqrtf(lensq); t1 = v3_make(-vh.y * il, vh.x * il, 0.0f); }
    v3 t2 = v3_cross(vh, t1);
    float r = sqrtf(u1), phi = 6.283185307f * u2, sn, cs;
    orc_sincos(phi, &sn, &cs);
    float p1 = r * cs, p2 = r * sn;
    float sw = 0.5f * (1.0f + vh.z);
    float om = 1.0f - p1 * p1;
    p2 = (1.0f - sw) * sqrtf(0.0f < om ? om : 0.0f) + sw * p2;
    float oz = (1.0f - p1 * p1) - p2 * p2;
    float pz = sqrtf(0.0f < oz ? oz : 0.0f);
    v3 nh = v3_add(v3_add(v3_scale(t1, p1), v3_scale(t2, p2)), v3_scale(vh, pz));
    v3 ne = v3_normalize(v3_make(alpha * nh.x, alpha * nh.y, 0.0f < nh.z ? nh.z : 0.0f));
    return v3_add(v3_add(v3_scale(t, ne.x), v3_scale(b, ne.y)), v3_scale(n, ne.z));
}

/* power heuristic (beta = 2) of the strategy with density a against the one with density b */
static inline float lit_mis_weight(float a, float b)
{
    float a2 = a * a, b2 = b * b;
    float w = a2 / (a2 + b2);
    if (!(w >= 0.0f)) w = 0.0f;
    return w;
}

static inline float lit_rand(const float uv[2], uint32_t depth, uint32_t acc, uint32_t seed, uint32_t dim)
{
    return orc_rand(uv[0], uv[1], depth, acc, seed + dim);
}

static void raygen_pixel_lit(job_t *j, uint32_t x, uint32_t y)
{
    const orc_params *p = &j->p;
    const orc_scene *s = j->s;
    const orc_scene_desc *d = &s->d;
    uint32_t W = j->W, H = j->H, acc = j->acc, rng_acc = acc + p->rng_sample_offset, seed = j->u.seed;
    float pcx = (float)x + 0.5f, pcy = (float)(H - y - 1) + 0.5f;
    float uv[2] = { pcx / (float)W, pcy / (float)H };
    float dx = uv[0] * 2.0f - 1.0f, dy = uv[1] * 2.0f - 1.0f;
    v3 o = mat4_mul_dir4(j->cam->view_inverse, 0.0f, 0.0f, 0.0f, 1.0f, NULL);
    v3 target = mat4_mul_dir4(j->cam->proj_inverse, dx, dy, 1.0f, 1.0f, NULL);
    v3 tn = v3_normalize(target);
    v3 dir = v3_normalize(mat4_mul_dir4(j->cam->view_inverse, tn.x, tn.y, tn.z, 0.0f, NULL));
    uint64_t pix = (uint64_t)y * W + x;

    v3 L = v3_make(0.0f, 0.0f, 0.0f), thr = v3_make(1.0f, 1.0f, 1.0f);
    float prev_pdf = 0.0f;      /* solid-angle pdf the BSDF sampling gave this ray's direction; <= 0: camera ray or delta lobe */
    for (uint32_t depth = 0;; ) {
        hit_rec h;
        int hit = trace(s, o, dir, p->t_min, p->t_max, p->flags, &h, &j->st);
        j->st.rays++;
        v3 gb_normal = v3_make(0.0f, 0.0f, 0.0f);
        float gb_dist = 3.402823466e+38f;
        int done = 0;
        v3 new_o = o, new_d = dir;
        if (!hit) {
            L = v3_add(L, v3_mul(thr, sky_color(dir)));
            done = 1;
        } else {
            j->st.hits++;
            const orc_offsets *off = &d->instances[h.instance].off;
            uint64_t ib = (uint64_t)off->index_offset + (uint64_t)h.prim * 3;
            const orc_vertex *v0 = &d->vertices[off->vertex_offset + d->indices[ib]];
            const orc_vertex *v1 = &d->vertices[off->vertex_offset + d->indices[ib + 1]];
            const orc_vertex *v2 = &d->vertices[off->vertex_offset + d->indices[ib + 2]];
            const wtri *wt = &s->tris[s->inst_first[h.instance] + h.prim];
            float bx = 1.0f - h.u - h.v, by = h.u, bz = h.v;
            v3 n0 = v3_make(v0->normal[0], v0->normal[1], v0->normal[2]);
            v3 n1 = v3_make(v1->normal[0], v1->normal[1], v1->normal[2]);
            v3 n2 = v3_make(v2->normal[0], v2->normal[1], v2->normal[2]);
            if (d->options & ORC_SCENE_WORLD_NORMALS) {
                n0 = normal_to_world(s->cof[h.instance], n0);
                n1 = normal_to_world(s->cof[h.instance], n1);
                n2 = normal_to_world(s->cof[h.instance], n2);
            }
            v3 ns = v3_normalize(v3_add(v3_add(v3_scale(n0, bx), v3_scale(n1, by)), v3_scale(n2, bz)));
            v3 ng = v3_normalize(v3_cross(v3_sub(wt->b, wt->a), v3_sub(wt->c, wt->a)));
            int front = 1;
            if (v3_dot(ng, dir) > 0.0f) { ng = v3_scale(ng, -1.0f); front = 0; }
            if (v3_dot(ns, ng) < 0.0f) ns = v3_scale(ns, -1.0f);
            float tu = (v0->uv[0] * bx + v1->uv[0] * by) + v2->uv[0] * bz;
            float tv = (v0->uv[1] * bx + v1->uv[1] * by) + v2->uv[1] * bz;
            v3 P = v3_add(o, v3_scale(dir, h.t));
            const orc_material *m = &d->materials[off->material_offset];
            const orc_material_ext *e = material_ext(s, off->material_offset);
            v3 albedo = v3_make(m->base_color[0], m->base_color[1], m->base_color[2]);
            if (m->has_base_texture) {
                albedo = v3_mul(sample_texture(&d->textures[m->base_texture_indice], tu, tv), albedo);
                j->st.textured_hits++;
            }
            if (e->normal_texture) {
                /* glTF normalTexture: tangent-space normal, +Y up.  The vertex format (shared.glsl:1-9) has no tangent, so
                 * the frame comes from the triangle: T along dP/du, B along -dP/dv (glTF's v runs downwards), both made
                 * orthogonal to the shading normal.  A degenerate uv mapping, or a perturbed normal that ends up below
                 * the geometric surface, leaves the shading normal as it is. */
                v3 tn = sample_texture(&d->textures[e->normal_texture - 1], tu, tv);
                /* glTF: scaledNormal = normalize((texel * 2 - 1) * vec3(scale, scale, 1)) */
                const float nsc = s->d.normal_scale ? s->d.normal_scale[off->material_offset] : 1.0f;
                float mx = (tn.x * 2.0f - 1.0f) * nsc, my = (tn.y * 2.0f - 1.0f) * nsc, mz = tn.z * 2.0f - 1.0f;
                v3 e1 = v3_sub(wt->b, wt->a), e2 = v3_sub(wt->c, wt->a);
                float du1 = v1->uv[0] - v0->uv[0], dv1 = v1->uv[1] - v0->uv[1];
                float du2 = v2->uv[0] - v0->uv[0], dv2 = v2->uv[1] - v0->uv[1];
                float det = du1 * dv2 - du2 * dv1;
                float sg = det < 0.0f ? -1.0f : 1.0f;
                v3 tt = v3_scale(v3_sub(v3_scale(e1, dv2), v3_scale(e2, dv1)), sg);        /* dP/du x |det| */
                v3 bt = v3_scale(v3_sub(v3_scale(e1, du2), v3_scale(e2, du1)), sg);        /* -dP/dv x |det| */
                tt = v3_sub(tt, v3_scale(ns, v3_dot(ns, tt)));
                float tl2 = v3_dot(tt, tt);
                if (det != 0.0f && tl2 > 0.0f && isfinite(tl2)) {
                    tt = v3_scale(tt, 1.0f / sqrtf(tl2));
                    v3 bb = v3_cross(ns, tt);
                    if (v3_dot(bb, bt) < 0.0f) bb = v3_scale(bb, -1.0f);
                    v3 np = v3_add(v3_add(v3_scale(tt, mx), v3_scale(bb, my)), v3_scale(ns, mz));
                    float nl2 = v3_dot(np, np);
                    if (nl2 > 0.0f && isfinite(nl2)) {
                        np = v3_scale(np, 1.0f / sqrtf(nl2));
                        if (v3_dot(np, ng) > 0.0f) ns = np;
                    }
                }
            }
            v3 wo = v3_scale(dir, -1.0f);
            v3 up_o = v3_add(P, v3_scale(ng, p->normal_eps));
            gb_normal = ns;
            gb_dist = v3_length(v3_sub(up_o, o));

            if (e->emissive[0] > 0.0f || e->emissive[1] > 0.0f || e->emissive[2] > 0.0f) {
                v3 em = v3_make(e->emissive[0], e->emissive[1], e->emissive[2]);
                if (e->emissive_texture) em = v3_mul(em, sample_texture(&d->textures[e->emissive_texture - 1], tu, tv));
                v3 le = v3_mul(thr, em);
                if (!(prev_pdf > 0.0f)) {
                    L = v3_add(L, le);
                } else if (s->n_lights) {
                    /* the light sampling of the previous vertex could have produced this point too */
                    float cos_l = fabsf(v3_dot(ng, dir));
                    float p_l = (s->mat_pdf_area[off->material_offset] * (h.t * h.t)) / cos_l;
                    float w = (p->flags & 2u) ? 0.0f : lit_mis_weight(prev_pdf, p_l);
                    L = v3_add(L, v3_scale(le, w));
                }
            }

            float u_lobe = lit_rand(uv, depth, rng_acc, seed, LIT_SEED_LOBE);
            float u1 = lit_rand(uv, depth, rng_acc, seed, LIT_SEED_U1);
            float u2 = lit_rand(uv, depth, rng_acc, seed, LIT_SEED_U2);
            if (e->transmission > 0.0f && u_lobe < e->transmission) {
                /* smooth dielectric */
                float ior = e->ior;
                float eta = front ? 1.0f / ior : ior;
                float cosi = v3_dot(wo, ns);
                if (cosi < 0.0f) cosi = 0.0f;
                if (cosi > 1.0f) cosi = 1.0f;
                float sin2t = (eta * eta) * (1.0f - cosi * cosi);
                int reflect = 1;
                float cost = 0.0f;
                if (sin2t < 1.0f) {
                    cost = sqrtf(1.0f - sin2t);
                    float r0 = (1.0f - ior) / (1.0f + ior);
                    r0 = r0 * r0;
                    float c = front ? cosi : cost;
                    float fr = r0 + (1.0f - r0) * pow5(1.0f - c);
                    reflect = u1 < fr;
                }
                if (reflect) {
                    new_d = v3_normalize(v3_add(dir, v3_scale(ns, 2.0f * cosi)));
                    new_o = up_o;
                } else {
                    new_d = v3_normalize(v3_add(v3_scale(dir, eta), v3_scale(ns, eta * cosi - cost)));
                    new_o = v3_sub(P, v3_scale(ng, p->normal_eps));
                    thr = v3_mul(thr, albedo);
                }
                prev_pdf = 0.0f;
            } else {
                float rough = m->roughness, metal = m->metallic;
                if (e->mr_texture) {        /* glTF: G = roughness, B = metallic, linear texels, multiplied into the factors */
                    v3 mr = sample_texture(&d->textures[e->mr_texture - 1], tu, tv);
                    rough = rough * mr.y;
                    metal = metal * mr.z;
                }
                float alpha = rough * rough;
                if (alpha < 0.0025f) alpha = 0.0025f;
                float a2 = alpha * alpha;
                float om = 1.0f - metal;
                v3 f0 = v3_make(0.04f * om + albedo.x * metal, 0.04f * om + albedo.y * metal, 0.04f * om + albedo.z * metal);
                v3 diff = v3_scale(albedo, om);
                float p_spec = 0.25f + 0.75f * metal;

                /* next-event estimation */
                if (s->n_lights) {
                    float ul = lit_rand(uv, depth, rng_acc, seed, LIT_SEED_LIGHT);
                    uint64_t lo = 0, hi = s->n_lights;
                    while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (s->lights[mid].cdf <= ul) lo = mid + 1; else hi = mid; }
                    if (lo >= s->n_lights) lo = s->n_lights - 1;
                    const light_t *l = &s->lights[lo];
                    float a = lit_rand(uv, depth, rng_acc, seed, LIT_SEED_LU);
                    float b = lit_rand(uv, depth, rng_acc, seed, LIT_SEED_LV);
                    float su = sqrtf(a);
                    float w0 = 1.0f - su, w1 = su * (1.0f - b), w2 = su * b;
                    v3 pl = v3_add(v3_add(v3_scale(l->a, w0), v3_scale(l->b, w1)), v3_scale(l->c, w2));
                    v3 dl = v3_sub(pl, up_o);
                    float dist2 = v3_dot(dl, dl);
                    float inv_dist = 1.0f / sqrtf(dist2);
                    v3 wl = v3_scale(dl, inv_dist);
                    float cos_g = v3_dot(ng, wl);
                    float cos_l = fabsf(v3_dot(l->n, wl));
                    v3 f; float p_b;
                    if (dist2 > 0.0f && cos_g > 0.0f && cos_l > 0.0f && bsdf_eval(ns, wo, wl, diff, f0, a2, p_spec, &f, &p_b)) {
                        float cos_s = v3_dot(ns, wl);
                        float gterm = (cos_s * cos_l) / (dist2 * l->pdf_area);
                        float p_l = (l->pdf_area * dist2) / cos_l;
                        v3 le_l = l->le;
                        if (l->etex) {
                            float lu = (l->uv[0][0] * w0 + l->uv[1][0] * w1) + l->uv[2][0] * w2;
                            float lv = (l->uv[0][1] * w0 + l->uv[1][1] * w1) + l->uv[2][1] * w2;
                            le_l = v3_mul(le_l, sample_texture(&d->textures[l->etex - 1], lu, lv));
                        }
                        v3 c = v3_scale(v3_scale(v3_mul(v3_mul(thr, f), le_l), gterm), (p->flags & 2u) ? 1.0f : lit_mis_weight(p_l, p_b));
                        hit_rec sh;
                        j->st.shadow_rays++;
                        if (!trace(s, up_o, dl, p->t_min, LIT_SHADOW_TMAX, p->flags, &sh, &j->st))
                            L = v3_add(L, c);
                    }
                }

                /* sample the next direction */
                float u_mix = lit_rand(uv, depth, rng_acc, seed, LIT_SEED_MIX);
                v3 wi;
                if (u_mix < p_spec) {
                    v3 hv = ggx_sample_vndf(ns, wo, alpha, u1, u2);
                    wi = v3_normalize(v3_sub(v3_scale(hv, 2.0f * v3_dot(wo, hv)), wo));
                } else {
                    float r = sqrtf(u1), phi = 6.283185307f * u2, sn, cs;
                    orc_sincos(phi, &sn, &cs);
                    float lx = r * cs, ly = r * sn, oz = 1.0f - u1;
                    float lz = sqrtf(0.0f < oz ? oz : 0.0f);
                    v3 t, b;
                    create_basis(ns, &t, &b);
                    wi = v3_normalize(v3_add(v3_add(v3_scale(t, lx), v3_scale(b, ly)), v3_scale(ns, lz)));
                }
                v3 f; float pdf;
                if (v3_dot(wi, ng) > 0.0f && bsdf_eval(ns, wo, wi, diff, f0, a2, p_spec, &f, &pdf) && pdf > 0.0f) {
                    float wgt = v3_dot(ns, wi) / pdf;
                    thr = v3_mul(thr, v3_scale(f, wgt));
                    new_d = wi;
                    new_o = up_o;
                    prev_pdf = pdf;
                } else {
                    done = 1;
                }
            }
            /* Russian roulette (rchit.glsl:256-265 sketches one; this one is unbiased) */
            if (!done && depth >= 3) {
                float pr = fmaxf(thr.x, fmaxf(thr.y, thr.z));
                if (pr > 1.0f) pr = 1.0f;
                if (pr < 0.05f) pr = 0.05f;
                if (lit_rand(uv, depth, rng_acc, seed, LIT_SEED_RR) >= pr) done = 1;
                else thr = v3_scale(thr, 1.0f / pr);
            }
            if (!done && !(finite3(thr) && finite3(new_d))) done = 1;
        }

        if (depth == 0) {
            if (j->ids && acc == j->u.accumulated_count) {
                j->ids[pix * 2 + 0] = hit ? h.instance : 0xFFFFFFFFu;
                j->ids[pix * 2 + 1] = hit ? h.prim : 0xFFFFFFFFu;
            }
            if (acc == 0) {
                float z_range = j->u.z_far - j->u.z_near;
                float zn = (gb_dist - j->u.z_near) / z_range;
                zn = fminf(fmaxf(zn, 0.0f), 1.0f);
                if (j->depth) j->depth[pix] = zn;
                if (j->normal) {
                    j->normal[pix * 4 + 0] = gb_normal.x; j->normal[pix * 4 + 1] = gb_normal.y;
                    j->normal[pix * 4 + 2] = gb_normal.z; j->normal[pix * 4 + 3] = 1.0f;
                }
            }
        }
        depth++;
        if (done || depth >= p->max_depth) break;
        o = new_o; dir = new_d;
    }

    /* rgen.glsl:124-141 running mean, with the radiance in place of the attenuation product */
    float n = (float)acc;
    float *px = &j->albedo[pix * 4];
    v3 cur = v3_make(px[0], px[1], px[2]);
    v3 scaled_cur = v3_make(cur.x * n / (n + 1.0f), cur.y * n / (n + 1.0f), cur.z * n / (n + 1.0f));
    v3 scaled_att = v3_make(L.x / (n + 1.0f), L.y / (n + 1.0f), L.z / (n + 1.0f));
    px[0] = scaled_att.x + scaled_cur.x;
    px[1] = scaled_att.y + scaled_cur.y;
    px[2] = scaled_att.z + scaled_cur.z;
    px[3] = 1.0f;
    j->st.samples++;
}

typedef struct { job_t *jobs; uint32_t n_bands, t, nt; } band_arg;

/* static round-robin assignment of row bands to threads */
static void *thread_bands(void *arg)
{
    band_arg *a = (band_arg *)arg;
    for (uint32_t b = a->t; b < a->n_bands; b += a->nt) {
        job_t *j = &a->jobs[b];
        for (uint32_t y = j->y0; y < j->y1; ++y)
            for (uint32_t x = 0; x < j->W; ++x) {
                if (j->p.integrator == ORC_INTEGRATOR_LIT) raygen_pixel_lit(j, x, y);
                else raygen_pixel(j, x, y);
            }
    }
    return NULL;
}

void orc_render(const orc_scene *s, const orc_camera *cam, const orc_uniforms *u,
                const orc_params *p, uint32_t width, uint32_t height, uint32_t n_samples,
                float *albedo, float *normal, float *depth, uint32_t *primary_ids,
                orc_stats *stats, int n_threads)
{
    orc_render_rows(s, cam, u, p, width, height, 0, height, n_samples, albedo, normal, depth, primary_ids, stats, n_threads);
}

/* Rows [row_begin, row_end) of the same frames: a pixel's value depends on nothing but its own launch id
 * (rgen.glsl:20-141 has no neighbourhood), so a band of a large frame equals the same rows of the full render.
 * The image pointers are those of the FULL W x H images; rows outside the band are left untouched. */
void orc_render_rows(const orc_scene *s, const orc_camera *cam, const orc_uniforms *u,
                     const orc_params *p, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end,
                     uint32_t n_samples, float *albedo, float *normal, float *depth, uint32_t *primary_ids,
                     orc_stats *stats, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (row_end > height) row_end = height;
    if (row_begin > row_end) row_begin = row_end;
    /* interleaved row bands keep threads balanced (sky rows are cheap) */
    uint32_t band = 4, n_bands = (row_end - row_begin + band - 1) / band;
    orc_stats total; memset(&total, 0, sizeof total);
    for (uint32_t f = 0; f < n_samples; ++f) {
        pthread_t th[256]; job_t *jobs = (job_t *)calloc(n_bands, sizeof(job_t));
        for (uint32_t b = 0; b < n_bands; ++b) {
            job_t *j = &jobs[b];
            j->s = s; j->cam = cam; j->u = *u; j->p = *p; j->W = width; j->H = height;
            j->acc = u->accumulated_count + f;
            j->albedo = albedo; j->normal = normal; j->depth = depth; j->ids = primary_ids;
            j->y0 = row_begin + b * band; j->y1 = j->y0 + band < row_end ? j->y0 + band : row_end;
        }
        band_arg *ta = (band_arg *)calloc((size_t)n_threads, sizeof(band_arg));
        for (int t = 0; t < n_threads; ++t) {
            ta[t].jobs = jobs; ta[t].n_bands = n_bands; ta[t].t = (uint32_t)t; ta[t].nt = (uint32_t)n_threads;
            if (n_threads > 1) pthread_create(&th[t], NULL, thread_bands, &ta[t]);
            else thread_bands(&ta[t]);
        }
        if (n_threads > 1) for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
        for (uint32_t b = 0; b < n_bands; ++b) {
            total.rays += jobs[b].st.rays; total.hits += jobs[b].st.hits; total.samples += jobs[b].st.samples;
            total.textured_hits += jobs[b].st.textured_hits;
            total.nodes_visited += jobs[b].st.nodes_visited; total.tris_tested += jobs[b].st.tris_tested;
            total.shadow_rays += jobs[b].st.shadow_rays;
        }
        free(ta); free(jobs);
    }
    if (stats) *stats = total;
}

/* ------------------------------------------------------------------ */
/* Host-side restatements (glam 0.30.2 is not vendored: parity unpinned) */

/* scene/mod.rs:557-566  Mat4::from_scale_rotation_translation(scale, rotation.normalize(), position)
 * then renderer.rs:1416-1423 keeps rows 0..2 of the matrix (row-major 3x4). */
void orc_transform_matrix(const float pos[3], const float q_in[4], const float scale[3], float m[3][4])
{
    float inv = 1.0f / sqrtf(((q_in[0] * q_in[0] + q_in[1] * q_in[1]) + q_in[2] * q_in[2]) + q_in[3] * q_in[3]);
    float x = q_in[0] * inv, y = q_in[1] * inv, z = q_in[2] * inv, w = q_in[3] * inv;
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    float ax[3] = { 1.0f - (yy + zz), xy + wz, xz - wy };   /* x axis (column 0) */
    float ay[3] = { xy - wz, 1.0f - (xx + zz), yz + wx };
    float az[3] = { xz + wy, yz - wx, 1.0f - (xx + yy) };
    for (int r = 0; r < 3; ++r) {
        m[r][0] = ax[r] * scale[0];
        m[r][1] = ay[r] * scale[1];
        m[r][2] = az[r] * scale[2];
        m[r][3] = pos[r];
    }
}

static v3 quat_mul_vec3(const float q[4], v3 v)
{
    /* glam Quat::mul_vec3 */
    v3 b = v3_make(q[0], q[1], q[2]);
    float w = q[3];
    float b2 = v3_dot(b, b);
    v3 r = v3_scale(v, w * w - b2);
    r = v3_add(r, v3_scale(b, v3_dot(v, b) * 2.0f));
    r = v3_add(r, v3_scale(v3_cross(b, v), w * 2.0f));
    return r;
}

/* GLM-derived cofactor inverse as used by glam Mat4::inverse; m, out column-major */
static void mat4_inverse(const float m[16], float out[16])
{
    float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    float m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    float c00 = m22 * m33 - m32 * m23, c02 = m12 * m33 - m32 * m13, c03 = m12 * m23 - m22 * m13;
    float c04 = m21 * m33 - m31 * m23, c06 = m11 * m33 - m31 * m13, c07 = m11 * m23 - m21 * m13;
    float c08 = m21 * m32 - m31 * m22, c10 = m11 * m32 - m31 * m12, c11 = m11 * m22 - m21 * m12;
    float c12 = m20 * m33 - m30 * m23, c14 = m10 * m33 - m30 * m13, c15 = m10 * m23 - m20 * m13;
    float c16 = m20 * m32 - m30 * m22, c18 = m10 * m32 - m30 * m12, c19 = m10 * m22 - m20 * m12;
    float c20 = m20 * m31 - m30 * m21, c22 = m10 * m31 - m30 * m11, c23 = m10 * m21 - m20 * m11;
    float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
    float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
    float v0[4] = { m10, m00, m00, m00 }, v1[4] = { m11, m01, m01, m01 };
    float v2[4] = { m12, m02, m02, m02 }, v3_[4] = { m13, m03, m03, m03 };
    static const float sa[4] = { 1.0f, -1.0f, 1.0f, -1.0f }, sb[4] = { -1.0f, 1.0f, -1.0f, 1.0f };
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        inv[0 + i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3_[i] * f2[i]) * sa[i];
        inv[4 + i] = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3_[i] * f4[i]) * sb[i];
        inv[8 + i] = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3_[i] * f5[i]) * sa[i];
        inv[12 + i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
    }
    float dot1 = ((m00 * inv[0] + m01 * inv[4]) + m02 * inv[8]) + m03 * inv[12];
    float rcp = 1.0f / dot1;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * rcp;
}

static void mat4_mul(const float a[16], const float b[16], float out[16])
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            out[c * 4 + r] = ((a[r] * b[c * 4] + a[4 + r] * b[c * 4 + 1]) + a[8 + r] * b[c * 4 + 2]) + a[12 + r] * b[c * 4 + 3];
}

/* renderer.rs:970-980,1011-1015: perspective_rh / look_to_rh / inverse */
void orc_camera_matrices(const float pos[3], const float q[4], float fov_y, float aspect,
                         float z_near, float z_far, orc_camera *out)
{
    float sin_fov = sinf(0.5f * fov_y), cos_fov = cosf(0.5f * fov_y);
    float h = cos_fov / sin_fov, w = h / aspect, r = z_far / (z_near - z_far);
    float proj[16] = { w, 0, 0, 0,  0, h, 0, 0,  0, 0, r, -1.0f,  0, 0, r * z_near, 0 };
    v3 eye = v3_make(pos[0], pos[1], pos[2]);
    v3 fwd = quat_mul_vec3(q, v3_make(0.0f, 0.0f, -1.0f));   /* Transform::forward, scene/mod.rs:569-571 */
    v3 up = quat_mul_vec3(q, v3_make(0.0f, 1.0f, 0.0f));     /* Transform::up, :577-579 */
    v3 f = v3_normalize(fwd);
    v3 s = v3_normalize(v3_cross(f, up));
    v3 uu = v3_cross(s, f);
    float view[16] = { s.x, uu.x, -f.x, 0,  s.y, uu.y, -f.y, 0,  s.z, uu.z, -f.z, 0,
                       -v3_dot(eye, s), -v3_dot(eye, uu), v3_dot(eye, f), 1.0f };
    mat4_mul(proj, view, out->view_proj);
    mat4_inverse(view, out->view_inverse);
    mat4_inverse(proj, out->proj_inverse);
}
