// trav_lab.cpp -- developer experiment (not part of the product or the tests): what would a different traversal policy or a
// tighter node encoding save on the product's BVH8?  A plain scalar closest-hit walk over the built tree with switches:
//   boxes:  0 = the node's 8-bit planes (what the kernels test), 1 = exact child boxes (upper bound for any finer encoding),
//           2 = planes requantised to 11 bits per node (fp16-exact integers 0..2047)
//   order:  0 = octant order (slot ^ octant, the kernels' order), 1 = children sorted by entry distance
//   cull:   1 = a stack entry carries its entry distance and is dropped at pop time when the closest hit has moved in front
//           2 = the entry carries the entry distance of its PARENT node's quantisation frame (one value per node group, what
//               fits the product's (group, mask) stack), truncated to the upper 16 bits of the float
//           4 = one distance per pushed GROUP: the smallest entry distance of the node's hit inner children
//           5 = one distance per pushed group: the smallest entry distance of the children that stay on the stack (the nearest
//               in pop order is taken at once)
// Counts node visits and triangle tests per ray.  Same triangle test as the product (pt_math.h).
#include "../../path_tracer_amd/csrc/bvh8.h"
#include "../../path_tracer_amd/csrc/bvh_quant.h"
#include "../../path_tracer_amd/csrc/traverse.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

using namespace ptx;

namespace {
struct Lab {
    const Bvh8 *bvh;
    std::vector<TriRec> tris;
    std::vector<Box> child;      // 8 per node: exact child boxes (leaf: padded triangle boxes)
    std::vector<Box> node_box;
};
}

static unsigned long long g_cull_first = 0, g_cull_later = 0;
static int g_delay = 0;     // triangle tests of a leaf hit happen `g_delay` node visits later (the kernels' wave-level triangle queue tests them in blocks)
extern "C" {
void tlab_set_delay(int d) { g_delay = d; }
void tlab_cull_counts(unsigned long long *o) { o[0] = g_cull_first; o[1] = g_cull_later; g_cull_first = g_cull_later = 0; }
void *tlab_create(void *bvh_handle, const float *verts)
{
    Lab *L = new Lab;
    L->bvh = (const Bvh8 *)bvh_handle;
    const Bvh8 &bvh = *L->bvh;
    L->tris.resize(bvh.tri_order.size());
    std::vector<Box> leaf(bvh.tri_order.size());
    Box scene; scene.reset();
    for (size_t i = 0; i < L->tris.size(); ++i) {
        uint32_t t = bvh.tri_order[i];
        for (int a = 0; a < 3; ++a) { L->tris[i].v0[a] = verts[t * 9 + a]; L->tris[i].v1[a] = verts[t * 9 + 3 + a]; L->tris[i].v2[a] = verts[t * 9 + 6 + a]; }
        L->tris[i].instance = 0; L->tris[i].prim = t; L->tris[i].rank = t;
        tri_raw_box(&verts[t * 9], leaf[i]);
        scene.grow(leaf[i]);
    }
    float ext = std::max(scene.hi[0] - scene.lo[0], std::max(scene.hi[1] - scene.lo[1], scene.hi[2] - scene.lo[2]));
    for (auto &b : leaf) pad_tri_box(b, ext);
    size_t n = bvh.nodes.size();
    L->child.resize(n * 8); L->node_box.resize(n);
    for (size_t i = n; i-- > 0;) {       // children have larger indices than their parent (breadth-first storage)
        const Bvh8Node &nd = bvh.nodes[i];
        Box nb; nb.reset();
        for (int s = 0; s < 8; ++s) {
            Box cb; cb.reset();
            if (nd.meta[s]) {
                if ((nd.imask >> s) & 1u) cb = L->node_box[nd.child_base + bq_popc(nd.imask & ((1u << s) - 1u))];
                else { uint32_t off = nd.meta[s] & 31u, cnt = bq_popc((uint32_t)nd.meta[s] >> 5); for (uint32_t k = 0; k < cnt; ++k) cb.grow(leaf[nd.tri_base + off + k]); }
                nb.grow(cb);
            }
            L->child[i * 8 + s] = cb;
        }
        L->node_box[i] = nb;
    }
    return L;
}

// out[0..3]: rays, hits, node visits, triangle tests
void tlab_trace(void *h, const float *rays, uint64_t n, float t_min, float t_max, int boxes, int order, int cull, uint64_t *out)
{
    Lab &L = *(Lab *)h;
    const Bvh8 &bvh = *L.bvh;
    for (int k = 0; k < 4; ++k) out[k] = 0;
    struct Ent { uint32_t node; float t; int pushed_at; bool first; };
    std::vector<Ent> st; st.reserve(256);
    for (uint64_t r = 0; r < n; ++r) {
        f3 o = mk3(rays[r * 6], rays[r * 6 + 1], rays[r * 6 + 2]), d = mk3(rays[r * 6 + 3], rays[r * 6 + 4], rays[r * 6 + 5]);
        float inv[3]; float ix, iy, iz; uint32_t oct = ray_slab_setup(o, d, ix, iy, iz); inv[0] = ix; inv[1] = iy; inv[2] = iz;
        const float org[3] = {o.x, o.y, o.z};
        float best = t_max; uint32_t bi = 0xFFFFFFFFu, bp = 0xFFFFFFFFu; bool hit = false;
        out[0]++;
        st.clear();
        if (!bvh.nodes.empty() && !(oct & kRayDegenerate)) st.push_back({0u, t_min, 0, true});
        const uint32_t octinv = oct & 7u;
        struct Pend { uint32_t tri; int due; };
        std::vector<Pend> pend;
        int visit = 0;
        auto run_tests = [&](bool all) {
            size_t w = 0;
            for (size_t q = 0; q < pend.size(); ++q) {
                if (!all && pend[q].due > visit) { pend[w++] = pend[q]; continue; }
                const TriRec &tr = L.tris[pend[q].tri];
                out[3]++;
                float t, u, v;
                if (tri_test(o, d, mk3(tr.v0[0], tr.v0[1], tr.v0[2]), mk3(tr.v1[0], tr.v1[1], tr.v1[2]), mk3(tr.v2[0], tr.v2[1], tr.v2[2]), t_min, t_max, t, u, v) &&
                    hit_better(t, tr.instance, tr.prim, best, bi, bp)) { best = t; bi = tr.instance; bp = tr.prim; hit = true; }
            }
            pend.resize(w);
        };
        while (!st.empty() || !pend.empty()) {
            if (st.empty()) { run_tests(true); continue; }
            run_tests(false);
            Ent e = st.back(); st.pop_back();
            ++visit;
            if (cull && e.t > best) { if (e.first && e.pushed_at == visit) g_cull_first++; else g_cull_later++; continue; }
            const Bvh8Node &nd = bvh.nodes[e.node];
            out[2]++;
            float t_frame = t_min;
            if (cull == 2) {
                float tf = best;
                for (int a = 0; a < 3; ++a) {
                    float stp = step_of(nd.e[a]);
                    float t0 = (nd.p[a] - org[a]) * inv[a], t1 = (nd.p[a] + 255.0f * stp - org[a]) * inv[a];
                    t_frame = fmaxf(t_frame, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                }
                uint32_t u; memcpy(&u, &t_frame, 4); u &= 0xFFFF0000u; memcpy(&t_frame, &u, 4);   // truncation of a positive float rounds down
            }
            if (cull == 3) t_frame = e.t;
            struct C { float t; int s; } hits[8]; int nh = 0;
            for (int s = 0; s < 8; ++s) {
                if (!nd.meta[s]) continue;
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    if (boxes == 0) { float stp = step_of(nd.e[a]); lo[a] = nd.p[a] + nd.qlo[a][s] * stp; hi[a] = nd.p[a] + nd.qhi[a][s] * stp; }
                    else if (boxes == 1) { lo[a] = L.child[e.node * 8 + s].lo[a]; hi[a] = L.child[e.node * 8 + s].hi[a]; }
                    else {
                        const Box &nb = L.node_box[e.node]; float ext = nb.hi[a] - nb.lo[a]; float stp = ext > 0 ? ext / 2047.0f : 1e-30f;
                        const Box &cb = L.child[e.node * 8 + s];
                        lo[a] = nb.lo[a] + std::floor((cb.lo[a] - nb.lo[a]) / stp) * stp; hi[a] = nb.lo[a] + std::ceil((cb.hi[a] - nb.lo[a]) / stp) * stp;
                    }
                }
                float tn = t_min, tf = best;
                for (int a = 0; a < 3; ++a) {
                    float t0 = (lo[a] - org[a]) * inv[a], t1 = (hi[a] - org[a]) * inv[a];
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                }
                if (tn <= tf) { hits[nh].t = tn; hits[nh].s = s; ++nh; }
            }
            // leaves first (their triangles are tested right away, like the kernels' queue), then inner children pushed far to near
            for (int k = 0; k < nh; ++k) {
                int s = hits[k].s;
                if ((nd.imask >> s) & 1u) continue;
                uint32_t off = nd.meta[s] & 31u, cnt = bq_popc((uint32_t)nd.meta[s] >> 5);
                for (uint32_t q = 0; q < cnt; ++q) pend.push_back({nd.tri_base + off + q, visit + g_delay});
                if (g_delay == 0) run_tests(true);
            }
            C inner[8]; int ni = 0;
            for (int k = 0; k < nh; ++k) if ((nd.imask >> hits[k].s) & 1u) inner[ni++] = hits[k];
            if (order == 0) std::sort(inner, inner + ni, [&](const C &a, const C &b) { return ((a.s ^ octinv)) < ((b.s ^ octinv)); });   // pop order: highest (slot ^ octinv) first
            else std::sort(inner, inner + ni, [](const C &a, const C &b) { return a.t > b.t; });                                       // far first -> near popped first
            float t_all = INFINITY, t_rest = INFINITY;
            for (int k = 0; k < ni; ++k) { t_all = fminf(t_all, inner[k].t); if (k + 1 < ni) t_rest = fminf(t_rest, inner[k].t); }
            for (int k = 0; k < ni; ++k) {
                int s = inner[k].s;
                float tt = cull == 2 ? t_frame : cull == 4 ? t_all : cull == 5 ? (k + 1 < ni ? t_rest : t_min) : inner[k].t;
                st.push_back({nd.child_base + bq_popc(nd.imask & ((1u << s) - 1u)), tt, visit, k + 1 == ni});
            }
        }
        if (hit) out[1]++;
    }
}
}
