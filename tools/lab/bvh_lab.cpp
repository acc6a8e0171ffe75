// bvh_lab.cpp -- developer experiment (not part of the product or the tests): host build of the BVH8 builder +
// traversal headers that counts, per closest-hit query, node visits, visits in which no child box was hit, and
// how many of those were stale (the visited node's own frame box already lies beyond the closest hit found since
// it was pushed).  Used to decide which traversal optimisations are worth their instructions.
#include "../../path_tracer_amd/csrc/bvh8.h"
#include "../../path_tracer_amd/csrc/traverse.h"

#include <cstdio>
#include <vector>

using namespace ptx;

namespace {
struct HostStack {
    uint32_t b[64], m[64]; int sp = 0;
    void push(uint32_t base, uint32_t mask) { b[sp] = base; m[sp] = mask; ++sp; }
    void pop(uint32_t &base, uint32_t &mask) { --sp; base = b[sp]; mask = m[sp]; }
    bool empty() const { return sp == 0; }
    void clear() { sp = 0; }
};
}

extern "C" {
void *lab_build(const float *verts, uint64_t n_tris, int threads, uint64_t *n_nodes)
{
    Bvh8 *b = new Bvh8;
    if (!build_bvh8(verts, n_tris, *b, threads)) { delete b; return nullptr; }
    *n_nodes = b->nodes.size();
    return b;
}
// out[0..7]: rays, hits, node visits, empty visits, stale visits (frame box beyond hit.t), tri tests, child slots used, child boxes hit
void lab_trace(void *h, const float *verts, const float *rays, uint64_t n, float t_min, float t_max, uint64_t *out, float *t_out)
{
    Bvh8 &bvh = *(Bvh8 *)h;
    std::vector<TriRec> tris(bvh.tri_order.size());
    for (size_t i = 0; i < tris.size(); ++i) {
        uint32_t t = bvh.tri_order[i];
        for (int a = 0; a < 3; ++a) { tris[i].v0[a] = verts[t * 9 + a]; tris[i].v1[a] = verts[t * 9 + 3 + a]; tris[i].v2[a] = verts[t * 9 + 6 + a]; }
        tris[i].instance = 0; tris[i].prim = t; tris[i].rank = t;
    }
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        HostStack st;
        f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        Traversal<HostStack, true> tr;
        bool go = tr.init((uint32_t)bvh.nodes.size(), o, d, t_min, t_max, st);
        out[0]++;
        while (go) {
            if (tr.wants_node()) {
                if (!tr.next_group(st)) break;
                uint32_t ni = tr.node_pick(st);
                const Bvh8Node &nd = bvh.nodes[ni];
                // the node's own frame box [p, p + 255*step] against the current closest hit
                float tn = t_min, tf = tr.hit.t;
                const float inv[3] = {tr.idx, tr.idy, tr.idz}, org[3] = {o.x, o.y, o.z};
                for (int a = 0; a < 3; ++a) {
                    uint32_t u = (uint32_t)nd.e[a] << 23; float step; memcpy(&step, &u, 4);
                    float t0 = (nd.p[a] - org[a]) * inv[a], t1 = (nd.p[a] + 255.0f * step - org[a]) * inv[a];
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                }
                bool stale = !(tn <= tf);
                tr.node_eval(load_node(bvh.nodes.data(), ni));
                out[2]++;
                for (int s = 0; s < 8; ++s) if (nd.meta[s]) out[6]++;
                uint32_t hm = (tr.ng_mask >> 24);
                out[7] += popc32(hm) + popc32(tr.tg_mask);   // (triangle bits: per triangle, not per leaf)
                if (!(tr.ng_mask & 0xFF000000u) && tr.tg_mask == 0) { out[3]++; if (stale) out[4]++; }
            }
            if (tr.wants_tri()) { tr.tri_step(tris.data()); out[5]++; }
        }
        if (tr.hit.tri != 0xFFFFFFFFu) out[1]++;
        if (t_out) t_out[i] = tr.hit.tri != 0xFFFFFFFFu ? tr.hit.t : -1.0f;
    }
}
}
