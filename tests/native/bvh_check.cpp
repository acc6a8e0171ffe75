// bvh_check.cpp -- TEST-ONLY host build of the product's BVH8 builder + traversal (the same
// headers the HIP kernels compile), so that builder and traversal logic can be checked against
// the oracle's brute-force closest hit on a machine without a GPU.  Not part of libptx_amd.so.
#include "../../path_tracer_amd/csrc/bvh8.h"
#include "../../path_tracer_amd/csrc/traverse.h"

#include <vector>

using namespace ptx;

namespace {
struct HostStack {
    uint32_t b[64], m[64]; int sp = 0; int max_sp = 0;
    void push(uint32_t base, uint32_t mask) { b[sp] = base; m[sp] = mask; ++sp; if (sp > max_sp) max_sp = sp; }
    void pop(uint32_t &base, uint32_t &mask) { --sp; base = b[sp]; mask = m[sp]; }
    bool empty() const { return sp == 0; }
    void clear() { sp = 0; }
};
struct Check {
    Bvh8 bvh;
    std::vector<TriRec> tris;
};
}  // namespace

extern "C" {

// verts: 9 floats per triangle; ids: (instance, prim) per triangle
void *bvhcheck_create(const float *verts, const uint32_t *ids, uint64_t n_tris, int threads)
{
    Check *c = new Check;
    if (!build_bvh8(verts, n_tris, c->bvh, threads)) { delete c; return nullptr; }
    c->tris.resize(c->bvh.tri_order.size());
    for (size_t i = 0; i < c->tris.size(); ++i) {
        uint32_t t = c->bvh.tri_order[i];
        TriRec &r = c->tris[i];
        for (int a = 0; a < 3; ++a) { r.v0[a] = verts[t * 9 + a]; r.v1[a] = verts[t * 9 + 3 + a]; r.v2[a] = verts[t * 9 + 6 + a]; }
        r.instance = ids[t * 2]; r.prim = ids[t * 2 + 1]; r.rank = (uint32_t)t;
    }
    return c;
}
// moved triangles, same topology: refit the tree in place.  Returns 1 on success and the cost figure in *sah.
int bvhcheck_refit(void *h, const float *verts, const uint32_t *ids, uint64_t n_tris, double *sah)
{
    Check *c = (Check *)h;
    if (!refit_bvh8(c->bvh, verts, n_tris, sah, 4)) return 0;
    for (size_t i = 0; i < c->tris.size(); ++i) {
        uint32_t t = c->bvh.tri_order[i];
        TriRec &r = c->tris[i];
        for (int a = 0; a < 3; ++a) { r.v0[a] = verts[t * 9 + a]; r.v1[a] = verts[t * 9 + 3 + a]; r.v2[a] = verts[t * 9 + 6 + a]; }
        r.instance = ids[t * 2]; r.prim = ids[t * 2 + 1];
    }
    return 1;
}
void bvhcheck_destroy(void *h) { delete (Check *)h; }
double bvhcheck_built_sah(void *h) { return ((Check *)h)->bvh.built_sah; }
// FNV-1a over the node array, the triangle order and the level table: equal for equal trees
uint64_t bvhcheck_hash(void *h)
{
    Check *c = (Check *)h;
    uint64_t x = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t nb) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < nb; ++i) { x ^= b[i]; x *= 1099511628211ull; } };
    mix(c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(Bvh8Node));
    mix(c->bvh.tri_order.data(), c->bvh.tri_order.size() * sizeof(uint32_t));
    mix(c->bvh.level_first.data(), c->bvh.level_first.size() * sizeof(uint32_t));
    return x;
}
void bvhcheck_info(void *h, uint64_t *n_nodes, uint64_t *n_tris, uint32_t *depth)
{
    Check *c = (Check *)h;
    *n_nodes = c->bvh.nodes.size(); *n_tris = c->tris.size(); *depth = c->bvh.depth;
}
// rays: 6 floats each (origin, direction). out: per ray t,u,v (float) and instance, prim (u32; ~0 on miss)
void bvhcheck_trace(void *h, const float *rays, uint64_t n, float t_min, float t_max, float *tuv, uint32_t *ids,
                    uint64_t *nodes_visited, uint64_t *tris_tested, uint32_t *max_stack)
{
    Check *c = (Check *)h;
    uint64_t nv = 0, tt = 0; int ms = 0;
    for (uint64_t i = 0; i < n; ++i) {
        HostStack st;
        f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        Traversal<HostStack, true> tr;
        tr.init((uint32_t)c->bvh.nodes.size(), o, d, t_min, t_max, st);
        while (!tr.step(c->bvh.nodes.data(), c->tris.data(), st)) {}
        const HitRec &hit = tr.hit; const TraverseStats &ts = tr.stats;
        nv += ts.nodes; tt += ts.tris; if (st.max_sp > ms) ms = st.max_sp;
        tuv[i * 3] = hit.t; tuv[i * 3 + 1] = hit.u; tuv[i * 3 + 2] = hit.v;
        if (hit.tri != 0xFFFFFFFFu) { ids[i * 2] = c->tris[hit.tri].instance; ids[i * 2 + 1] = c->tris[hit.tri].prim; }
        else { ids[i * 2] = 0xFFFFFFFFu; ids[i * 2 + 1] = 0xFFFFFFFFu; }
    }
    if (nodes_visited) *nodes_visited = nv;
    if (tris_tested) *tris_tested = tt;
    if (max_stack) *max_stack = (uint32_t)ms;
}
}
