// bvh_build.cpp -- host builder: binned-SAH BVH2 -> greedy collapse to 8-wide -> quantise.
// Replaces the driver-side BLAS/TLAS builds of the reference (scene/geometry/mod.rs:136-278,
// renderer.rs:1196-1388).  Plain C++17, no HIP.
#include "bvh8.h"
#include "bvh_quant.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <thread>

namespace ptx {
namespace {

float env_float(const char *name, float dflt);

struct Node2 {
    Box box;
    uint32_t left = 0, right = 0;   // children (inner)
    uint32_t first = 0, count = 0;  // leaf range in `order` (count > 0 => leaf)
    uint32_t tfirst = 0, tcount = 0; // triangle range of the whole subtree (contiguous in `order`)
};

// One triangle as the builder moves it around: box, centroid and index travel together, so that every pass
// over a range of `prims` is a sequential stream (an index array into separate box / centroid arrays made
// each pass a random gather: 2 x the build time on the Sponza stand-in).
struct Prim {
    Box box;
    float c[3];
    uint32_t tri;
};

struct Builder2 {
    std::vector<Prim> prims;
    std::vector<Node2> nodes;
    std::atomic<uint32_t> n_nodes{0};

    static constexpr int kBins = 16;
    static constexpr uint32_t kMaxLeaf = 3;

    uint32_t alloc2() { return n_nodes.fetch_add(2); }

    void build(uint32_t node, uint32_t first, uint32_t count, int spawn_depth)
    {
        Node2 &n = nodes[node];
        n.tfirst = first; n.tcount = count;
        Box cb; cb.reset(); n.box.reset();
        const Prim *pr = prims.data();
        for (uint32_t i = first; i < first + count; ++i) {
            n.box.grow(pr[i].box);
            for (int a = 0; a < 3; ++a) {
                cb.lo[a] = std::min(cb.lo[a], pr[i].c[a]);
                cb.hi[a] = std::max(cb.hi[a], pr[i].c[a]);
            }
        }
        if (count == 1) { n.first = first; n.count = 1; return; }

        // binned SAH over the three axes
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; ++a) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f)) continue;
            Box bins[kBins]; uint32_t cnt[kBins];
            for (int b = 0; b < kBins; ++b) { bins[b].reset(); cnt[b] = 0; }
            float k = kBins / ext;
            for (uint32_t i = first; i < first + count; ++i) {
                int b = std::min(kBins - 1, std::max(0, (int)((pr[i].c[a] - cb.lo[a]) * k)));
                bins[b].grow(pr[i].box); cnt[b]++;
            }
            float right_area[kBins]; uint32_t right_cnt[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(bins[b]); c += cnt[b];
                right_area[b] = acc.half_area(); right_cnt[b] = c;
            }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bins[b]); c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        // leaf if it is cheaper than the best split (cost model: node visit 1.5, triangle test 1)
        if (count <= kMaxLeaf) {
            float leaf_cost = (float)count * n.box.half_area();
            if (best_axis < 0 || best_cost + 1.5f * n.box.half_area() >= leaf_cost) {
                n.first = first; n.count = count; return;
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = first + count / 2;       // coincident centroids: split the range
        } else {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float k = kBins / ext, lo = cb.lo[best_axis];
            int a = best_axis, bb = best_bin;
            auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim &p) {
                int b = std::min(kBins - 1, std::max(0, (int)((p.c[a] - lo) * k)));
                return b <= bb;
            });
            mid = (uint32_t)(it - prims.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        uint32_t l = alloc2();
        n.left = l; n.right = l + 1; n.count = 0;
        uint32_t lc = mid - first, rc = count - lc;
        if (spawn_depth > 0 && count > 8192) {
            auto fut = std::async(std::launch::async, [=]() { build(l, first, lc, spawn_depth - 1); });
            build(l + 1, mid, rc, spawn_depth - 1);
            fut.get();
        } else {
            build(l, first, lc, 0);
            build(l + 1, mid, rc, 0);
        }
    }
};

float env_float(const char *name, float dflt)
{
    const char *e = getenv(name);
    return e ? (float)atof(e) : dflt;
}

// static slices of [0, n) over up to n_threads threads
template <class F> void par_for(size_t n, int n_threads, F &&fn)
{
    int nt = std::max(1, std::min(n_threads, 64));
    if (nt == 1 || n < 4096) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    size_t per = (n + (size_t)nt - 1) / (size_t)nt;
    for (int t = 0; t < nt; ++t) {
        size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back([=, &fn] { fn(lo, hi); });
    }
    for (std::thread &t : th) t.join();
}

// Boxes of the finite triangles, padded so that box culling can never contradict the triangle test (the
// test's edge functions are exact only up to a few ulps of the coordinate magnitude).  Returns the indices
// of the finite triangles; a triangle with a non-finite coordinate can never be hit and is left out.
std::vector<uint32_t> padded_tri_boxes(const float *verts, size_t n_tris, std::vector<Box> &tri_box, int n_threads = 1)
{
    tri_box.resize(n_tris);
    int nt = std::max(1, std::min(n_threads, 64));
    std::vector<Box> part((size_t)nt);
    for (Box &b : part) b.reset();
    size_t per = (n_tris + (size_t)nt - 1) / (size_t)nt;
    par_for((size_t)nt, n_tris < 4096 ? 1 : nt, [&](size_t plo, size_t phi) {
        for (size_t pi = plo; pi < phi; ++pi) {
            size_t lo = std::min(n_tris, pi * per), hi = std::min(n_tris, lo + per);
            Box sb; sb.reset();
            for (size_t t = lo; t < hi; ++t) {
                Box b;
                if (!tri_raw_box(verts + t * 9, b)) { tri_box[t].reset(); continue; }
                tri_box[t] = b; sb.grow(b);
            }
            part[pi] = sb;
        }
    });
    std::vector<uint32_t> order; order.reserve(n_tris);
    for (size_t t = 0; t < n_tris; ++t) if (tri_box[t].lo[0] <= tri_box[t].hi[0]) order.push_back((uint32_t)t);
    if (order.empty()) return order;
    Box sb; sb.reset();
    for (const Box &b : part) if (b.lo[0] <= b.hi[0]) sb.grow(b);
    float scene_ext = 0.0f;
    for (int a = 0; a < 3; ++a) scene_ext = std::max(scene_ext, sb.hi[a] - sb.lo[a]);
    par_for(n_tris, n_threads, [&](size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; ++t) {
            Box &b = tri_box[t];
            if (!b.valid()) continue;
            pad_tri_box(b, scene_ext);
        }
    });
    return order;
}

}  // namespace

bool build_bvh8(const float *verts, size_t n_tris, Bvh8 &out, int n_threads)
{
    out.nodes.clear(); out.tri_order.clear(); out.depth = 0; out.built_sah = 0.0;
    if (n_tris == 0) return true;
    if (n_tris > 1000000000u) return false;

    static const bool verbose = getenv("PTX_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ptx] build_bvh8 %-10s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<Box> tri_box;
    std::vector<uint32_t> order = padded_tri_boxes(verts, n_tris, tri_box, n_threads);
    if (order.empty()) return true;
    lap("boxes");

    Builder2 b2;
    uint32_t n = (uint32_t)order.size();
    b2.prims.resize(n);
    par_for(n, n_threads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            Prim &p = b2.prims[i];
            p.tri = order[i]; p.box = tri_box[p.tri];
            for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * p.box.lo[a] + 0.5f * p.box.hi[a];
        }
    });
    b2.nodes.resize((size_t)2 * n + 2);
    b2.n_nodes = 1;
    // SAH splits are lopsided: 2 x more subtrees than threads, so that no thread is left with one huge subtree.  (Every
    // split above the spawn depth holds a thread until its subtree is done: 2^(spawn + 1) threads at most per build -- eight
    // ranks building at once on one node stay far below any task limit.)
    int spawn = 0;
    while ((1 << spawn) < std::max(1, n_threads)) ++spawn;
    if (n_threads > 1) spawn += 1;
    lap("setup");
    b2.build(0, 0, n, spawn);
    lap("bvh2");

    // ---- collapse to 8-wide ----
    // Which BVH2 nodes become the (up to 8) children of each wide node is chosen by the dynamic
    // programme of Ylitie, Karras & Laine, "Efficient Incoherent Ray Traversal on GPUs Through
    // Compressed Wide BVHs" (HPG 2017, section 3.1): C(n, i) = cheapest SAH cost of the subtree of
    // n when it may occupy i child slots of its parent; a subtree of <= 3 triangles may become one
    // leaf slot.  A greedy open-the-largest-child collapse left the nodes half empty (4.1 of 8
    // slots on the Sponza stand-in), and every visit pays for all 8 slots.
    const uint32_t n2 = b2.n_nodes.load();
    const float c_node = env_float("PTX_BVH_CNODE", 1.8f), c_prim = env_float("PTX_BVH_CPRIM", 1.0f);
    // developer experiment (tools/lab/trav_lab.py): nodes of at most W children in the same node format
    const int W = std::max(2, std::min(8, (int)env_float("PTX_LAB_BVH_WIDTH", 8.0f)));
    const float inf = std::numeric_limits<float>::infinity();
    struct Dp { float c[8]; uint8_t k[9]; uint8_t leaf; };      // c[i], i = 1..7; k[j], j = 2..8: slots given to the left child
    std::vector<Dp> dp(n2);
    for (uint32_t ni = n2; ni-- > 0;) {                          // children have larger indices than their parent
        const Node2 &nd2 = b2.nodes[ni];
        if (ni != 0 && nd2.tcount == 0) continue;                // the unused second half of an allocation
        Dp &d = dp[ni];
        float area = nd2.box.half_area();
        float c_leaf = nd2.tcount <= Builder2::kMaxLeaf ? area * (float)nd2.tcount * c_prim : inf;
        if (nd2.count > 0) {
            for (int i = 1; i <= 7; ++i) d.c[i] = c_leaf;
            d.leaf = 1;
            continue;
        }
        const Dp &L = dp[nd2.left], &R = dp[nd2.right];
        float dist[9];
        for (int j = 2; j <= 8; ++j) {
            float best = inf; int bk = 1;
            for (int k = 1; k < j; ++k) {
                if (k > 7 || j - k > 7) continue;
                float v = L.c[k] + R.c[j - k];
                if (v < best) { best = v; bk = k; }
            }
            if (j > W) best = inf;
            dist[j] = best; d.k[j] = (uint8_t)bk;
        }
        float c_internal = dist[W] + area * c_node;
        d.leaf = c_leaf <= c_internal;
        d.c[1] = d.leaf ? c_leaf : c_internal;
        for (int i = 2; i <= 7; ++i) d.c[i] = std::min(dist[i], d.c[i - 1]);
    }
    lap("dp");
    // children of the wide node rooted at BVH2 node `n` when it may use `i` slots
    struct Gather {
        const std::vector<Node2> &nodes; const std::vector<Dp> &dp;
        void run(uint32_t n, int i, uint32_t *kids, int &nk) const
        {
            const Dp &d = dp[n];
            while (i > 1 && !(d.c[i] < d.c[i - 1])) --i;         // C(n, i) fell back to fewer slots
            if (i == 1 || nodes[n].count > 0) { kids[nk++] = n; return; }
            int k = d.k[i];
            run(nodes[n].left, k, kids, nk);
            run(nodes[n].right, i - k, kids, nk);
        }
    } gather{b2.nodes, dp};

    // Breadth-first, one level at a time: (1) per node, in parallel: its children and their slots; (2) a serial
    // prefix over the level: where each node's children and leaf triangles start; (3) per node, in parallel:
    // the quantised node and its part of the triangle order.  Node indices = breadth-first positions.
    struct Wide { uint32_t node2; uint32_t kids[8]; int8_t slot_child[8]; uint32_t n_inner, n_leaf_tris, child_base, tri_base; };
    std::vector<Wide> level(1), next;
    level[0].node2 = 0;
    out.nodes.clear();
    out.tri_order.clear();
    out.level_first.clear();
    uint32_t n_nodes_total = 1, n_tris_total = 0;
    std::vector<float> node_area;             // half-area of every wide node's box, by node index
    while (!level.empty()) {
        out.level_first.push_back((uint32_t)out.nodes.size());
        out.depth++;
        par_for(level.size(), n_threads, [&](size_t lo, size_t hi) {
            for (size_t wi = lo; wi < hi; ++wi) {
                Wide &w = level[wi];
                const Node2 &src = b2.nodes[w.node2];
                uint32_t *kids = w.kids; int nk = 0;
                if (src.count > 0 || dp[w.node2].leaf) {
                    kids[nk++] = w.node2;             // a leaf root becomes the single leaf slot
                } else {
                    int k = dp[w.node2].k[W];
                    gather.run(src.left, k, kids, nk);
                    gather.run(src.right, W - k, kids, nk);
                }
                // slot assignment: slot s is visited first by rays whose direction is negative exactly
                // on the axes whose bit is set in s (bit2 = x, bit1 = y, bit0 = z), so it should hold the
                // child lying furthest along ds(s) = (+1 where the bit is set, -1 elsewhere).
                float center[3];
                for (int a = 0; a < 3; ++a) center[a] = 0.5f * src.box.lo[a] + 0.5f * src.box.hi[a];
                float cost[8][8];
                for (int c = 0; c < nk; ++c) {
                    const Box &cbx = b2.nodes[kids[c]].box;
                    float d[3];
                    for (int a = 0; a < 3; ++a) d[a] = (0.5f * cbx.lo[a] + 0.5f * cbx.hi[a]) - center[a];
                    for (int s = 0; s < 8; ++s)
                        cost[c][s] = ((s & 4) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 1) ? d[2] : -d[2]);
                }
                bool child_done[8] = {false};
                for (int s = 0; s < 8; ++s) w.slot_child[s] = -1;
                for (int it = 0; it < nk; ++it) {
                    float best = -std::numeric_limits<float>::infinity(); int bc = -1, bs = -1;
                    for (int c = 0; c < nk; ++c) {
                        if (child_done[c]) continue;
                        for (int s = 0; s < 8; ++s)
                            if (w.slot_child[s] < 0 && cost[c][s] > best) { best = cost[c][s]; bc = c; bs = s; }
                    }
                    w.slot_child[bs] = (int8_t)bc; child_done[bc] = true;
                }
                w.n_inner = 0; w.n_leaf_tris = 0;
                for (int s = 0; s < 8; ++s) {
                    int c = w.slot_child[s];
                    if (c < 0) continue;
                    if (!dp[kids[c]].leaf) ++w.n_inner; else w.n_leaf_tris += b2.nodes[kids[c]].tcount;
                }
            }
        });
        size_t level_start = out.nodes.size();
        n_nodes_total = (uint32_t)(level_start + level.size());
        size_t n_next = 0;
        for (Wide &w : level) {
            w.child_base = n_nodes_total + (uint32_t)n_next; w.tri_base = n_tris_total;
            n_next += w.n_inner; n_tris_total += w.n_leaf_tris;
        }
        out.nodes.resize(level_start + level.size());
        node_area.resize(out.nodes.size());
        out.tri_order.resize(n_tris_total);
        next.assign(n_next, Wide());
        par_for(level.size(), n_threads, [&](size_t lo, size_t hi) {
            for (size_t wi = lo; wi < hi; ++wi) {
                const Wide &w = level[wi];
                const Node2 &src = b2.nodes[w.node2];
                Bvh8Node nd; memset(&nd, 0, sizeof nd);
                set_node_frame(nd, src.box);
                nd.child_base = w.child_base;
                nd.tri_base = w.tri_base;
                uint32_t tri_off = 0, inner = 0;
                for (int s = 0; s < 8; ++s) {
                    int c = w.slot_child[s];
                    if (c < 0) continue;
                    const Node2 &ch = b2.nodes[w.kids[c]];
                    set_child_planes(nd, s, ch.box);
                    if (!dp[w.kids[c]].leaf) {
                        // internal children in slot order => relative index = popcount(imask below the slot)
                        nd.imask |= (uint8_t)(1u << s);
                        nd.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
                        next[(size_t)(w.child_base - n_nodes_total) + inner].node2 = w.kids[c];
                        ++inner;
                    } else {
                        nd.meta[s] = (uint8_t)((((1u << ch.tcount) - 1u) << 5) | tri_off);
                        for (uint32_t i = 0; i < ch.tcount; ++i) out.tri_order[(size_t)w.tri_base + tri_off + i] = b2.prims[ch.tfirst + i].tri;
                        tri_off += ch.tcount;
                    }
                }
                out.nodes[level_start + wi] = nd;
                node_area[level_start + wi] = src.box.half_area();
            }
        });
        level.swap(next);
    }
    out.level_first.push_back((uint32_t)out.nodes.size());
    {   // what refit_bvh8 would report for this tree: a node's refitted box is the union of its triangles' padded boxes,
        // which is the BVH2 node's box (min / max are exact), summed in the same order
        double area_sum = 0.0;
        for (float a : node_area) area_sum += (double)a;
        out.built_sah = node_area[0] > 0.0f ? area_sum / (double)node_area[0] : 0.0;
    }
    lap("collapse");
    if (verbose) {      // FNV-1a of the tree, to compare builder variants
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t nb) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < nb; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
        mix(out.nodes.data(), out.nodes.size() * sizeof(Bvh8Node));
        mix(out.tri_order.data(), out.tri_order.size() * sizeof(uint32_t));
        fprintf(stderr, "[ptx] build_bvh8 %zu nodes, %zu triangles, hash %016llx\n", out.nodes.size(), out.tri_order.size(), (unsigned long long)h);
    }
    return true;
}

// Refit: same topology, new triangle positions (instances moved).  Children have larger indices than their
// parents, so one backward sweep recomputes every node's box from its leaf triangles / inner children and
// re-quantises the child planes.  `sah` = sum of the node half-areas relative to the root's: the caller
// compares it with the value right after a build to decide when a moved scene deserves a full rebuild.
bool refit_bvh8(Bvh8 &bvh, const float *verts, size_t n_tris, double *sah, int n_threads)
{
    if (sah) *sah = 0.0;
    if (bvh.nodes.empty()) return true;
    std::vector<Box> tri_box;
    std::vector<uint32_t> finite = padded_tri_boxes(verts, n_tris, tri_box, n_threads);
    if (finite.size() != bvh.tri_order.size()) return false;      // a triangle became (non-)finite: rebuild
    for (uint32_t t : bvh.tri_order)
        if (!(tri_box[t].lo[0] <= tri_box[t].hi[0])) return false;  // a referenced triangle is no longer finite
    std::vector<Box> node_box(bvh.nodes.size());
    auto refit_node = [&](size_t i) {
        node_box[i] = refit_one_node(bvh.nodes[i], node_box.data(), [&](uint32_t k) { return tri_box[bvh.tri_order[k]]; });
    };
    // deepest level first; the nodes of one level are independent of each other
    std::vector<uint32_t> lf = bvh.level_first;
    if (lf.size() < 2 || lf.front() != 0 || lf.back() != bvh.nodes.size()) { lf.clear(); lf.push_back(0); }   // unknown layout: one node at a time, backwards
    if (lf.size() >= 2) {
        for (size_t l = lf.size() - 1; l-- > 0;) {
            size_t first = lf[l], count = lf[l + 1] - lf[l];
            par_for(count, n_threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) refit_node(first + i); });
        }
    } else {
        for (size_t i = bvh.nodes.size(); i-- > 0;) refit_node(i);
    }
    double area_sum = 0.0;
    for (const Box &b : node_box) area_sum += (double)b.half_area();
    double root = (double)node_box[0].half_area();
    if (sah) *sah = root > 0.0 ? area_sum / root : 0.0;
    return true;
}

}  // namespace ptx
