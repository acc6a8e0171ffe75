// bvh8.h -- compressed 8-wide BVH over world-space triangles: the software stand-in for the
// TLAS/BLAS pair the reference builds in the Vulkan driver (renderer.rs:1196-1388,
// scene/geometry/mod.rs:136-278).  CDNA4 has no ray-tracing units, so both build and traversal
// are the build's own.
//
// Layout (after Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs Through
// Compressed Wide BVHs", HPG 2017): an 80-byte node holds 8 child boxes quantised to 8 bits per
// plane on a per-node power-of-two grid.  A lane fetches a node with five 16-byte loads.
#pragma once

#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace ptx {

struct alignas(16) Bvh8Node {
    float p[3];            // quantisation origin = node box low corner
    uint8_t e[3];          // per-axis grid step as a biased fp32 exponent: step = bits(e << 23)
    uint8_t imask;         // bit s set: slot s is an internal node
    uint32_t child_base;   // index of this node's first internal child; the rest follow in slot order
    uint32_t tri_base;     // index of the first triangle referenced by this node's leaf slots
    uint8_t meta[8];       // internal: 0b001'11sss (s = slot); leaf: unary(count 1..3) << 5 | offset 0..23; empty: 0
    uint8_t qlo[3][8];     // quantised child box planes [axis][slot]
    uint8_t qhi[3][8];
};
static_assert(sizeof(Bvh8Node) == 80, "Bvh8Node must be 80 bytes");

// Triangle in traversal (leaf) order, 48 B = three 16-byte loads.
struct alignas(16) TriRec {
    float v0[3]; uint32_t instance;
    float v1[3]; uint32_t prim;
    float v2[3]; uint32_t rank;       // position in (instance, primitive) order = index in the builder's input soup: the tie-break key of equal-distance hits
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

struct Bvh8 {
    std::vector<Bvh8Node> nodes;     // nodes[0] is the root (empty vector: empty scene)
    std::vector<uint32_t> tri_order; // leaf order -> input triangle index
    uint32_t depth = 0;              // levels of Bvh8Node below and including the root
    std::vector<uint32_t> level_first; // nodes are stored level by level (breadth first): first node of each level, then nodes.size()
    double built_sah = 0.0;          // the cost figure refit_bvh8 reports (sum of node half-areas / root half-area), of the built tree
};

// `verts`: 9 floats per triangle (world space).  Deterministic for a given input.
// Returns false (and leaves `out` empty) if a coordinate is not finite.
bool build_bvh8(const float *verts, size_t n_tris, Bvh8 &out, int n_threads);

// Same tree, moved triangles (same indexing as the build's input): recomputes every box bottom-up.
// Returns false if the set of finite triangles changed (the caller rebuilds).  *sah: sum of node
// half-areas / root half-area, a traversal-cost figure to compare against the freshly built tree's.
bool refit_bvh8(Bvh8 &bvh, const float *verts, size_t n_tris, double *sah, int n_threads = 1);

}  // namespace ptx
