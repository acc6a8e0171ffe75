// sanitize_main.cpp -- TEST-ONLY driver, built with -fsanitize=address,undefined (host code only; GPU sanitizers are
// not available on this pool).  Exercises the native host pieces that parse untrusted bytes or do index arithmetic:
// the PNG / baseline-JPEG decoders (whole files, truncations, bit flips), the JSON parser behind the glTF import, and
// the BVH8 builder + refit + traversal on random triangle soups.  Any sanitizer report aborts with a non-zero status.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../path_tracer_amd/csrc/bvh8.h"
#include "../../path_tracer_amd/csrc/traverse.h"
#include "../../path_tracer_amd/host/pth_image.h"
#include "../../path_tracer_amd/host/pth_json.h"

using namespace ptx;

namespace {
struct HostStack {
    uint32_t b[64], m[64]; int sp = 0;
    void push(uint32_t base, uint32_t mask) { if (sp >= 64) abort(); b[sp] = base; m[sp] = mask; ++sp; }
    void pop(uint32_t &base, uint32_t &mask) { if (sp <= 0) abort(); --sp; base = b[sp]; mask = m[sp]; }
    bool empty() const { return sp == 0; }
    void clear() { sp = 0; }
};

std::vector<uint8_t> slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    std::string s = ss.str();
    return std::vector<uint8_t>(s.begin(), s.end());
}

int decode_bytes(const std::vector<uint8_t> &bytes, const std::string &tmp)
{
    { std::ofstream o(tmp, std::ios::binary); o.write((const char *)bytes.data(), (std::streamsize)bytes.size()); }
    pth::Image img; std::string err;
    bool ok = pth::decode_image_file(tmp, img, err);
    if (ok && img.pixels.size() != (size_t)img.width * img.height * img.channels) { fprintf(stderr, "bad pixel count\n"); exit(3); }
    return ok ? 1 : 0;
}
}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: sanitize_main <tmpfile> <image>...\n"); return 2; }
    std::string tmp = argv[1];
    std::mt19937 rng(12345);
    int decoded = 0, rejected = 0;
    for (int i = 2; i < argc; ++i) {
        if (!strncmp(argv[i], "bad:", 4)) {                // a crafted malformed file: must be rejected cleanly
            std::vector<uint8_t> b = slurp(argv[i] + 4);
            if (b.empty()) { fprintf(stderr, "cannot read %s\n", argv[i] + 4); return 2; }
            if (decode_bytes(b, tmp)) { fprintf(stderr, "%s decodes, it must not\n", argv[i] + 4); return 3; }
            ++rejected;
            continue;
        }
        std::vector<uint8_t> d = slurp(argv[i]);
        if (d.empty()) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
        if (!decode_bytes(d, tmp)) { fprintf(stderr, "%s does not decode\n", argv[i]); return 3; }
        ++decoded;
        for (int k = 0; k < 24; ++k) {                     // truncations
            std::vector<uint8_t> t(d.begin(), d.begin() + (size_t)(rng() % d.size()));
            (decode_bytes(t, tmp) ? decoded : rejected)++;
        }
        for (int k = 0; k < 48; ++k) {                     // byte corruption, mostly in the headers
            std::vector<uint8_t> t = d;
            size_t span = (k & 1) ? t.size() : std::min<size_t>(t.size(), 1024);
            for (int j = 0; j < 1 + (k % 4); ++j) t[rng() % span] = (uint8_t)rng();
            (decode_bytes(t, tmp) ? decoded : rejected)++;
        }
    }
    // JSON parser: valid documents, truncations and corruptions must throw or parse, never crash
    const std::string doc = "{\"asset\":{\"version\":\"2.0\"},\"nodes\":[{\"mesh\":0,\"translation\":[1,2.5e-3,-3E2],\"name\":\"a\\n\\u00e9\"}],"
                            "\"meshes\":[{\"primitives\":[{\"attributes\":{\"POSITION\":0},\"indices\":1}]}],\"x\":[true,false,null,[],{}]}";
    int parsed = 0, thrown = 0;
    for (int k = 0; k < 400; ++k) {
        std::string s = doc;
        if (k > 0 && k < 150) s = s.substr(0, rng() % s.size());
        else if (k >= 150) for (int j = 0; j < 1 + k % 3; ++j) s[rng() % s.size()] = (char)(rng() % 128);
        try { pth::Json j = pth::JsonParser(s).parse(); (void)j; ++parsed; } catch (const std::exception &) { ++thrown; }
    }
    // BVH8: random soups (with degenerate, coincident and non-finite triangles), build, traverse, move, refit, traverse
    uint64_t hits = 0;
    for (int round = 0; round < 6; ++round) {
        size_t n = (size_t[]){1, 2, 9, 200, 5000, 40000}[round];
        std::vector<float> v(n * 9);
        std::uniform_real_distribution<float> U(-1.0f, 1.0f);
        for (size_t t = 0; t < n; ++t) {
            float c[3] = {U(rng) * 10, U(rng) * 10, U(rng) * 10};
            for (int k = 0; k < 9; ++k) v[t * 9 + k] = c[k % 3] + U(rng) * (t % 7 == 0 ? 0.0f : 0.3f);
            if (t % 1000 == 999) v[t * 9 + 4] = NAN;
            if (t % 50 == 3 && t) memcpy(&v[t * 9], &v[(t - 1) * 9], 36);
        }
        Bvh8 bvh;
        if (!build_bvh8(v.data(), n, bvh, 4)) { fprintf(stderr, "build failed\n"); return 3; }
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<TriRec> tris(bvh.tri_order.size());
            for (size_t i = 0; i < tris.size(); ++i) {
                uint32_t t = bvh.tri_order[i];
                memcpy(tris[i].v0, &v[t * 9], 12); memcpy(tris[i].v1, &v[t * 9 + 3], 12); memcpy(tris[i].v2, &v[t * 9 + 6], 12);
                tris[i].instance = 0; tris[i].prim = t; tris[i].rank = t;
            }
            for (int r = 0; r < 3000; ++r) {
                HostStack st;
                f3 o = mk3(U(rng) * 12, U(rng) * 12, U(rng) * 12), d = mk3(U(rng), U(rng), r % 17 ? U(rng) : 0.0f);
                Traversal<HostStack, true> tr;
                if (tr.init((uint32_t)bvh.nodes.size(), o, d, 1e-3f, 1e4f, st))
                    while (!tr.step(bvh.nodes.data(), tris.data(), st)) {}
                if (tr.hit.tri != 0xFFFFFFFFu) { if (tr.hit.tri >= tris.size()) abort(); ++hits; }
            }
            // move everything (finite triangles stay finite) and refit
            for (size_t t = 0; t < n; ++t) for (int k = 0; k < 9; ++k) v[t * 9 + k] = v[t * 9 + k] * 1.1f + (k % 3 == 0 ? 0.5f : 0.0f);
            double sah = 0;
            if (!refit_bvh8(bvh, v.data(), n, &sah, 4)) { fprintf(stderr, "refit failed\n"); return 3; }
        }
    }
    printf("images decoded %d rejected %d; json parsed %d thrown %d; bvh hits %llu\n", decoded, rejected, parsed, thrown, (unsigned long long)hits);
    return 0;
}
