// ptx_api.cpp -- the C ABI of include/ptx.h: context, scene buffers, acceleration-structure
// build, wavefront render loop, readback.  Mirrors the host half of the reference's hot path:
// Renderer::new / record_commands / resize_render_buffer / readback (renderer.rs:183,583,668,906),
// Scene::add_* (scene/mod.rs:270-520) and Geometry::create (scene/geometry/mod.rs:54-103).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptx.h"
#include "bvh8.h"
#include "ptx_context_access.h"
#include "ptx_kernels.h"

using namespace ptx;

namespace {

thread_local std::string g_create_error;

struct Geometry { uint64_t first_vertex, n_vertices, first_index, n_indices; };

struct EventPair { hipEvent_t a, b; int kind; };   // kind 0 extend, 1 shade, 2 whole render call

// Wavefront state of one chunk of paths in flight.  Consecutive chunks (frames of the interactive loop,
// chunks of a batch) go to different sets on different streams, so the thin tail of one chunk's late
// bounces overlaps the next chunk's wide early ones; their resolves are chained by events, so the image
// receives its samples in accumulated_count order exactly as with one stream.
constexpr int kMaxSets = 4;
struct PathSet {
    hipStream_t stream = nullptr;
    uint64_t capacity = 0;
    void *d_rays[3] = {nullptr, nullptr, nullptr};   // two ray queues (four 16-byte planes each), ping-pong per bounce; [2]: occlusion rays (lit integrator, allocated on first use)
    HitOut *d_hits = nullptr;
    float4 *d_results = nullptr;
    uint32_t *d_counters = nullptr;
    uint32_t *d_order = nullptr, *d_class_count = nullptr;   // material sorting option of the lit integrator (allocated on first use)
    hipEvent_t resolved = nullptr;                   // recorded after the set's latest resolve
    hipEvent_t first_shade = nullptr;                // recorded after the set's latest bounce-0 shading (G-buffer / id image writes)
    // queue lengths of the set's latest chunk, copied to page-locked host memory behind the chunk (never waited for)
    uint32_t *h_counts = nullptr;
    hipEvent_t counts_ready = nullptr;
    bool counts_pending = false;
    uint32_t counts_paths = 0, counts_bounces = 0;
    bool counts_lit = false;
};

// ---- ptx_render_static: the wavefront state of one frame in memory that is never allocated at run time.
// The two ray queues, the hit records, the per-path results and the chunk counters of a frame of up to
// kStaticMaxPaths pixels live in ONE file-scope device array.  Every GPU the library runs on holds its own copy
// (31.59 MB) from module load until the process ends.  In host code the array's name is only a stand-in for the
// symbol: its device address is looked up once per device (static_scratch_for, called from ptx_create with that
// device current) and only that address is ever passed to a kernel.
constexpr uint64_t kStaticMaxPaths = 512ull * 512ull;
constexpr uint64_t kStaticCounterBytes = (kNumCounters * sizeof(uint32_t) + 255ull) & ~255ull;
constexpr uint64_t kStaticRayOff = kStaticCounterBytes;
constexpr uint64_t kStaticHitOff = kStaticRayOff + 2ull * kStaticMaxPaths * kRayBytes;
constexpr uint64_t kStaticResultOff = kStaticHitOff + kStaticMaxPaths * sizeof(HitOut);
constexpr uint64_t kStaticBytes = kStaticResultOff + kStaticMaxPaths * sizeof(float4);
static_assert(kStaticRayOff % 256 == 0 && kStaticHitOff % 256 == 0 && kStaticResultOff % 256 == 0, "sub-buffers stay 256-byte aligned");
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void *g_scratch_ptr[kMaxDevices] = {};      // device address of `scratch` on each device, looked up on first use
// Recorded behind every ptx_render_static frame of the device, whatever the context: the next static frame on that device
// -- any context, any stream -- waits for it before it touches the shared array, so back-to-back calls from one host
// thread on different contexts of one GPU are ordered on the device, not just on the host.  Created with the address
// (set-up time), so the render call itself still creates nothing.
hipEvent_t g_scratch_event[kMaxDevices] = {};

// `device` must be the current device.  Returns NULL (and the HIP error in `err`) if the symbol cannot be resolved.
void *static_scratch_for(int device, hipError_t &err)
{
    err = hipSuccess;
    if (device < 0 || device >= kMaxDevices) { err = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        err = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (err != hipSuccess || !p) { if (err == hipSuccess) err = hipErrorInvalidSymbol; return nullptr; }
        hipEvent_t ev = nullptr;
        err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (err != hipSuccess) return nullptr;
        g_scratch_event[device] = ev;
        g_scratch_ptr[device] = p;
    }
    return g_scratch_ptr[device];
}

}  // namespace

struct ptx_context {
    int device = 0;
    hipStream_t stream = nullptr;
    ptx_limits limits{};
    std::string err;

    // host mirrors of the reference's shared scene buffers (scene/mod.rs:48-54)
    std::vector<ptx_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<ptx_material> materials;
    std::vector<ptx_material_ext> material_ext;   // build extension: one row per material
    std::vector<float> normal_scale;              // glTF normalTexture.scale, one per material (ptx_set_material_normal_scale)
    uint32_t scene_options = 0;                   // ptx_scene_option bits
    std::vector<Geometry> geometries;
    std::vector<TexDesc> textures;          // device texel pointers
    uint32_t n_instances = 0;

    // device scene
    Bvh8Node *d_nodes = nullptr; uint32_t n_nodes = 0;
    TriRec *d_tris = nullptr; ShadeRec *d_shade = nullptr; uint64_t n_tris = 0;
    uint32_t *d_tri_of_rank = nullptr; uint64_t cap_rank = 0;   // soup index (= TriRec::rank) -> leaf-order index; k_extend's closest hit is kept by rank
    TexDesc *d_textures = nullptr;
    float *d_lut = nullptr;
    LitMaterial *d_lit_materials = nullptr;       // lit integrator
    uint32_t n_dev_lit_materials = 0;             // rows in d_lit_materials (the material count at the last build)
    LightRec *d_lights = nullptr; float *d_light_cdf = nullptr; uint32_t n_lights = 0;
    // texel sets of the lit integrator (ptx_kernels.h, LitMaterial): built on the first lit frame after a scene build,
    // kept across builds (a texture never changes once added), keyed by the three texture slots (index + 1, 0 = none)
    struct TexelSet { uint32_t base, normal, mr, flags, wh; uint32_t *d; uint64_t bytes; };
    std::vector<TexelSet> texel_sets;
    std::vector<LitMaterial> h_lit_materials;     // host copy of d_lit_materials (the set fields are filled in later)
    bool texel_sets_on = true;                    // PTX_OPTION_TEXEL_SETS
    bool texel_sets_stale = false;                // d_lit_materials does not carry the set fields the option asks for
    uint64_t texel_set_bytes = 0;
    uint32_t built_options = 0;                   // scene options the current device scene was built with
    void *staging = nullptr; uint64_t staging_bytes = 0;   // pinned host staging for scene uploads
    // device refit (ptx_refit.hip): what the kernels need to re-flatten and refit without the host
    RefitLeafSrc *d_leaf_src = nullptr; ptx_vertex *d_vertices = nullptr; uint64_t n_dev_vertices = 0;
    RefitInstance *d_refit_inst = nullptr; uint64_t cap_refit_inst = 0;
    void *d_raw_box = nullptr, *d_node_box = nullptr; RefitState *d_refit_state = nullptr;
    bool refit_ready = false;
    uint64_t cap_nodes = 0, cap_tris = 0;         // element capacities of d_nodes / d_tris + d_shade (reused across rebuilds)
    Bvh8 bvh;                                     // host copy of the acceleration structure (refit on instance motion)
    std::vector<ptx_instance> built_instances;    // the list it was built / last refitted for
    double built_sah = 0.0;                       // refit_bvh8's cost figure right after the build
    uint64_t n_builds = 0, n_refits = 0, n_device_refits = 0, n_tail_launches = 0;
    uint32_t bvh_depth = 0;
    bool scene_ready = false;
    bool textures_dirty = true;
    bool materials_dirty = true;                  // a material row or its extension changed since the last host-path build
    uint32_t n_dev_textures = 0;                  // descriptors in d_textures (textures added later wait for the next host-path build)

    // images (renderer.rs:476-531)
    uint32_t width = 0, height = 0;
    float4 *d_albedo = nullptr, *d_normal = nullptr; float *d_depth = nullptr; uint2 *d_ids = nullptr;

    // wavefront state
    uint64_t target_paths = 64u << 20;        // paths per chunk (PTX_OPTION_TARGET_PATHS): 64 Mi x 120 B = 7.7 GB of queue state per set in flight (lit: 176 B); 32 Mi measured 1.3 - 1.8 % slower, 128 Mi another 1 % faster on the lit workload only
    uint32_t extend_waves = 4096;           // persistent grid of the traversal kernel
    PathSet sets[kMaxSets];
    int n_sets = 2;                         // chunks in flight (PTX_FRAMES_IN_FLIGHT; measured: 2 gives the whole gain, 3 is slower than 2 and 4)
    int next_set = 0;
    // Queue length per bounce as a fraction of the chunk's paths, from the latest chunk whose counters have arrived: tells
    // from which bounce on the rest of a chunk is short enough for the fused tail launch.  A guess that only affects speed.
    bool have_frac = false, frac_lit = false;
    uint32_t frac_bounces = 0;
    float live_frac[kMaxDepthLimit] = {};
    uint32_t tail_shift = 7;                // ... or below 1/128 of the chunk's paths (PTX_TAIL_SHIFT): big batches hand over earlier
    uint32_t shade_grid = 0;                // most blocks of a shading launch (PTX_OPTION_SHADE_GRID); 0: the library default (8192)
    bool host_refit_only = false;           // PTX_OPTION_HOST_REFIT: instance motion never takes the device refit
    bool refit_verify = false;              // PTX_OPTION_REFIT_VERIFY: every device refit is checked against a host refit (tests)
    bool material_sort = false;             // PTX_MATERIAL_SORT=1: the lit integrator groups each bounce's queue by material class before shading (measured option)
    uint32_t tail_rays = 65536;             // PTX_TAIL_RAYS: expected queue length at or below which the tail takes over; 0: never
    hipEvent_t last_resolved = nullptr;     // = sets[k].resolved of the most recent chunk: the next resolve waits for it
    hipEvent_t last_first_shade = nullptr;  // = sets[k].first_shade of the most recent chunk: the next bounce-0 shading waits for it
    hipEvent_t static_done = nullptr;       // recorded on `stream` behind each ptx_render_static frame
    unsigned char *static_scratch = nullptr; // this device's copy of the file-scope scratch array (never freed)
    bool counted = false;                   // rccl_context_created has been called for this context
    uint64_t *d_totals = nullptr;           // rays, hits, samples, nodes fetched, triangles tested

    // profiling
    bool profiling = false;
    bool count_traversal = false;           // profiling level 2: instrumented extend kernel
    std::vector<EventPair> events;          // recorded, not yet read
    std::vector<EventPair> event_pool;      // free
    double extend_ms = 0, shade_ms = 0, total_ms = 0;
    uint64_t extend_launches = 0;
};

namespace {

int fail(ptx_context *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), PTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

int bind(ptx_context *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    return PTX_OK;
}

template <class T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

void free_paths(PathSet &ps)
{
    dfree(ps.d_rays[0]); dfree(ps.d_rays[1]); dfree(ps.d_rays[2]); dfree(ps.d_hits); dfree(ps.d_results); dfree(ps.d_order);
    ps.capacity = 0;
}

// the set's stream must be idle (its buffers may be replaced)
int ensure_paths(ptx_context *c, PathSet &ps, uint64_t n, bool shadow_queue)
{
    if (!ps.stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&ps.stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&ps.resolved, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&ps.first_shade, hipEventDisableTiming));
        HIP_TRY(c, hipMalloc((void **)&ps.d_counters, kNumCounters * sizeof(uint32_t)));
        HIP_TRY(c, hipEventCreateWithFlags(&ps.counts_ready, hipEventDisableTiming));
        HIP_TRY(c, hipHostMalloc((void **)&ps.h_counts, kMaxDepthLimit * sizeof(uint32_t), hipHostMallocDefault));
    }
    if (n > ps.capacity) {
        HIP_TRY(c, hipStreamSynchronize(ps.stream));
        free_paths(ps);
        HIP_TRY(c, hipMalloc(&ps.d_rays[0], n * kRayBytes));
        HIP_TRY(c, hipMalloc(&ps.d_rays[1], n * kRayBytes));
        HIP_TRY(c, hipMalloc((void **)&ps.d_hits, n * sizeof(HitOut)));
        HIP_TRY(c, hipMalloc((void **)&ps.d_results, n * sizeof(float4)));
        ps.capacity = n;
    }
    if (shadow_queue && !ps.d_rays[2]) HIP_TRY(c, hipMalloc(&ps.d_rays[2], ps.capacity * kRayBytes));
    if (shadow_queue && c->material_sort && !ps.d_order) HIP_TRY(c, hipMalloc((void **)&ps.d_order, ps.capacity * sizeof(uint32_t)));
    if (shadow_queue && c->material_sort && !ps.d_class_count) HIP_TRY(c, hipMalloc((void **)&ps.d_class_count, 8 * sizeof(uint32_t)));
    return PTX_OK;
}

// Host loops over triangles (flatten, pack) are embarrassingly parallel: static slices over up to 16 threads.
template <class F> void parallel_for(uint64_t n, F &&fn)
{
    unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 65536 || nt == 1) { fn((uint64_t)0, n); return; }
    std::vector<std::thread> th;
    uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        uint64_t lo = std::min<uint64_t>(n, t * per), hi = std::min<uint64_t>(n, lo + per);
        if (lo < hi) th.emplace_back([=, &fn] { fn(lo, hi); });
    }
    for (std::thread &t : th) t.join();
}

// page-locked staging buffer of the packed scene (TriRec + ShadeRec): uploads run at the link's rate
int ensure_staging(ptx_context *c, uint64_t bytes)
{
    if (bytes <= c->staging_bytes) return PTX_OK;
    if (c->staging) { (void)hipHostFree(c->staging); c->staging = nullptr; c->staging_bytes = 0; }
    HIP_TRY(c, hipHostMalloc(&c->staging, bytes, hipHostMallocDefault));
    c->staging_bytes = bytes;
    return PTX_OK;
}

// every stream of the context idle
int sync_all(ptx_context *c)
{
    for (int i = 0; i < kMaxSets; ++i)
        if (c->sets[i].stream) HIP_TRY(c, hipStreamSynchronize(c->sets[i].stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTX_OK;
}

// Build extension (PTX_SCENE_WORLD_NORMALS): cofactor matrix of the 3x3 part of the instance transform with
// the sign of its determinant folded in (= det x inverse transpose); same expression tree as the oracle's.
void normal_matrix(const float m[3][4], float c[9])
{
    const float *r0 = m[0], *r1 = m[1], *r2 = m[2];
    float c0[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    float c1[3] = {r2[1] * r0[2] - r2[2] * r0[1], r2[2] * r0[0] - r2[0] * r0[2], r2[0] * r0[1] - r2[1] * r0[0]};
    float c2[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    float det = (r0[0] * c0[0] + r0[1] * c0[1]) + r0[2] * c0[2];
    float sg = det < 0.0f ? -1.0f : 1.0f;
    for (int k = 0; k < 3; ++k) { c[k] = c0[k] * sg; c[3 + k] = c1[k] * sg; c[6 + k] = c2[k] * sg; }
}

void normal_to_world(const float c[9], const float n[3], float out[3])
{
    for (int r = 0; r < 3; ++r) out[r] = (c[r * 3] * n[0] + c[r * 3 + 1] * n[1]) + c[r * 3 + 2] * n[2];
}

void srgb_lut(float *lut)
{
    // IEC 61966-2-1 decode, as sampling an R8G8B8A8_SRGB image does (graphics.rs:494-495)
    for (int i = 0; i < 256; ++i) {
        double v = i / 255.0;
        double l = (v <= 0.04045) ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4);
        lut[i] = (float)l;
    }
}

// A timing pair from the pool, or a fresh one; `ok` = false when the events could not be created (the launch
// then simply goes untimed).
EventPair get_events(ptx_context *c, int kind, bool &ok)
{
    EventPair p{};
    ok = true;
    if (!c->event_pool.empty()) { p = c->event_pool.back(); c->event_pool.pop_back(); }
    else if (hipEventCreate(&p.a) != hipSuccess) { (void)hipGetLastError(); p.a = nullptr; ok = false; }
    else if (hipEventCreate(&p.b) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(p.a); p.a = p.b = nullptr; ok = false; }
    p.kind = kind;
    return p;
}

void account_pair(ptx_context *c, const EventPair &p)
{
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        if (p.kind == 0) { c->extend_ms += ms; c->extend_launches++; }
        else if (p.kind == 1) c->shade_ms += ms;
        else c->total_ms += ms;
    } else {
        (void)hipGetLastError();
    }
}

// every recorded pair; the streams must be idle
void drain_events(ptx_context *c)
{
    for (EventPair &p : c->events) { account_pair(c, p); c->event_pool.push_back(p); }
    c->events.clear();
}

// Profiling left on without ptx_get_stats: the pairs whose second event has completed are read and recycled, so
// the list stays bounded by the work in flight.
constexpr size_t kEventCap = 4096;
void drain_completed_events(ptx_context *c)
{
    if (c->events.size() < kEventCap) return;
    size_t keep = 0;
    for (size_t i = 0; i < c->events.size(); ++i) {
        EventPair &p = c->events[i];
        if (p.b && hipEventQuery(p.b) == hipSuccess) { account_pair(c, p); c->event_pool.push_back(p); }
        else { (void)hipGetLastError(); c->events[keep++] = p; }
    }
    c->events.resize(keep);
}

// One timed launch: begin() records the first event on `st`, end() the second.  If the events cannot be had the
// launch goes untimed; a pair that is begun and never ended (an early error return) goes back to the pool.
struct Timed {
    ptx_context *c = nullptr; EventPair p{}; bool on = false;
    Timed() = default;
    Timed(const Timed &) = delete;
    Timed &operator=(const Timed &) = delete;
    ~Timed() { if (on) c->event_pool.push_back(p); }
    void begin(ptx_context *ctx, int kind, hipStream_t st)
    {
        c = ctx;
        if (!c->profiling) return;
        bool ok = false;
        p = get_events(c, kind, ok);
        if (!ok) return;
        if (hipEventRecord(p.a, st) != hipSuccess) { (void)hipGetLastError(); c->event_pool.push_back(p); return; }
        on = true;
    }
    void end(hipStream_t st)
    {
        if (!on) return;
        on = false;
        if (hipEventRecord(p.b, st) != hipSuccess) { (void)hipGetLastError(); c->event_pool.push_back(p); return; }
        c->events.push_back(p);
    }
};

inline void xform_point(const float m[3][4], const float p[3], float out[3])
{
    for (int r = 0; r < 3; ++r) out[r] = ((m[r][0] * p[0] + m[r][1] * p[1]) + m[r][2] * p[2]) + m[r][3];
}

}  // namespace

namespace ptx {
uint64_t ctx_pixels(ptx_context *c) { return (uint64_t)c->width * c->height; }
int ctx_device(ptx_context *c) { return c->device; }
void *ctx_albedo(ptx_context *c) { return c->d_albedo; }
hipStream_t ctx_stream(ptx_context *c) { return c->stream; }
int ctx_fail(ptx_context *c, int code, const std::string &msg) { return fail(c, code, msg); }
}  // namespace ptx

extern "C" {

void ptx_default_limits(ptx_limits *l)
{
    l->max_vertices = 2500000; l->max_indices = 5000000; l->max_materials = 10000; l->max_offsets = 10000;
    l->max_textures = 5000; l->_pad = 0;
}

void ptx_default_params(ptx_render_params *p)
{
    memset(p, 0, sizeof *p);
    p->max_depth = 16; p->t_min = 0.001f; p->t_max = 10000.0f; p->normal_eps = 0.001f;
}

int ptx_create_with_limits(int device_ordinal, const ptx_limits *limits, ptx_context **out)
{
    if (!out) return fail(nullptr, PTX_ERR_INVALID, "ptx_create: ctx is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, PTX_ERR_HIP, std::string("ptx_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, PTX_ERR_INVALID, "ptx_create: device ordinal out of range");
    ptx_context *c = new (std::nothrow) ptx_context;
    if (!c) return fail(nullptr, PTX_ERR_INVALID, "ptx_create: out of host memory");
    c->device = device_ordinal;
    rccl_context_created(device_ordinal); c->counted = true;
    if (limits) c->limits = *limits; else ptx_default_limits(&c->limits);
    if (const char *tp = getenv("PTX_TARGET_PATHS")) {
        unsigned long long v = strtoull(tp, nullptr, 10);
        if (v >= 4096) c->target_paths = v;
    }
#define CREATE_TRY(expr)                                                                            \
    do {                                                                                            \
        hipError_t e2_ = (expr);                                                                    \
        if (e2_ != hipSuccess) {                                                                    \
            fail(nullptr, PTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e2_));         \
            ptx_destroy(c);                                                                         \
            return PTX_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)
    CREATE_TRY(hipSetDevice(device_ordinal));
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&c->static_done, hipEventDisableTiming));
    {
        hipError_t se = hipSuccess;
        c->static_scratch = (unsigned char *)static_scratch_for(device_ordinal, se);
        CREATE_TRY(se);
    }
    if (const char *ts = getenv("PTX_TAIL_SHIFT")) { long v = strtol(ts, nullptr, 10); if (v >= 0 && v <= 31) c->tail_shift = (uint32_t)v; }
    if (const char *tr = getenv("PTX_TAIL_RAYS")) { long v = strtol(tr, nullptr, 10); if (v >= 0) c->tail_rays = (uint32_t)std::min<long>(v, 1 << 24); }
    if (const char *ms = getenv("PTX_MATERIAL_SORT")) c->material_sort = atoi(ms) != 0;
    if (const char *fi = getenv("PTX_FRAMES_IN_FLIGHT")) { long v = strtol(fi, nullptr, 10); if (v >= 1 && v <= kMaxSets) c->n_sets = (int)v; }
    c->host_refit_only = getenv("PTX_HOST_REFIT") != nullptr;
    c->refit_verify = getenv("PTX_REFIT_VERIFY") != nullptr;
    if (const char *e = getenv("PTX_TEXEL_SETS")) c->texel_sets_on = atoi(e) != 0;
    CREATE_TRY(hipMalloc((void **)&c->d_totals, kNumTotals * sizeof(uint64_t)));
    CREATE_TRY(hipMemset(c->d_totals, 0, kNumTotals * sizeof(uint64_t)));
    CREATE_TRY(hipMalloc((void **)&c->d_lut, 256 * sizeof(float)));
    c->extend_waves = extend_resident_waves(device_ordinal);
    if (const char *ew = getenv("PTX_EXTEND_WAVES")) { long v = strtol(ew, nullptr, 10); if (v > 0) c->extend_waves = (uint32_t)v; }
    float lut[256]; srgb_lut(lut);
    CREATE_TRY(hipMemcpy(c->d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
#undef CREATE_TRY
    *out = c;
    return PTX_OK;
}

int ptx_create(int device_ordinal, ptx_context **out) { return ptx_create_with_limits(device_ordinal, nullptr, out); }

int ptx_destroy(ptx_context *c)
{
    if (!c) return PTX_OK;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    drain_events(c);
    for (EventPair &p : c->event_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (TexDesc &t : c->textures) { void *p = (void *)t.texels; if (p) (void)hipFree(p); }
    dfree(c->d_nodes); dfree(c->d_tris); dfree(c->d_shade); dfree(c->d_tri_of_rank); dfree(c->d_textures); dfree(c->d_lut);
    dfree(c->d_lit_materials); dfree(c->d_lights); dfree(c->d_light_cdf);
    for (auto &ts : c->texel_sets) if (ts.d) (void)hipFree(ts.d);
    if (c->staging) (void)hipHostFree(c->staging);
    dfree(c->d_leaf_src); dfree(c->d_vertices); dfree(c->d_refit_inst); dfree(c->d_raw_box); dfree(c->d_node_box); dfree(c->d_refit_state);
    dfree(c->d_albedo); dfree(c->d_normal); dfree(c->d_depth); dfree(c->d_ids);
    for (int i = 0; i < kMaxSets; ++i) {
        PathSet &ps = c->sets[i];
        free_paths(ps);
        dfree(ps.d_counters); dfree(ps.d_class_count);
        if (ps.resolved) (void)hipEventDestroy(ps.resolved);
        if (ps.first_shade) (void)hipEventDestroy(ps.first_shade);
        if (ps.counts_ready) (void)hipEventDestroy(ps.counts_ready);
        if (ps.h_counts) (void)hipHostFree(ps.h_counts);
        if (ps.stream) (void)hipStreamDestroy(ps.stream);
    }
    dfree(c->d_totals);
    if (c->static_done) (void)hipEventDestroy(c->static_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->counted) rccl_context_destroyed(c->device);
    delete c;
    return PTX_OK;
}

const char *ptx_last_error(ptx_context *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ptx_add_vertices(ptx_context *c, const ptx_vertex *v, uint64_t n, uint64_t *first)
{
    if (!c) return PTX_ERR_INVALID;
    if ((!v && n) || !first) return fail(c, PTX_ERR_INVALID, "ptx_add_vertices: NULL argument");
    if (c->vertices.size() + n > c->limits.max_vertices)
        return fail(c, PTX_ERR_CAPACITY, "ptx_add_vertices: vertex buffer overflow");   // scene/mod.rs:281-283
    *first = c->vertices.size();
    c->vertices.insert(c->vertices.end(), v, v + n);
    return PTX_OK;
}

int ptx_add_indices(ptx_context *c, const uint32_t *idx, uint64_t n, uint64_t *first)
{
    if (!c) return PTX_ERR_INVALID;
    if ((!idx && n) || !first) return fail(c, PTX_ERR_INVALID, "ptx_add_indices: NULL argument");
    if (c->indices.size() + n > c->limits.max_indices)
        return fail(c, PTX_ERR_CAPACITY, "ptx_add_indices: index buffer overflow");     // scene/mod.rs:412-414
    *first = c->indices.size();
    c->indices.insert(c->indices.end(), idx, idx + n);
    return PTX_OK;
}

int ptx_add_material(ptx_context *c, const ptx_material *m, uint32_t *index)
{
    if (!c) return PTX_ERR_INVALID;
    if (!m || !index) return fail(c, PTX_ERR_INVALID, "ptx_add_material: NULL argument");
    if (c->materials.size() + 1 > c->limits.max_materials)
        return fail(c, PTX_ERR_CAPACITY, "ptx_add_material: material buffer overflow"); // scene/mod.rs:347-349
    *index = (uint32_t)c->materials.size();
    c->materials.push_back(*m);
    ptx_material_ext e; memset(&e, 0, sizeof e); e.ior = 1.5f;
    c->material_ext.push_back(e);
    c->normal_scale.push_back(1.0f);
    c->materials_dirty = true;
    return PTX_OK;
}

int ptx_set_material_normal_scale(ptx_context *c, uint32_t index, float scale)
{
    if (!c) return PTX_ERR_INVALID;
    if (index >= c->materials.size()) return fail(c, PTX_ERR_INVALID, "ptx_set_material_normal_scale: material index out of range");
    if (!std::isfinite(scale)) return fail(c, PTX_ERR_INVALID, "ptx_set_material_normal_scale: scale must be finite");
    c->normal_scale[index] = scale;
    c->materials_dirty = true;
    return PTX_OK;
}

int ptx_set_material_ext(ptx_context *c, uint32_t index, const ptx_material_ext *ext)
{
    if (!c) return PTX_ERR_INVALID;
    if (!ext) return fail(c, PTX_ERR_INVALID, "ptx_set_material_ext: NULL argument");
    if (index >= c->materials.size()) return fail(c, PTX_ERR_INVALID, "ptx_set_material_ext: material index out of range");
    if (!(ext->ior > 0.0f) || !(ext->transmission >= 0.0f && ext->transmission <= 1.0f) ||
        !(ext->emissive[0] >= 0.0f && ext->emissive[1] >= 0.0f && ext->emissive[2] >= 0.0f))
        return fail(c, PTX_ERR_INVALID, "ptx_set_material_ext: ior must be > 0, transmission in [0,1], emission >= 0");
    if (ext->mr_texture > c->textures.size())
        return fail(c, PTX_ERR_INVALID, "ptx_set_material_ext: mr_texture names a texture that has not been added");
    if (ext->mr_texture && (c->textures[ext->mr_texture - 1].srgb & PTX_TEXTURE_SRGB))
        return fail(c, PTX_ERR_INVALID, "ptx_set_material_ext: mr_texture must be a linear (non-sRGB) texture");
    if (ext->normal_texture > c->textures.size() || ext->emissive_texture > c->textures.size())
        return fail(c, PTX_ERR_INVALID, "ptx_set_material_ext: normal_texture / emissive_texture names a texture that has not been added");
    if (ext->normal_texture && (c->textures[ext->normal_texture - 1].srgb & PTX_TEXTURE_SRGB))
        return fail(c, PTX_ERR_INVALID, "ptx_set_material_ext: normal_texture must be a linear (non-sRGB) texture");
    c->material_ext[index] = *ext;
    c->materials_dirty = true;
    return PTX_OK;
}

int ptx_set_scene_options(ptx_context *c, uint32_t options)
{
    if (!c) return PTX_ERR_INVALID;
    if (options & ~(uint32_t)(PTX_SCENE_WORLD_NORMALS | PTX_SCENE_TEXTURE_TIMES_FACTOR))
        return fail(c, PTX_ERR_INVALID, "ptx_set_scene_options: unknown option bit");
    c->scene_options = options;
    return PTX_OK;
}

int ptx_add_texture(ptx_context *c, const uint8_t *rgba8, uint32_t w, uint32_t h, int srgb, uint32_t *index)
{
    if (!c) return PTX_ERR_INVALID;
    if (!rgba8 || !index || w == 0 || h == 0) return fail(c, PTX_ERR_INVALID, "ptx_add_texture: bad argument");
    if (srgb & ~(PTX_TEXTURE_SRGB | PTX_TEXTURE_REPEAT | PTX_TEXTURE_LINEAR)) return fail(c, PTX_ERR_INVALID, "ptx_add_texture: unknown flag bit");
    if (c->textures.size() + 1 > c->limits.max_textures)
        return fail(c, PTX_ERR_CAPACITY, "ptx_add_texture: bindless texture pool size exceeded"); // renderer.rs:899-901
    if (int r = bind(c)) return r;
    // device layout: tiles of 8 x 4 texels, one cache line each (TexDesc)
    const uint32_t tiles_x = (w + kTexTileW - 1u) / kTexTileW, tiles_y = (h + kTexTileH - 1u) / kTexTileH;
    if ((uint64_t)tiles_x * tiles_y * 32ull > 0xFFFFFFFFull) return fail(c, PTX_ERR_INVALID, "ptx_add_texture: more than 2^32 texels");
    std::vector<uint32_t> tiled((size_t)tiles_x * tiles_y * 32u, 0u);
    for (uint32_t y = 0; y < h; ++y) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(rgba8) + (size_t)y * w;   // (memcpy below: rgba8 need not be 4-byte aligned)
        for (uint32_t x = 0; x < w; ++x) {
            uint32_t px;
            memcpy(&px, reinterpret_cast<const uint8_t *>(row) + (size_t)x * 4u, 4);
            tiled[((size_t)(y >> 2) * tiles_x + (x >> 3)) * 32u + ((y & 3u) << 3) + (x & 7u)] = px;
        }
    }
    void *d = nullptr;
    uint64_t bytes = (uint64_t)tiled.size() * 4;
    HIP_TRY(c, hipMalloc(&d, bytes));
    hipError_t e = hipMemcpy(d, tiled.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(c, PTX_ERR_HIP, std::string("ptx_add_texture: ") + hipGetErrorString(e)); }
    TexDesc t; t.texels = (const uint8_t *)d; t.width = w; t.height = h; t.tiles_x = tiles_x;
    t.srgb = (uint32_t)srgb & (uint32_t)(PTX_TEXTURE_SRGB | PTX_TEXTURE_REPEAT | PTX_TEXTURE_LINEAR);
    *index = (uint32_t)c->textures.size();
    c->textures.push_back(t);
    c->textures_dirty = true;
    return PTX_OK;
}

int ptx_create_geometry(ptx_context *c, uint64_t first_vertex, uint64_t n_vertices, uint64_t first_index,
                        uint64_t n_indices, uint32_t *geometry)
{
    if (!c) return PTX_ERR_INVALID;
    if (!geometry) return fail(c, PTX_ERR_INVALID, "ptx_create_geometry: NULL argument");
    if (first_vertex + n_vertices > c->vertices.size() || first_index + n_indices > c->indices.size())
        return fail(c, PTX_ERR_INVALID, "ptx_create_geometry: range outside the shared buffers");
    if (n_indices % 3 != 0) return fail(c, PTX_ERR_INVALID, "ptx_create_geometry: index count is not a multiple of 3");
    for (uint64_t i = 0; i < n_indices; ++i)
        if (c->indices[first_index + i] >= n_vertices)
            return fail(c, PTX_ERR_INVALID, "ptx_create_geometry: index refers past the geometry's vertices");
    *geometry = (uint32_t)c->geometries.size();
    c->geometries.push_back({first_vertex, n_vertices, first_index, n_indices});
    return PTX_OK;
}

static int set_instances_impl(ptx_context *c, const ptx_instance *inst, uint32_t n, bool try_refit);

int ptx_set_instances(ptx_context *c, const ptx_instance *inst, uint32_t n) { return set_instances_impl(c, inst, n, false); }
int ptx_update_instances(ptx_context *c, const ptx_instance *inst, uint32_t n) { return set_instances_impl(c, inst, n, true); }

}  // extern "C"

// A moved scene keeps its tree while the refitted tree's cost figure stays below this multiple of the built tree's.
static constexpr double kRefitSahLimit = 1.6;

// Instance motion on the device: new transforms up (96 B per instance), k_refit_pack re-flattens the leaf triangles in
// place, k_refit_nodes refits level by level from the deepest, the cost figure comes back (32 B).  `ok` = false: the
// moved scene needs the host path (a non-finite triangle, or the tree has become too loose) -- the device scene is
// then rebuilt from scratch by the caller.
static int device_refit(ptx_context *c, const ptx_instance *inst, uint32_t n, bool &ok)
{
    ok = false;
    std::vector<RefitInstance> ri(n);
    for (uint32_t i = 0; i < n; ++i) {
        memcpy(ri[i].xform, inst[i].xform, 48);
        normal_matrix(inst[i].xform, ri[i].cof);
        ri[i]._pad[0] = ri[i]._pad[1] = ri[i]._pad[2] = 0.0f;
    }
    if (n > c->cap_refit_inst) {
        dfree(c->d_refit_inst); c->cap_refit_inst = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_refit_inst, (size_t)n * sizeof(RefitInstance)));
        c->cap_refit_inst = n;
    }
    if (int r = sync_all(c)) return r;          // frames in flight read the triangles and nodes that are about to move
    // from here on the device scene is being moved in place: it is not renderable again until this refit has been
    // accepted (below) or the host path has rebuilt it (the caller's fallback); an error in between leaves it off
    c->scene_ready = false;
    RefitState st; refit_state_reset(&st);
    HIP_TRY(c, hipMemcpyAsync(c->d_refit_state, &st, sizeof st, hipMemcpyHostToDevice, c->stream));
    if (n) HIP_TRY(c, hipMemcpyAsync(c->d_refit_inst, ri.data(), (size_t)n * sizeof(RefitInstance), hipMemcpyHostToDevice, c->stream));
    launch_refit_pack(c->stream, c->d_leaf_src, c->d_vertices, c->d_refit_inst, (uint32_t)c->n_tris,
                      (c->built_options & PTX_SCENE_WORLD_NORMALS) != 0, c->d_tris, c->d_shade, c->d_raw_box, c->d_refit_state);
    const std::vector<uint32_t> &lf = c->bvh.level_first;
    for (size_t l = lf.size() - 1; l-- > 0;)
        launch_refit_nodes(c->stream, c->d_nodes, c->d_node_box, c->d_raw_box, lf[l], lf[l + 1] - lf[l], c->d_refit_state);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&st, c->d_refit_state, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sah = st.root_area > 0.0 ? st.area_sum / st.root_area : 0.0;
    if (st.non_finite || !(sah <= c->built_sah * kRefitSahLimit)) return PTX_OK;
    if (c->refit_verify) {
        // developer / test check: the host refit of the same motion must give byte-identical nodes and triangles
        uint64_t n_tris = 0;
        for (uint32_t i = 0; i < n; ++i) n_tris += c->geometries[inst[i].geometry].n_indices / 3;
        std::vector<float> wv(n_tris * 9);
        uint64_t k = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const Geometry &g = c->geometries[inst[i].geometry];
            for (uint64_t p = 0; p < g.n_indices / 3; ++p, ++k)
                for (int v = 0; v < 3; ++v)
                    xform_point(inst[i].xform, c->vertices[g.first_vertex + c->indices[g.first_index + p * 3 + v]].position, &wv[k * 9 + v * 3]);
        }
        Bvh8 host = c->bvh;
        double hsah = 0.0;
        if (!refit_bvh8(host, wv.data(), n_tris, &hsah, 4)) return fail(c, PTX_ERR_STATE, "PTX_REFIT_VERIFY: host refit refused what the device refit accepted");
        std::vector<Bvh8Node> dn(host.nodes.size());
        std::vector<TriRec> dt(c->n_tris);
        HIP_TRY(c, hipMemcpy(dn.data(), c->d_nodes, dn.size() * sizeof(Bvh8Node), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(dt.data(), c->d_tris, dt.size() * sizeof(TriRec), hipMemcpyDeviceToHost));
        if (memcmp(dn.data(), host.nodes.data(), dn.size() * sizeof(Bvh8Node)) != 0)
            return fail(c, PTX_ERR_STATE, "PTX_REFIT_VERIFY: device-refitted nodes differ from the host refit");
        for (uint64_t j = 0; j < c->n_tris; ++j)
            if (memcmp(dt[j].v0, &wv[(uint64_t)host.tri_order[j] * 9], 12) != 0 || memcmp(dt[j].v2, &wv[(uint64_t)host.tri_order[j] * 9 + 6], 12) != 0)
                return fail(c, PTX_ERR_STATE, "PTX_REFIT_VERIFY: device-flattened triangles differ from the host's");
        if (!(std::fabs(hsah - sah) <= 1e-9 * std::fabs(hsah))) return fail(c, PTX_ERR_STATE, "PTX_REFIT_VERIFY: cost figures differ");
    }
    c->built_instances.assign(inst, inst + n);
    c->n_refits++;
    c->n_device_refits++;
    c->scene_ready = true;
    ok = true;
    return PTX_OK;
}

static int set_instances_impl(ptx_context *c, const ptx_instance *inst, uint32_t n, bool try_refit)
{
    if (!c) return PTX_ERR_INVALID;
    if (!inst && n) return fail(c, PTX_ERR_INVALID, "ptx_set_instances: NULL argument");
    if (n > c->limits.max_offsets) return fail(c, PTX_ERR_CAPACITY, "ptx_set_instances: offsets buffer overflow");
    if (int r = bind(c)) return r;

    static const bool verbose = getenv("PTX_VERBOSE") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ptx] set_instances %-10s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    uint64_t n_tris = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const ptx_instance &in = inst[i];
        if (in.geometry >= c->geometries.size())
            return fail(c, PTX_ERR_INVALID, "ptx_set_instances: unknown geometry handle");
        const Geometry &g = c->geometries[in.geometry];
        // the Offsets row must address the geometry the instance names (renderer.rs:927-939)
        if (in.off.vertex_offset != g.first_vertex || in.off.index_offset != g.first_index ||
            in.off.index_count != g.n_indices || in.off.vertex_count != g.n_vertices)
            return fail(c, PTX_ERR_INVALID, "ptx_set_instances: offsets row does not match its geometry");
        if (in.off.material_offset >= c->materials.size())
            return fail(c, PTX_ERR_INVALID, "ptx_set_instances: material offset out of range");
        const ptx_material &m = c->materials[in.off.material_offset];
        if (m.has_base_texture && m.base_texture_indice >= c->textures.size())
            return fail(c, PTX_ERR_INVALID, "ptx_set_instances: material refers to a texture that was never added");
        n_tris += g.n_indices / 3;
    }
    if (n_tris > kMaxSceneTriangles)
        return fail(c, PTX_ERR_CAPACITY, "ptx_set_instances: more than 67 108 863 instanced triangles (the traversal kernel's 26-bit triangle index)");

    // Same entities, new transforms (the reference rebuilds its TLAS every frame, renderer.rs:961, and wants not
    // to: todo.md:3): keep the tree and refit its boxes, unless the moved scene has made it too loose.
    bool same_list = try_refit && c->scene_ready && c->built_instances.size() == n && c->built_options == c->scene_options;
    for (uint32_t i = 0; i < n && same_list; ++i)
        same_list = inst[i].geometry == c->built_instances[i].geometry && memcmp(&inst[i].off, &c->built_instances[i].off, sizeof(ptx_offsets)) == 0;
    // on the device when nothing else has to change (an emitter that moves takes the light table with it: host path)
    const bool host_refit_only = c->host_refit_only;
    bool emitter_moved = false;
    if (same_list && c->n_lights)
        for (uint32_t i = 0; i < n && !emitter_moved; ++i) {
            const ptx_material_ext &e = c->material_ext[inst[i].off.material_offset];
            if ((e.emissive[0] > 0.0f || e.emissive[1] > 0.0f || e.emissive[2] > 0.0f) &&
                memcmp(inst[i].xform, c->built_instances[i].xform, sizeof inst[i].xform) != 0) emitter_moved = true;
        }
    // ... and nothing else is waiting for the host path: material rows / extensions edited or textures added since the
    // last build reach the device (ShadeRec, LitMaterial, light table, texture descriptors) only there
    if (same_list && c->refit_ready && !emitter_moved && !host_refit_only && !c->materials_dirty && !c->textures_dirty &&
        c->bvh.level_first.size() >= 2) {
        bool ok = false;
        if (int r = device_refit(c, inst, n, ok)) return r;
        lap("dev refit");
        if (ok) return PTX_OK;
    }

    // flatten to world space: the instance transform is applied once here instead of per ray
    std::vector<float> wv(n_tris * 9);
    std::vector<uint32_t> tri_inst(n_tris), tri_prim(n_tris);
    {
        uint64_t k = 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t np = c->geometries[inst[i].geometry].n_indices / 3;
            for (uint64_t p = 0; p < np; ++p, ++k) { tri_inst[k] = i; tri_prim[k] = (uint32_t)p; }
        }
        parallel_for(n_tris, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t t = lo; t < hi; ++t) {
                const ptx_instance &in = inst[tri_inst[t]];
                const Geometry &g = c->geometries[in.geometry];
                for (int v = 0; v < 3; ++v) {
                    const ptx_vertex &vx = c->vertices[g.first_vertex + c->indices[g.first_index + (uint64_t)tri_prim[t] * 3 + v]];
                    xform_point(in.xform, vx.position, &wv[t * 9 + v * 3]);
                }
            }
        });
    }
    lap("flatten");
    bool refitted = false;
    if (same_list) {
        double sah = 0.0;
        int rthreads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (refit_bvh8(c->bvh, wv.data(), n_tris, &sah, rthreads) && sah <= c->built_sah * kRefitSahLimit) refitted = true;
    }
    Bvh8 &bvh = c->bvh;
    if (!refitted) {
        int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        c->scene_ready = false;
        if (!build_bvh8(wv.data(), n_tris, bvh, threads)) return fail(c, PTX_ERR_INVALID, "ptx_set_instances: BVH build failed");
        if (bvh.depth > kMaxBvhDepth) return fail(c, PTX_ERR_CAPACITY, "ptx_set_instances: BVH deeper than the traversal stack");
        c->built_sah = bvh.built_sah;         // the cost figure of the fresh tree, the one a refit of unmoved triangles reports
        c->n_builds++;
    } else {
        c->n_refits++;
    }
    c->built_instances.assign(inst, inst + n);
    lap(refitted ? "refit" : "build");

    const bool world_normals = (c->scene_options & PTX_SCENE_WORLD_NORMALS) != 0;
    std::vector<float> cof((size_t)n * 9);
    for (uint32_t i = 0; i < n; ++i) normal_matrix(inst[i].xform, &cof[(size_t)i * 9]);

    // Lit integrator: the light table = every triangle whose material emits, in (instance, primitive)
    // order, picked with probability proportional to area x luminance (oracle: build_lights).
    std::vector<LightRec> lights; std::vector<float> light_cdf;
    double light_total = 0.0;
    {
        std::vector<float> power, area;
        double total = 0.0;
        for (uint64_t t = 0; t < n_tris; ++t) {
            const ptx_material_ext &e = c->material_ext[inst[tri_inst[t]].off.material_offset];
            if (!(e.emissive[0] > 0.0f || e.emissive[1] > 0.0f || e.emissive[2] > 0.0f)) continue;
            const float *a = &wv[t * 9], *b = a + 3, *d = a + 6;
            float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {d[0] - a[0], d[1] - a[1], d[2] - a[2]};
            float cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            float len = std::sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
            float ar = 0.5f * len;
            if (!(ar > 0.0f) || !std::isfinite(ar)) continue;
            float lum = (0.2126f * e.emissive[0] + 0.7152f * e.emissive[1]) + 0.0722f * e.emissive[2];
            float pw = ar * lum;
            total += (double)pw;
            LightRec l;
            memcpy(l.a, a, 12); memcpy(l.b, b, 12); memcpy(l.c, d, 12);
            float il = 1.0f / len;
            l.n[0] = cr[0] * il; l.n[1] = cr[1] * il; l.n[2] = cr[2] * il;
            l.le_r = e.emissive[0]; l.le_g = e.emissive[1]; l.le_b = e.emissive[2];
            l.pdf_area = 0.0f;
            {
                const Geometry &g = c->geometries[inst[tri_inst[t]].geometry];
                const uint32_t *ix = &c->indices[g.first_index + (uint64_t)tri_prim[t] * 3];
                memcpy(l.uv0, c->vertices[g.first_vertex + ix[0]].uv, 8); memcpy(l.uv1, c->vertices[g.first_vertex + ix[1]].uv, 8);
                memcpy(l.uv2, c->vertices[g.first_vertex + ix[2]].uv, 8);
                l.etex = e.emissive_texture; l._pad = 0;
            }
            lights.push_back(l); power.push_back(pw); area.push_back(ar);
        }
        if (!(total > 0.0)) lights.clear();
        double run = 0.0;
        light_cdf.resize(lights.size());
        for (size_t i = 0; i < lights.size(); ++i) {
            run += (double)power[i];
            light_cdf[i] = (float)(run / total);
            lights[i].pdf_area = (float)((double)power[i] / total) / area[i];
        }
        if (!lights.empty()) light_cdf.back() = 1.0f;
        light_total = lights.empty() ? 0.0 : total;
    }
    std::vector<LitMaterial> lit_materials(c->materials.size());
    for (size_t i = 0; i < c->materials.size(); ++i) {
        LitMaterial &lm = lit_materials[i];
        memset(&lm, 0, sizeof lm);
        memcpy(lm.emissive, c->material_ext[i].emissive, 12);
        const float *em = c->material_ext[i].emissive;
        if (light_total > 0.0 && (em[0] > 0.0f || em[1] > 0.0f || em[2] > 0.0f)) {
            // the pick density seen from a BSDF-sampled ray that lands on an emitter (oracle: build_lights, mat_pdf_area)
            float lum = (0.2126f * em[0] + 0.7152f * em[1]) + 0.0722f * em[2];
            lm.pdf_area = (float)((double)lum / light_total);
        }
        lm.metallic = c->materials[i].metallic;
        lm.transmission = c->material_ext[i].transmission; lm.ior = c->material_ext[i].ior;
        lm.mr_texture = c->material_ext[i].mr_texture;
        lm.normal_texture = c->material_ext[i].normal_texture; lm.emissive_texture = c->material_ext[i].emissive_texture;
        lm.base_slot = c->materials[i].has_base_texture ? c->materials[i].base_texture_indice + 1u : 0u;
        lm.normal_scale = c->normal_scale[i];
    }

    lap("lights");
    uint64_t nt = bvh.tri_order.size();
    if (int r = ensure_staging(c, nt * (sizeof(TriRec) + sizeof(ShadeRec)) + 256)) return r;
    ShadeRec *shade = reinterpret_cast<ShadeRec *>(c->staging);                 // 128-byte records first (alignment)
    TriRec *tris = reinterpret_cast<TriRec *>(shade + nt);
    std::vector<RefitLeafSrc> leaf_src(refitted ? 0 : nt);                      // topology changed: the device refit's inputs too
    parallel_for(nt, [&](uint64_t klo, uint64_t khi) {
    for (uint64_t k = klo; k < khi; ++k) {
        uint32_t t = bvh.tri_order[k];
        const ptx_instance &in = inst[tri_inst[t]];
        const Geometry &g = c->geometries[in.geometry];
        const uint32_t *ix = &c->indices[g.first_index + (uint64_t)tri_prim[t] * 3];
        const ptx_vertex &a = c->vertices[g.first_vertex + ix[0]];
        const ptx_vertex &b = c->vertices[g.first_vertex + ix[1]];
        const ptx_vertex &d = c->vertices[g.first_vertex + ix[2]];
        if (!refitted) {
            RefitLeafSrc &ls = leaf_src[k];
            ls.instance = tri_inst[t];
            for (int v = 0; v < 3; ++v) ls.v[v] = (uint32_t)(g.first_vertex + ix[v]);
        }
        TriRec &tr = tris[k];
        memcpy(tr.v0, &wv[(uint64_t)t * 9], 12); memcpy(tr.v1, &wv[(uint64_t)t * 9 + 3], 12); memcpy(tr.v2, &wv[(uint64_t)t * 9 + 6], 12);
        tr.instance = tri_inst[t]; tr.prim = tri_prim[t]; tr.rank = t;     // the soup is laid out instance by instance, primitive by primitive
        ShadeRec &sr = shade[k];
        memset(&sr, 0, sizeof sr);
        memcpy(sr.v0, tr.v0, 12); memcpy(sr.v1, tr.v1, 12); memcpy(sr.v2, tr.v2, 12);
        sr.instance = tr.instance; sr.prim = tr.prim;
        const ptx_material &m = c->materials[in.off.material_offset];
        sr.texture = m.has_base_texture ? m.base_texture_indice : 0xFFFFFFFFu;
        sr.roughness = m.roughness; sr.base_r = m.base_color[0]; sr.base_g = m.base_color[1]; sr.base_b = m.base_color[2];
        sr.material = in.off.material_offset;
        if (world_normals) {
            const float *cf = &cof[(size_t)tri_inst[t] * 9];
            normal_to_world(cf, a.normal, sr.n0); normal_to_world(cf, b.normal, sr.n1); normal_to_world(cf, d.normal, sr.n2);
        } else {
            memcpy(sr.n0, a.normal, 12); memcpy(sr.n1, b.normal, 12); memcpy(sr.n2, d.normal, 12);   // object space, as the reference
        }
        memcpy(sr.uv0, a.uv, 8); memcpy(sr.uv1, b.uv, 8); memcpy(sr.uv2, d.uv, 8);
    }
    });

    lap("pack");
    if (int r = sync_all(c)) return r;
    lap("sync");
    dfree(c->d_lit_materials); dfree(c->d_lights); dfree(c->d_light_cdf);
    c->n_lights = 0; c->n_dev_lit_materials = 0;
    c->scene_ready = false;
    c->n_nodes = (uint32_t)bvh.nodes.size(); c->n_tris = nt; c->bvh_depth = bvh.depth;
    // device buffers are kept across rebuilds / refits while they are large enough
    if (bvh.nodes.size() > c->cap_nodes) {
        dfree(c->d_nodes); c->cap_nodes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_nodes, bvh.nodes.size() * sizeof(Bvh8Node)));
        c->cap_nodes = bvh.nodes.size();
    }
    if (nt > c->cap_tris) {
        dfree(c->d_tris); dfree(c->d_shade); c->cap_tris = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_tris, nt * sizeof(TriRec)));
        HIP_TRY(c, hipMalloc((void **)&c->d_shade, nt * sizeof(ShadeRec)));
        c->cap_tris = nt;
    }
    if (c->n_nodes) HIP_TRY(c, hipMemcpyAsync(c->d_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(Bvh8Node), hipMemcpyHostToDevice, c->stream));
    if (n_tris > c->cap_rank) {
        dfree(c->d_tri_of_rank); c->cap_rank = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_tri_of_rank, n_tris * sizeof(uint32_t)));
        c->cap_rank = n_tris;
    }
    std::vector<uint32_t> tri_of_rank(n_tris, 0xFFFFFFFFu);          // non-finite triangles are in no leaf: never hit
    for (uint64_t k = 0; k < nt; ++k) tri_of_rank[bvh.tri_order[k]] = (uint32_t)k;
    if (n_tris) HIP_TRY(c, hipMemcpyAsync(c->d_tri_of_rank, tri_of_rank.data(), n_tris * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    if (nt) {
        HIP_TRY(c, hipMemcpyAsync(c->d_tris, tris, nt * sizeof(TriRec), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_shade, shade, nt * sizeof(ShadeRec), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!lit_materials.empty()) {
        HIP_TRY(c, hipMalloc((void **)&c->d_lit_materials, lit_materials.size() * sizeof(LitMaterial)));
        HIP_TRY(c, hipMemcpy(c->d_lit_materials, lit_materials.data(), lit_materials.size() * sizeof(LitMaterial), hipMemcpyHostToDevice));
        c->n_dev_lit_materials = (uint32_t)lit_materials.size();
    }
    c->h_lit_materials = lit_materials;          // without set fields: ensure_texel_sets fills them in on the first lit frame
    c->texel_sets_stale = c->texel_sets_on;
    if (!lights.empty()) {
        HIP_TRY(c, hipMalloc((void **)&c->d_lights, lights.size() * sizeof(LightRec)));
        HIP_TRY(c, hipMemcpy(c->d_lights, lights.data(), lights.size() * sizeof(LightRec), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMalloc((void **)&c->d_light_cdf, light_cdf.size() * sizeof(float)));
        HIP_TRY(c, hipMemcpy(c->d_light_cdf, light_cdf.data(), light_cdf.size() * sizeof(float), hipMemcpyHostToDevice));
        c->n_lights = (uint32_t)lights.size();
    }
    c->built_options = c->scene_options;
    if (c->textures_dirty) {
        dfree(c->d_textures);
        c->n_dev_textures = 0;
        if (!c->textures.empty()) {
            HIP_TRY(c, hipMalloc((void **)&c->d_textures, c->textures.size() * sizeof(TexDesc)));
            HIP_TRY(c, hipMemcpy(c->d_textures, c->textures.data(), c->textures.size() * sizeof(TexDesc), hipMemcpyHostToDevice));
            c->n_dev_textures = (uint32_t)c->textures.size();
        }
        c->textures_dirty = false;
    }
    c->materials_dirty = false;
    // inputs of the device refit: per leaf triangle its instance and vertices, the shared vertex buffer, scratch boxes
    if (!refitted) {
        c->refit_ready = false;
        dfree(c->d_leaf_src); dfree(c->d_raw_box); dfree(c->d_node_box);
        if (nt && c->n_nodes) {
            HIP_TRY(c, hipMalloc((void **)&c->d_leaf_src, nt * sizeof(RefitLeafSrc)));
            HIP_TRY(c, hipMemcpy(c->d_leaf_src, leaf_src.data(), nt * sizeof(RefitLeafSrc), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMalloc(&c->d_raw_box, nt * (size_t)kRefitBoxBytes));
            HIP_TRY(c, hipMalloc(&c->d_node_box, (size_t)c->n_nodes * kRefitBoxBytes));
            if (!c->d_refit_state) HIP_TRY(c, hipMalloc((void **)&c->d_refit_state, sizeof(RefitState)));
            if (c->n_dev_vertices != c->vertices.size()) {      // the shared vertex buffer only ever grows (scene/mod.rs:270)
                dfree(c->d_vertices); c->n_dev_vertices = 0;
                HIP_TRY(c, hipMalloc((void **)&c->d_vertices, c->vertices.size() * sizeof(ptx_vertex)));
                HIP_TRY(c, hipMemcpy(c->d_vertices, c->vertices.data(), c->vertices.size() * sizeof(ptx_vertex), hipMemcpyHostToDevice));
                c->n_dev_vertices = c->vertices.size();
            }
            c->refit_ready = true;
        }
    }
    c->n_instances = n;
    c->scene_ready = true;
    lap("upload");
    return PTX_OK;
}


// Texel sets of the lit integrator (ptx_kernels.h, LitMaterial).  Called by the first lit frame after a scene build (the
// reference integrator never pays for them): every material whose base colour / normal / metallic-roughness textures --
// those it has -- share one size and one wrap / filter mode, and which fetches more than one plain nearest texel, gets
// its set (shared by materials with the same three textures, kept across builds), and the material rows go to the device
// again with the set fields filled in.  A set that would not fit the budget is skipped: that material keeps the plain taps.
static constexpr uint64_t kTexelSetBudget = 24ull << 30;      // bytes of texel sets per context
static int ensure_texel_sets(ptx_context *c)
{
    if (!c->texel_sets_stale) return PTX_OK;
    if (int r = sync_all(c)) return r;                        // frames in flight read the rows that are about to change
    std::vector<LitMaterial> rows = c->h_lit_materials;
    const size_t n = rows.size();
    for (size_t i = 0; c->texel_sets_on && i < n; ++i) {
        LitMaterial &lm = rows[i];
        const uint32_t slot[3] = {lm.base_slot, lm.normal_texture, lm.mr_texture};   // as of the build the device scene comes from
        uint32_t present = 0, w = 0, h = 0, mode = 0;
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            if (!slot[k]) continue;
            if (slot[k] - 1u >= c->n_dev_textures) { ok = false; break; }     // not on the device (or not a texture): the plain path's business
            const TexDesc &t = c->textures[slot[k] - 1u];
            const uint32_t md = t.srgb & (uint32_t)(PTX_TEXTURE_REPEAT | PTX_TEXTURE_LINEAR);
            if (!present) { w = t.width; h = t.height; mode = md; }
            else if (t.width != w || t.height != h || md != mode) ok = false;
            present |= 1u << k;
        }
        if (!ok || !present || w > 65533u || h > 65533u) continue;
        const bool linear = (mode & PTX_TEXTURE_LINEAR) != 0;
        if (!linear && present == 1u) continue;               // one nearest texel: nothing to gather
        uint32_t flags = kTexelSetOn | ((mode & PTX_TEXTURE_REPEAT) ? kTexelSetRepeat : 0u) | (linear ? kTexelSetLinear : 0u);
        if ((present & 6u) || !linear) flags |= kTexelSetHasB;
        if ((present & 1u) && (c->textures[slot[0] - 1u].srgb & PTX_TEXTURE_SRGB)) flags |= kTexelSetBaseSrgb;
        if ((present & 2u) && (c->textures[slot[1] - 1u].srgb & PTX_TEXTURE_SRGB)) flags |= kTexelSetNormalSrgb;
        if ((present & 4u) && (c->textures[slot[2] - 1u].srgb & PTX_TEXTURE_SRGB)) flags |= kTexelSetMrSrgb;
        const uint32_t wh = w | (h << 16);
        const ptx_context::TexelSet *found = nullptr;
        for (const auto &ts : c->texel_sets)
            if (ts.base == slot[0] && ts.normal == slot[1] && ts.mr == slot[2] && ts.flags == flags && ts.wh == wh) { found = &ts; break; }
        if (!found) {
            const uint64_t bytes = texel_set_bytes(w, h, flags);
            if (c->texel_set_bytes + bytes > kTexelSetBudget) continue;
            ptx_context::TexelSet ts{slot[0], slot[1], slot[2], flags, wh, nullptr, bytes};
            hipError_t e = hipMalloc((void **)&ts.d, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); continue; }   // no memory for it: the plain taps stay
            const TexDesc none{};
            launch_build_texel_set(c->stream, ts.d, (present & 1u) ? c->textures[slot[0] - 1u] : none, (present & 2u) ? c->textures[slot[1] - 1u] : none,
                                   (present & 4u) ? c->textures[slot[2] - 1u] : none, present, w, h, flags);
            c->texel_set_bytes += bytes;
            c->texel_sets.push_back(ts);
            found = &c->texel_sets.back();
        }
        lm.set_flags = found->flags; lm.set_wh = found->wh; lm.set_texels = found->d;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!rows.empty() && c->d_lit_materials && rows.size() == c->n_dev_lit_materials)
        HIP_TRY(c, hipMemcpy(c->d_lit_materials, rows.data(), rows.size() * sizeof(LitMaterial), hipMemcpyHostToDevice));
    c->texel_sets_stale = false;
    return PTX_OK;
}

extern "C" {

int ptx_resize(ptx_context *c, uint32_t w, uint32_t h)
{
    if (!c) return PTX_ERR_INVALID;
    if (w == 0 || h == 0 || (uint64_t)w * h > 0x40000000ull) return fail(c, PTX_ERR_INVALID, "ptx_resize: bad extent");
    if (int r = bind(c)) return r;
    if (int r = sync_all(c)) return r;
    dfree(c->d_albedo); dfree(c->d_normal); dfree(c->d_depth); dfree(c->d_ids);
    c->width = c->height = 0;
    uint64_t np = (uint64_t)w * h;
    HIP_TRY(c, hipMalloc((void **)&c->d_albedo, np * 16));
    HIP_TRY(c, hipMalloc((void **)&c->d_normal, np * 16));
    HIP_TRY(c, hipMalloc((void **)&c->d_depth, np * 4));
    HIP_TRY(c, hipMalloc((void **)&c->d_ids, np * 8));
    HIP_TRY(c, hipMemset(c->d_albedo, 0, np * 16));
    HIP_TRY(c, hipMemset(c->d_normal, 0, np * 16));
    HIP_TRY(c, hipMemset(c->d_depth, 0, np * 4));
    HIP_TRY(c, hipMemset(c->d_ids, 0xFF, np * 8));
    c->width = w; c->height = h;
    return PTX_OK;
}

int ptx_render_batch(ptx_context *c, const ptx_uniforms *u, const ptx_camera *cam, const ptx_render_params *pp,
                     uint32_t n_samples)
{
    if (!c) return PTX_ERR_INVALID;
    if (!u || !cam) return fail(c, PTX_ERR_INVALID, "ptx_render: NULL argument");
    if (!c->scene_ready) return fail(c, PTX_ERR_STATE, "ptx_render: ptx_set_instances has not been called");
    if (!c->width) return fail(c, PTX_ERR_STATE, "ptx_render: ptx_resize has not been called");
    ptx_render_params p;
    if (pp) p = *pp; else ptx_default_params(&p);
    if (p.max_depth > kMaxDepthLimit) return fail(c, PTX_ERR_INVALID, "ptx_render: max_depth above 64");
    // the traversal orders hits by the bit pattern of t and tests boxes in t_min-shifted time: both need 0 <= t_min < t_max
    if (!(p.t_min >= 0.0f) || !(p.t_max > p.t_min)) return fail(c, PTX_ERR_INVALID, "ptx_render: needs 0 <= t_min < t_max");
    if (p.integrator != PTX_INTEGRATOR_REFERENCE && p.integrator != PTX_INTEGRATOR_LIT)
        return fail(c, PTX_ERR_INVALID, "ptx_render: unknown integrator");
    const bool lit = p.integrator == PTX_INTEGRATOR_LIT;
    if (n_samples == 0) return PTX_OK;
    if (int r = bind(c)) return r;
    if (lit) if (int r = ensure_texel_sets(c)) return r;

    const uint64_t npix = (uint64_t)c->width * c->height;
    uint64_t chunk = std::max<uint64_t>(1, c->target_paths / npix);
    chunk = std::min<uint64_t>(chunk, n_samples);
    if (chunk * npix > 0xFFFFFF00ull) chunk = 0xFFFFFF00ull / npix;

    SceneDev sc;
    sc.lit_materials = c->d_lit_materials; sc.n_lit_materials = c->n_dev_lit_materials; sc.lights = c->d_lights; sc.light_cdf = c->d_light_cdf; sc.n_lights = c->n_lights;
    sc.nodes = c->d_nodes; sc.n_nodes = c->n_nodes; sc.tris = c->d_tris; sc.shade = c->d_shade; sc.tri_of_rank = c->d_tri_of_rank;
    sc.textures = c->d_textures; sc.n_textures = c->n_dev_textures; sc.srgb_lut = c->d_lut;
    ImagesDev img; img.albedo = c->d_albedo; img.normal = c->d_normal; img.depth = c->d_depth; img.ids = c->d_ids;
    FrameDev f;
    f.cam = *cam;
    {   // rgen.glsl:30 origin = view_inverse * vec4(0, 0, 0, 1): c0*x + c1*y + c2*z + c3*w summed left to right, as the oracle does
        const float *m = cam->view_inverse;
        for (int a = 0; a < 3; ++a) f.cam_origin[a] = ((m[a] * 0.0f + m[4 + a] * 0.0f) + m[8 + a] * 0.0f) + m[12 + a] * 1.0f;
    }
    f.width = c->width; f.height = c->height; f.n_pixels = (uint32_t)npix;
    f.seed = u->seed; f.rng_offset = p.rng_sample_offset;
    f.max_depth = p.max_depth; f.t_min = p.t_min; f.t_max = p.t_max; f.normal_eps = p.normal_eps;
    f.z_near = u->z_near; f.z_far = u->z_far;
    f.tex_times_factor = (c->built_options & PTX_SCENE_TEXTURE_TIMES_FACTOR) ? 1u : 0u;
    const uint32_t bounces = std::max(1u, p.max_depth);

    // Chunks overlap with the chunks before them (other path sets, other streams).  Two orders are kept by events:
    // the bounce-0 shading passes -- the only writers of the G-buffers and the id image -- run chunk after chunk, so
    // a frame with a moved camera or a restarted accumulation lands after its predecessors; and the resolves run
    // chunk after chunk, so every pixel receives its samples in accumulated_count order.

    for (uint32_t s0 = 0; s0 < n_samples; s0 += (uint32_t)chunk) {
        uint32_t ns = (uint32_t)std::min<uint64_t>(chunk, n_samples - s0);
        uint32_t n_paths = (uint32_t)(ns * npix);
        // profiling level 2 keeps every chunk on ONE path set (one stream): the HIP-event time of a stage launch is
        // then that kernel's own time, not its time plus the wait behind the other set's persistent traversal kernel
        PathSet &ps = c->sets[c->count_traversal ? 0 : c->next_set];
        c->next_set = c->count_traversal ? 0 : (c->next_set + 1) % c->n_sets;
        if (int r = ensure_paths(c, ps, chunk * npix, lit)) return r;
        hipStream_t st = ps.stream;
        // queue lengths of finished chunks (any set)
        for (int si = 0; si < c->n_sets; ++si) {
            PathSet &q = c->sets[si];
            if (!q.counts_pending) continue;
            if (hipEventQuery(q.counts_ready) != hipSuccess) { (void)hipGetLastError(); continue; }   // "not ready" is no error: do not leave it for the check at the end
            q.counts_pending = false;
            c->have_frac = q.counts_paths > 0; c->frac_lit = q.counts_lit; c->frac_bounces = q.counts_bounces;
            for (uint32_t b = 0; b < q.counts_bounces && q.counts_paths; ++b)
                c->live_frac[b] = (float)q.h_counts[b] / (float)q.counts_paths;
        }
        // the bounce from which the fused tail launch takes the chunk to its end (reference integrator, uninstrumented runs)
        uint32_t tail_at = bounces;
        if (c->tail_rays && !c->count_traversal && !c->material_sort && c->have_frac && c->frac_lit == lit && c->frac_bounces == bounces)
            for (uint32_t b = 1; b < bounces; ++b)
                if (1.5 * (double)c->live_frac[b] * (double)n_paths <= (double)std::max<uint64_t>(c->tail_rays, n_paths >> c->tail_shift)) { tail_at = b; break; }
        // Frame-sized chunks (<= 4 Mi paths) with another chunk in flight: half the resident waves each, so that the two
        // traversal kernels share the chip instead of queueing behind each other's persistent grid (1080p frame
        // 8.16 -> 7.74 ms; 16-spp batches want the full grid: 327 vs 302 Msamples/s)
        const uint32_t waves = (c->n_sets >= 2 && n_paths <= (4u << 20)) ? std::max(256u, c->extend_waves / 2u) : c->extend_waves;
        drain_completed_events(c);
        Timed whole;
        whole.begin(c, 2, st);
        f.mean_base = u->accumulated_count + s0;
        f.ids_sample = (s0 == 0) ? 0u : 0xFFFFFFFFu;
        launch_raygen(st, f, n_paths, make_ray_queue(ps.d_rays[0], ps.capacity), ps.d_counters);
        for (uint32_t b = 0; b < bounces; ++b) {
            RayQueue rin = make_ray_queue(ps.d_rays[b & 1], ps.capacity), rout = make_ray_queue(ps.d_rays[(b & 1) ^ 1], ps.capacity);
            RayQueue shadow_q = make_ray_queue(lit ? ps.d_rays[2] : ps.d_rays[0], ps.capacity);
            uint64_t *trav = c->count_traversal ? c->d_totals + 3 : nullptr;
            if (b == tail_at) {      // (its time shows in total_ms only: extend_ms / shade_ms keep meaning the two kernels)
                if (lit) launch_tail_lit(st, sc, f, b, rin, ps.d_counters, n_paths, ps.d_results);
                else launch_tail(st, sc, f, b, rin, ps.d_counters, n_paths, ps.d_results);
                c->n_tail_launches++;
                break;
            }
            Timed e, e2;
            e.begin(c, 0, st);
            launch_extend(st, sc, f, rin, ps.d_counters + b, ps.d_counters + fetch_heads(b, 0), n_paths, waves, ps.d_hits, trav, false, b == 0);
            e.end(st);
            if (b == 0 && c->last_first_shade && c->last_first_shade != ps.first_shade)
                HIP_TRY(c, hipStreamWaitEvent(st, c->last_first_shade, 0));
            e2.begin(c, 1, st);
            if (!lit) {
                launch_shade(st, sc, f, img, b, rin, ps.d_hits, ps.d_counters, n_paths, rout, ps.d_results, c->shade_grid);
            } else {
                const uint32_t *order = nullptr;
                if (c->material_sort) {
                    launch_material_sort(st, sc, ps.d_hits, ps.d_counters + b, n_paths, ps.d_class_count, ps.d_order, c->shade_grid);
                    order = ps.d_order;
                }
                launch_shade_lit(st, sc, f, img, b, rin, ps.d_hits, ps.d_counters, n_paths, rout, shadow_q, ps.d_results, order, c->shade_grid);
            }
            if (b == 0) { HIP_TRY(c, hipEventRecord(ps.first_shade, st)); c->last_first_shade = ps.first_shade; }
            e2.end(st);
            if (lit && c->n_lights) {
                // occlusion rays of this bounce: any-hit traversal, then the unoccluded contributions join the radiance
                Timed e3;
                e3.begin(c, 0, st);
                launch_extend(st, sc, f, shadow_q, ps.d_counters + kShadowCount + b, ps.d_counters + fetch_heads(b, 1), n_paths,
                              waves, ps.d_hits, nullptr, true);
                e3.end(st);
                launch_shadow_resolve(st, b, shadow_q, ps.d_hits, ps.d_counters, n_paths, ps.d_results, c->shade_grid);
            }
        }
        // samples join the running mean in accumulated_count order: this chunk's resolve after the previous chunk's
        if (c->last_resolved && c->last_resolved != ps.resolved) HIP_TRY(c, hipStreamWaitEvent(st, c->last_resolved, 0));
        launch_resolve(st, f, img, ps.d_results, ns);
        launch_accumulate_stats(st, ps.d_counters, bounces, (uint64_t)n_paths, c->d_totals);
        HIP_TRY(c, hipEventRecord(ps.resolved, st));
        c->last_resolved = ps.resolved;
        if (c->tail_rays) {
            HIP_TRY(c, hipMemcpyAsync(ps.h_counts, ps.d_counters, bounces * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipEventRecord(ps.counts_ready, st));
            ps.counts_pending = true; ps.counts_paths = n_paths; ps.counts_bounces = bounces; ps.counts_lit = lit;
        }
        whole.end(st);
    }
    HIP_TRY(c, hipGetLastError());
    return PTX_OK;
}

int ptx_render(ptx_context *c, const ptx_uniforms *u, const ptx_camera *cam, const ptx_render_params *p)
{
    return ptx_render_batch(c, u, cam, p, 1);
}

// One frame (1 sample per pixel) like ptx_render, with the wavefront state in the file-scope scratch array of the
// context's device: no hipMalloc / hipFree, no event or stream creation, no host allocation.  Runs on the context's
// main stream, bounce by bounce (no frames in flight, no fused tail, no profiling events), ordered after every
// ptx_render / ptx_render_batch chunk issued before it and before every one issued after it.
int ptx_render_static(ptx_context *c, const ptx_uniforms *u, const ptx_camera *cam, const ptx_render_params *pp)
{
    if (!c) return PTX_ERR_INVALID;
    if (!u || !cam) return fail(c, PTX_ERR_INVALID, "ptx_render_static: NULL argument");
    if (!c->scene_ready) return fail(c, PTX_ERR_STATE, "ptx_render_static: ptx_set_instances has not been called");
    if (!c->width) return fail(c, PTX_ERR_STATE, "ptx_render_static: ptx_resize has not been called");
    ptx_render_params p;
    if (pp) p = *pp; else ptx_default_params(&p);
    if (p.max_depth > kMaxDepthLimit) return fail(c, PTX_ERR_INVALID, "ptx_render_static: max_depth above 64");
    if (!(p.t_min >= 0.0f) || !(p.t_max > p.t_min)) return fail(c, PTX_ERR_INVALID, "ptx_render_static: needs 0 <= t_min < t_max");
    if (p.integrator != PTX_INTEGRATOR_REFERENCE)
        return fail(c, PTX_ERR_INVALID, "ptx_render_static: only PTX_INTEGRATOR_REFERENCE (the lit integrator's occlusion queue is not part of the static scratch)");
    const uint64_t npix = (uint64_t)c->width * c->height;
    if (npix > kStaticMaxPaths)
        return fail(c, PTX_ERR_CAPACITY, "ptx_render_static: image has more than 262144 pixels (512 x 512), the size the static scratch holds; use ptx_render");
    if (!c->static_scratch) return fail(c, PTX_ERR_STATE, "ptx_render_static: the static scratch of this device was not resolved");
    if (int r = bind(c)) return r;

    unsigned char *base = c->static_scratch;
    uint32_t *counters = reinterpret_cast<uint32_t *>(base);
    void *rays[2] = {base + kStaticRayOff, base + kStaticRayOff + kStaticMaxPaths * kRayBytes};
    HitOut *hits = reinterpret_cast<HitOut *>(base + kStaticHitOff);
    float4 *results = reinterpret_cast<float4 *>(base + kStaticResultOff);
    const uint32_t n_paths = (uint32_t)npix;

    SceneDev sc;
    sc.lit_materials = c->d_lit_materials; sc.n_lit_materials = c->n_dev_lit_materials; sc.lights = c->d_lights; sc.light_cdf = c->d_light_cdf; sc.n_lights = c->n_lights;
    sc.nodes = c->d_nodes; sc.n_nodes = c->n_nodes; sc.tris = c->d_tris; sc.shade = c->d_shade; sc.tri_of_rank = c->d_tri_of_rank;
    sc.textures = c->d_textures; sc.n_textures = c->n_dev_textures; sc.srgb_lut = c->d_lut;
    ImagesDev img; img.albedo = c->d_albedo; img.normal = c->d_normal; img.depth = c->d_depth; img.ids = c->d_ids;
    FrameDev f;
    f.cam = *cam;
    {
        const float *m = cam->view_inverse;     // rgen.glsl:30, as ptx_render_batch
        for (int a = 0; a < 3; ++a) f.cam_origin[a] = ((m[a] * 0.0f + m[4 + a] * 0.0f) + m[8 + a] * 0.0f) + m[12 + a] * 1.0f;
    }
    f.width = c->width; f.height = c->height; f.n_pixels = n_paths;
    f.seed = u->seed; f.rng_offset = p.rng_sample_offset;
    f.max_depth = p.max_depth; f.t_min = p.t_min; f.t_max = p.t_max; f.normal_eps = p.normal_eps;
    f.z_near = u->z_near; f.z_far = u->z_far;
    f.tex_times_factor = (c->built_options & PTX_SCENE_TEXTURE_TIMES_FACTOR) ? 1u : 0u;
    f.mean_base = u->accumulated_count;
    f.ids_sample = 0u;
    const uint32_t bounces = std::max(1u, p.max_depth);

    hipStream_t st = c->stream;
    // behind the chunks already in flight on the path sets: their G-buffer writes and their resolves
    if (c->last_first_shade && c->last_first_shade != c->static_done) HIP_TRY(c, hipStreamWaitEvent(st, c->last_first_shade, 0));
    if (c->last_resolved && c->last_resolved != c->static_done) HIP_TRY(c, hipStreamWaitEvent(st, c->last_resolved, 0));
    // behind the latest static frame of ANY context on this device: they all share the one scratch array
    hipEvent_t dev_ev = (c->device >= 0 && c->device < kMaxDevices) ? g_scratch_event[c->device] : nullptr;
    if (dev_ev) HIP_TRY(c, hipStreamWaitEvent(st, dev_ev, 0));
    // k_raygen's first block clears every counter and fetch head the frame uses: nothing is kept from the last call
    launch_raygen(st, f, n_paths, make_ray_queue(rays[0], kStaticMaxPaths), counters);
    for (uint32_t b = 0; b < bounces; ++b) {
        RayQueue rin = make_ray_queue(rays[b & 1], kStaticMaxPaths), rout = make_ray_queue(rays[(b & 1) ^ 1], kStaticMaxPaths);
        launch_extend(st, sc, f, rin, counters + b, counters + fetch_heads(b, 0), n_paths, c->extend_waves, hits, nullptr, false, b == 0);
        launch_shade(st, sc, f, img, b, rin, hits, counters, n_paths, rout, results, c->shade_grid);
    }
    launch_resolve(st, f, img, results, 1);
    launch_accumulate_stats(st, counters, bounces, (uint64_t)n_paths, c->d_totals);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->static_done, st));
    if (dev_ev) HIP_TRY(c, hipEventRecord(dev_ev, st));
    c->last_resolved = c->static_done;
    c->last_first_shade = c->static_done;
    return PTX_OK;
}

int ptx_read_image(ptx_context *c, int kind, void *dst, uint64_t bytes)
{
    if (!c) return PTX_ERR_INVALID;
    if (!dst) return fail(c, PTX_ERR_INVALID, "ptx_read_image: NULL destination");
    if (!c->width) return fail(c, PTX_ERR_STATE, "ptx_read_image: no image (ptx_resize not called)");
    uint64_t np = (uint64_t)c->width * c->height;
    const void *src = nullptr; uint64_t want = 0;
    switch (kind) {
    case PTX_IMAGE_ALBEDO: src = c->d_albedo; want = np * 16; break;
    case PTX_IMAGE_NORMAL: src = c->d_normal; want = np * 16; break;
    case PTX_IMAGE_DEPTH: src = c->d_depth; want = np * 4; break;
    case PTX_IMAGE_PRIMARY_ID: src = c->d_ids; want = np * 8; break;
    default: return fail(c, PTX_ERR_INVALID, "ptx_read_image: unknown image kind");
    }
    if (bytes != want) return fail(c, PTX_ERR_INVALID, "ptx_read_image: byte count does not match the image");
    if (int r = bind(c)) return r;
    if (int r = sync_all(c)) return r;
    HIP_TRY(c, hipMemcpyAsync(dst, src, want, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTX_OK;
}

int ptx_write_albedo(ptx_context *c, const void *src, uint64_t bytes)
{
    if (!c) return PTX_ERR_INVALID;
    if (!src) return fail(c, PTX_ERR_INVALID, "ptx_write_albedo: NULL source");
    if (!c->width) return fail(c, PTX_ERR_STATE, "ptx_write_albedo: no image");
    if (bytes != (uint64_t)c->width * c->height * 16) return fail(c, PTX_ERR_INVALID, "ptx_write_albedo: byte count does not match the image");
    if (int r = bind(c)) return r;
    if (int r = sync_all(c)) return r;
    HIP_TRY(c, hipMemcpyAsync(c->d_albedo, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTX_OK;
}

int ptx_export_albedo_device(ptx_context *c, void *device_dst, uint64_t bytes, float scale)
{
    if (!c) return PTX_ERR_INVALID;
    if (!device_dst) return fail(c, PTX_ERR_INVALID, "ptx_export_albedo_device: NULL destination");
    if (!c->width) return fail(c, PTX_ERR_STATE, "ptx_export_albedo_device: no image");
    uint64_t np = (uint64_t)c->width * c->height;
    if (bytes != np * 16) return fail(c, PTX_ERR_INVALID, "ptx_export_albedo_device: byte count does not match the image");
    if (int r = bind(c)) return r;
    if (int r = sync_all(c)) return r;
    launch_scale_copy(c->stream, c->d_albedo, (float4 *)device_dst, np, scale);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTX_OK;
}

int ptx_import_albedo_device(ptx_context *c, const void *device_src, uint64_t bytes, float scale)
{
    if (!c) return PTX_ERR_INVALID;
    if (!device_src) return fail(c, PTX_ERR_INVALID, "ptx_import_albedo_device: NULL source");
    if (!c->width) return fail(c, PTX_ERR_STATE, "ptx_import_albedo_device: no image");
    uint64_t np = (uint64_t)c->width * c->height;
    if (bytes != np * 16) return fail(c, PTX_ERR_INVALID, "ptx_import_albedo_device: byte count does not match the image");
    if (int r = bind(c)) return r;
    if (int r = sync_all(c)) return r;
    launch_scale_copy(c->stream, (const float4 *)device_src, c->d_albedo, np, scale);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTX_OK;
}

// Build extension: per-context tuning knobs (each has an environment variable of the same meaning that only sets the
// DEFAULT of contexts created afterwards).  None of them changes a result bit: they steer scheduling only.
int ptx_set_option(ptx_context *c, int key, uint64_t value)
{
    if (!c) return PTX_ERR_INVALID;
    if (int r = bind(c)) return r;
    switch (key) {
    case PTX_OPTION_FRAMES_IN_FLIGHT:
        if (value < 1 || value > (uint64_t)kMaxSets) return fail(c, PTX_ERR_INVALID, "ptx_set_option: frames in flight must be 1 .. 4");
        if (int r = sync_all(c)) return r;          // chunks in flight keep their sets; the rotation restarts on set 0
        c->n_sets = (int)value; c->next_set = 0;
        return PTX_OK;
    case PTX_OPTION_TAIL_RAYS:
        c->tail_rays = (uint32_t)std::min<uint64_t>(value, 1u << 24);
        return PTX_OK;
    case PTX_OPTION_TAIL_SHIFT:
        if (value > 31) return fail(c, PTX_ERR_INVALID, "ptx_set_option: tail shift must be 0 .. 31");
        c->tail_shift = (uint32_t)value;
        return PTX_OK;
    case PTX_OPTION_TARGET_PATHS:
        if (value < 4096 || value > 0xFFFFFF00ull) return fail(c, PTX_ERR_INVALID, "ptx_set_option: paths per chunk must be 4096 .. 2^32 - 256");
        c->target_paths = value;
        return PTX_OK;
    case PTX_OPTION_MATERIAL_SORT:
        if (value > 1) return fail(c, PTX_ERR_INVALID, "ptx_set_option: material sort is 0 or 1");
        if (int r = sync_all(c)) return r;
        c->material_sort = value != 0;
        return PTX_OK;
    case PTX_OPTION_EXTEND_WAVES:
        if (value > (1u << 20)) return fail(c, PTX_ERR_INVALID, "ptx_set_option: traversal grid above 2^20 waves");
        c->extend_waves = value ? (uint32_t)value : extend_resident_waves(c->device);
        return PTX_OK;
    case PTX_OPTION_SHADE_GRID:
        if (value > (1u << 24)) return fail(c, PTX_ERR_INVALID, "ptx_set_option: shading grid above 2^24 blocks");
        c->shade_grid = (uint32_t)value;
        return PTX_OK;
    case PTX_OPTION_HOST_REFIT: c->host_refit_only = value != 0; return PTX_OK;
    case PTX_OPTION_REFIT_VERIFY: c->refit_verify = value != 0; return PTX_OK;
    case PTX_OPTION_TEXEL_SETS:
        if (value > 1) return fail(c, PTX_ERR_INVALID, "ptx_set_option: texel sets are 0 or 1");
        if (c->texel_sets_on != (value != 0)) { c->texel_sets_on = value != 0; c->texel_sets_stale = true; }   // the next lit frame re-writes the material rows
        return PTX_OK;
    case PTX_OPTION_TEXEL_SET_BYTES: return fail(c, PTX_ERR_INVALID, "ptx_set_option: the texel set size is read-only");
    default: return fail(c, PTX_ERR_INVALID, "ptx_set_option: unknown option key");
    }
}

int ptx_get_option(ptx_context *c, int key, uint64_t *value)
{
    if (!c) return PTX_ERR_INVALID;
    if (!value) return fail(c, PTX_ERR_INVALID, "ptx_get_option: NULL argument");
    switch (key) {
    case PTX_OPTION_FRAMES_IN_FLIGHT: *value = (uint64_t)c->n_sets; return PTX_OK;
    case PTX_OPTION_TAIL_RAYS: *value = c->tail_rays; return PTX_OK;
    case PTX_OPTION_TAIL_SHIFT: *value = c->tail_shift; return PTX_OK;
    case PTX_OPTION_TARGET_PATHS: *value = c->target_paths; return PTX_OK;
    case PTX_OPTION_MATERIAL_SORT: *value = c->material_sort ? 1 : 0; return PTX_OK;
    case PTX_OPTION_EXTEND_WAVES: *value = c->extend_waves; return PTX_OK;
    case PTX_OPTION_SHADE_GRID: *value = c->shade_grid ? c->shade_grid : default_shade_grid(); return PTX_OK;
    case PTX_OPTION_HOST_REFIT: *value = c->host_refit_only ? 1 : 0; return PTX_OK;
    case PTX_OPTION_REFIT_VERIFY: *value = c->refit_verify ? 1 : 0; return PTX_OK;
    case PTX_OPTION_TEXEL_SETS: *value = c->texel_sets_on ? 1 : 0; return PTX_OK;
    case PTX_OPTION_TEXEL_SET_BYTES: *value = c->texel_set_bytes; return PTX_OK;
    default: return fail(c, PTX_ERR_INVALID, "ptx_get_option: unknown option key");
    }
}

int ptx_synchronize(ptx_context *c)
{
    if (!c) return PTX_ERR_INVALID;
    if (int r = bind(c)) return r;
    return sync_all(c);
}

int ptx_set_profiling(ptx_context *c, int enabled)
{
    if (!c) return PTX_ERR_INVALID;
    c->profiling = enabled != 0;
    c->count_traversal = enabled >= 2;
    return PTX_OK;
}

int ptx_get_stats(ptx_context *c, ptx_stats *s)
{
    if (!c) return PTX_ERR_INVALID;
    if (!s) return fail(c, PTX_ERR_INVALID, "ptx_get_stats: NULL argument");
    if (int r = bind(c)) return r;
    if (int r = sync_all(c)) return r;
    drain_events(c);
    uint64_t t[kNumTotals];
    HIP_TRY(c, hipMemcpy(t, c->d_totals, sizeof t, hipMemcpyDeviceToHost));
    memset(s, 0, sizeof *s);
    s->rays = t[0]; s->hits = t[1]; s->samples = t[2]; s->nodes_fetched = t[3]; s->tris_tested = t[4];
    for (int i = 0; i < 6; ++i) s->wave_sched[i] = t[5 + i];
    s->extend_launches = c->extend_launches; s->extend_ms = c->extend_ms; s->shade_ms = c->shade_ms; s->total_ms = c->total_ms;
    s->bvh_nodes = c->n_nodes; s->bvh_triangles = c->n_tris;
    s->bvh_bytes = (uint64_t)c->n_nodes * sizeof(Bvh8Node) + c->n_tris * (sizeof(TriRec) + sizeof(ShadeRec));
    s->bvh_depth = c->bvh_depth;
    s->shadow_rays = t[11]; s->lights = c->n_lights;
    for (int i = 0; i < 8; ++i) s->extend_phase_cycles[i] = t[12 + i];
    for (int i = 0; i < 3; ++i) s->extend_append[i] = t[20 + i];
    s->bvh_builds = c->n_builds; s->bvh_refits = c->n_refits; s->bvh_device_refits = c->n_device_refits;
    s->tail_launches = c->n_tail_launches;
    return PTX_OK;
}

int ptx_reset_stats(ptx_context *c)
{
    if (!c) return PTX_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (int r = sync_all(c)) return r;
    drain_events(c);
    HIP_TRY(c, hipMemset(c->d_totals, 0, kNumTotals * sizeof(uint64_t)));
    c->extend_ms = c->shade_ms = c->total_ms = 0; c->extend_launches = 0;
    c->n_tail_launches = 0;
    return PTX_OK;
}

}  // extern "C"
