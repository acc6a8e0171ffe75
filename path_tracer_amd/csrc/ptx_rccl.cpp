// ptx_rccl.cpp -- ptx_reduce_accumulation: sum the RGBA32F accumulation buffers of one context
// per GPU into the root's with an RCCL reduce over xGMI.  The reference is
// single-GPU and has no counterpart.  librccl is resolved on the first multi-context call, so that
// single-GPU hosts never load it; the library handle and the communicators (one set per device
// list) are then kept for the life of the contexts.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdlib>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "ptx_context_access.h"

namespace {

typedef void *rcclComm;
typedef int (*fn_CommInitAll)(rcclComm *, int, const int *);
typedef int (*fn_CommDestroy)(rcclComm);
typedef int (*fn_GroupStart)();
typedef int (*fn_GroupEnd)();
typedef int (*fn_Reduce)(const void *, void *, size_t, int, int, int, rcclComm, hipStream_t);
typedef const char *(*fn_GetErrorString)(int);

constexpr int kRcclFloat32 = 7;   // ncclFloat32
constexpr int kRcclSum = 0;       // ncclSum

// librccl is loaded once per process; communicators are created once per device set and kept until the last
// context on one of the set's devices is destroyed (ncclCommInitAll costs hundreds of milliseconds on 8 GPUs).
struct RcclApi {
    void *lib = nullptr;
    fn_CommInitAll comm_init_all = nullptr;
    fn_CommDestroy comm_destroy = nullptr;
    fn_GroupStart group_start = nullptr;
    fn_GroupEnd group_end = nullptr;
    fn_Reduce reduce = nullptr;
    fn_GetErrorString err_str = nullptr;
    bool test_stub = false;         // the loaded library is tests/native/rccl_stub.cpp (exports ptx_rccl_test_stub)
    std::string load_error;
};
struct CommSet {
    std::vector<int> devs;          // in rank order
    std::vector<rcclComm> comms;
};
std::mutex g_mutex;                 // guards everything below
RcclApi g_api;
std::vector<CommSet> g_sets;
std::map<int, int> g_live_contexts; // device -> contexts alive on it

bool load_api()
{
    if (g_api.lib) return true;
    // PTX_RCCL_LIB: path of the collective library to load instead of the loader's librccl.so (a site build of RCCL; the
    // call-recording stub of the test suite)
    void *lib = nullptr;
    if (const char *path = getenv("PTX_RCCL_LIB")) lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { const char *e = dlerror(); g_api.load_error = std::string("cannot load librccl: ") + (e ? e : "?"); return false; }
    RcclApi a;
    a.comm_init_all = (fn_CommInitAll)dlsym(lib, "ncclCommInitAll");
    a.comm_destroy = (fn_CommDestroy)dlsym(lib, "ncclCommDestroy");
    a.group_start = (fn_GroupStart)dlsym(lib, "ncclGroupStart");
    a.group_end = (fn_GroupEnd)dlsym(lib, "ncclGroupEnd");
    a.reduce = (fn_Reduce)dlsym(lib, "ncclReduce");
    a.err_str = (fn_GetErrorString)dlsym(lib, "ncclGetErrorString");
    if (!a.comm_init_all || !a.comm_destroy || !a.group_start || !a.group_end || !a.reduce) {
        g_api.load_error = "librccl lacks a required symbol";
        dlclose(lib);
        return false;
    }
    a.test_stub = dlsym(lib, "ptx_rccl_test_stub") != nullptr;
    a.lib = lib;
    g_api = a;
    return true;
}

void destroy_set(CommSet &cs)
{
    for (rcclComm c : cs.comms) if (c) g_api.comm_destroy(c);
    cs.comms.clear();
}

}  // namespace

namespace ptx {

void rccl_context_created(int device)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    g_live_contexts[device]++;
}

// ptx_destroy: when the last context of a device goes, the communicators that span that device go with it
void rccl_context_destroyed(int device)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_live_contexts.find(device);
    if (it == g_live_contexts.end() || --it->second > 0) return;
    g_live_contexts.erase(it);
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (size_t i = 0; i < g_sets.size();) {
        bool spans = false;
        for (int d : g_sets[i].devs) spans = spans || d == device;
        if (spans) { destroy_set(g_sets[i]); g_sets.erase(g_sets.begin() + (long)i); }
        else ++i;
    }
    if (have_cur) (void)hipSetDevice(cur);
}

}  // namespace ptx

extern "C" int ptx_reduce_accumulation(ptx_context **ctxs, int n, int root)
{
    if (!ctxs || n <= 0 || root < 0 || root >= n || !ctxs[root]) return PTX_ERR_INVALID;
    ptx_context *rc = ctxs[root];
    uint64_t np = ptx::ctx_pixels(rc);
    if (!np) return ptx::ctx_fail(rc, PTX_ERR_STATE, "ptx_reduce_accumulation: no image");
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return ptx::ctx_fail(rc, PTX_ERR_INVALID, "ptx_reduce_accumulation: NULL context");
        if (ptx::ctx_pixels(ctxs[i]) != np) return ptx::ctx_fail(rc, PTX_ERR_INVALID, "ptx_reduce_accumulation: image extents differ");
        devs[i] = ptx::ctx_device(ctxs[i]);
    }
    bool shared_device = false;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) shared_device = shared_device || devs[j] == devs[i];
    // one context: nothing to combine -- unless PTX_RCCL_SINGLE_RANK is set (tests): then the whole sequence below runs with a
    // one-rank communicator, which is how a one-GPU box exercises the REAL librccl through this file's dlsym'd prototypes
    if (n == 1 && !getenv("PTX_RCCL_SINGLE_RANK")) return ptx_synchronize(rc);

    // every return path below leaves the caller's current device as it was (ptx_synchronize binds each context's)
    int caller_device = 0;
    const bool have_caller_device = hipGetDevice(&caller_device) == hipSuccess;
    auto leave = [&](int code) { if (have_caller_device) (void)hipSetDevice(caller_device); return code; };
    for (int i = 0; i < n; ++i)
        if (int r = ptx_synchronize(ctxs[i])) return leave(r);
    int status = PTX_OK;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!load_api()) return leave(ptx::ctx_fail(rc, PTX_ERR_HIP, "ptx_reduce_accumulation: " + g_api.load_error));
        // RCCL wants one rank per device; only the test suite's host-side stub can sum two buffers of one GPU
        if (shared_device && !g_api.test_stub)
            return leave(ptx::ctx_fail(rc, PTX_ERR_INVALID, "ptx_reduce_accumulation: two contexts on one device"));
        CommSet *cs = nullptr;
        for (CommSet &c : g_sets) if (c.devs == devs) cs = &c;
        if (!cs) {
            CommSet fresh;
            fresh.devs = devs;
            fresh.comms.assign(n, nullptr);
            int e = g_api.comm_init_all(fresh.comms.data(), n, devs.data());
            if (e) {
                destroy_set(fresh);
                return leave(ptx::ctx_fail(rc, PTX_ERR_HIP, std::string("ncclCommInitAll: ") + (g_api.err_str ? g_api.err_str(e) : "error")));
            }
            g_sets.push_back(fresh);
            cs = &g_sets.back();
        }
        int e = g_api.group_start();
        for (int i = 0; i < n && !e; ++i) {
            (void)hipSetDevice(devs[i]);
            void *buf = ptx::ctx_albedo(ctxs[i]);
            e = g_api.reduce(buf, buf, np * 4, kRcclFloat32, kRcclSum, root, cs->comms[i], ptx::ctx_stream(ctxs[i]));
        }
        int e2 = g_api.group_end();
        if (e || e2) status = ptx::ctx_fail(rc, PTX_ERR_HIP, std::string("ncclReduce: ") + (g_api.err_str ? g_api.err_str(e ? e : e2) : "error"));
    }
    for (int i = 0; i < n; ++i) (void)ptx_synchronize(ctxs[i]);
    // mean of the per-GPU means (each GPU rendered the same number of samples); alpha back to 1
    if (!status) status = ptx_import_albedo_device(rc, ptx::ctx_albedo(rc), np * 16, 1.0f / (float)n);
    return leave(status);
}
