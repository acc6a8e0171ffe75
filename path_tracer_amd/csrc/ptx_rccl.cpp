// ptx_rccl.cpp -- ptx_reduce_accumulation: sum the RGBA32F accumulation buffers of one context
// per GPU into the root's with an RCCL reduce over xGMI.  The reference is
// single-GPU and has no counterpart.  librccl is resolved at call time so that single-GPU hosts
// never load it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

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

}  // namespace

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
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return ptx::ctx_fail(rc, PTX_ERR_INVALID, "ptx_reduce_accumulation: two contexts on one device");
    }
    if (n == 1) return ptx_synchronize(rc);

    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return ptx::ctx_fail(rc, PTX_ERR_HIP, std::string("ptx_reduce_accumulation: cannot load librccl: ") + dlerror());
    fn_CommInitAll comm_init_all = (fn_CommInitAll)dlsym(lib, "ncclCommInitAll");
    fn_CommDestroy comm_destroy = (fn_CommDestroy)dlsym(lib, "ncclCommDestroy");
    fn_GroupStart group_start = (fn_GroupStart)dlsym(lib, "ncclGroupStart");
    fn_GroupEnd group_end = (fn_GroupEnd)dlsym(lib, "ncclGroupEnd");
    fn_Reduce reduce = (fn_Reduce)dlsym(lib, "ncclReduce");
    fn_GetErrorString err_str = (fn_GetErrorString)dlsym(lib, "ncclGetErrorString");
    if (!comm_init_all || !comm_destroy || !group_start || !group_end || !reduce)
        return ptx::ctx_fail(rc, PTX_ERR_HIP, "ptx_reduce_accumulation: librccl lacks a required symbol");

    for (int i = 0; i < n; ++i)
        if (int r = ptx_synchronize(ctxs[i])) return r;
    std::vector<rcclComm> comms(n, nullptr);
    int e = comm_init_all(comms.data(), n, devs.data());
    if (e) return ptx::ctx_fail(rc, PTX_ERR_HIP, std::string("ncclCommInitAll: ") + (err_str ? err_str(e) : "error"));
    int status = PTX_OK;
    e = group_start();
    for (int i = 0; i < n && !e; ++i) {
        (void)hipSetDevice(devs[i]);
        void *buf = ptx::ctx_albedo(ctxs[i]);
        e = reduce(buf, buf, np * 4, kRcclFloat32, kRcclSum, root, comms[i], ptx::ctx_stream(ctxs[i]));
    }
    int e2 = group_end();
    if (e || e2) status = ptx::ctx_fail(rc, PTX_ERR_HIP, std::string("ncclReduce: ") + (err_str ? err_str(e ? e : e2) : "error"));
    for (int i = 0; i < n; ++i) (void)ptx_synchronize(ctxs[i]);
    for (int i = 0; i < n; ++i) if (comms[i]) comm_destroy(comms[i]);
    if (status) return status;
    // mean of the per-GPU means (each GPU rendered the same number of samples); alpha back to 1
    return ptx_import_albedo_device(rc, ptx::ctx_albedo(rc), np * 16, 1.0f / (float)n);
}
