// rccl_stub.cpp -- TEST INFRASTRUCTURE: a call-recording stand-in for librccl, loaded by ptx_reduce_accumulation through
// PTX_RCCL_LIB (tests/test_gpu_rccl_stub.py).  A builder's box has ONE GPU, where RCCL refuses two ranks on a device, so
// the communicator cache of path_tracer_amd/csrc/ptx_rccl.cpp (one ncclCommInitAll per device list, reuse, destruction
// with the last context of a device, re-initialisation) and the order of its calls would otherwise never execute before
// the driver's 8-GPU run.  The stub exports the six entry points the library resolves with dlsym, sums the buffers on
// the host in rank order (fp32, like ncclSum on floats) and counts every call.  The marker symbol ptx_rccl_test_stub
// tells ptx_reduce_accumulation that two contexts may share a device here.  It also leaves the current device
// switched to the last rank's on purpose: the caller-device restore is the library's job.
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <vector>

namespace {
struct Comm { int rank, n_ranks, device; };
struct Call { const void *send; void *recv; size_t count; int root; Comm *comm; hipStream_t stream; };
std::vector<Call> g_pending;
bool g_in_group = false;
int g_counts[6] = {0, 0, 0, 0, 0, 0};   // CommInitAll, CommDestroy, GroupStart, GroupEnd, Reduce, live communicators
}  // namespace

extern "C" {

int ptx_rccl_test_stub = 1;

int ncclCommInitAll(void **comms, int n, const int *devs)
{
    g_counts[0]++;
    for (int i = 0; i < n; ++i) { comms[i] = new Comm{i, n, devs ? devs[i] : i}; g_counts[5]++; }
    return 0;
}

int ncclCommDestroy(void *c)
{
    g_counts[1]++; g_counts[5]--;
    delete static_cast<Comm *>(c);
    return 0;
}

int ncclGroupStart() { g_counts[2]++; g_in_group = true; g_pending.clear(); return 0; }

int ncclReduce(const void *send, void *recv, size_t count, int dtype, int op, int root, void *comm, hipStream_t stream)
{
    g_counts[4]++;
    if (dtype != 7 /* ncclFloat32 */ || op != 0 /* ncclSum */ || !comm || !g_in_group) return 4;   // ncclInvalidArgument
    g_pending.push_back(Call{send, recv, count, root, static_cast<Comm *>(comm), stream});
    return 0;
}

int ncclGroupEnd()
{
    g_counts[3]++;
    g_in_group = false;
    if (g_pending.empty()) return 0;
    const size_t count = g_pending[0].count;
    const int n = g_pending[0].comm->n_ranks, root = g_pending[0].root;
    if ((int)g_pending.size() != n) return 4;
    std::vector<std::vector<float>> part((size_t)n, std::vector<float>(count));
    void *root_recv = nullptr; int root_dev = 0;
    for (const Call &c : g_pending) {
        if (c.count != count || c.root != root || c.comm->rank < 0 || c.comm->rank >= n) return 4;
        if (hipSetDevice(c.comm->device) != hipSuccess) return 1;
        if (hipStreamSynchronize(c.stream) != hipSuccess) return 1;
        if (hipMemcpy(part[(size_t)c.comm->rank].data(), c.send, count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (c.comm->rank == root) { root_recv = c.recv; root_dev = c.comm->device; }
    }
    if (!root_recv) return 4;
    std::vector<float> sum(part[0]);
    for (int r = 1; r < n; ++r)
        for (size_t i = 0; i < count; ++i) sum[i] = sum[i] + part[(size_t)r][i];
    if (hipSetDevice(root_dev) != hipSuccess) return 1;
    if (hipMemcpy(root_recv, sum.data(), count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
    (void)hipSetDevice(g_pending.back().comm->device);      // leaves the device switched, like a real group call may
    g_pending.clear();
    return 0;
}

const char *ncclGetErrorString(int e) { return e == 4 ? "stub: invalid argument" : (e ? "stub: HIP error" : "no error"); }

void ptx_rccl_stub_counts(int out[6]) { for (int i = 0; i < 6; ++i) out[i] = g_counts[i]; }

}  // extern "C"
