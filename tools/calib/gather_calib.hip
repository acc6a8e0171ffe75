// gather_calib.hip -- how fast can MI355X serve the traversal kernel's access pattern, VALU aside?  Every lane walks a
// dependent chain of random record fetches out of a table that lives in L2 / Infinity Cache (the BVH of the bench
// scene is 2 MB of nodes + 12.6 MB of triangles): record i -> a few integer ops -> next index.  Record sizes are the
// node formats under discussion (80 B = 5 x dwordx4 today, 96 / 112 / 128 B) and the 48-B triangle record.
// Reports records per second and bytes per clock per CU at 4 and 8 waves per SIMD.
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/calib/gather_calib.hip -o tools/calib/gather_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kQuads>      // record = kQuads x 16 bytes
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ table, uint32_t n_records, int steps, uint32_t *out)
{
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_records;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint4 *p = table + (size_t)idx * kQuads;
        uint4 q[kQuads];
#pragma unroll
        for (int k = 0; k < kQuads; ++k) q[k] = p[k];
        uint32_t h = 0;
#pragma unroll
        for (int k = 0; k < kQuads; ++k) h ^= q[k].x + q[k].y + q[k].z + q[k].w;
        acc += h;
        idx = (h * 2654435761u + (uint32_t)s) % n_records;      // dependent: the next record is known only now
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int kQuads>
static void run(const char *what, const uint4 *d_table, size_t table_bytes, uint32_t *d_out, int cus, bool &first)
{
    const uint32_t n_records = (uint32_t)(table_bytes / (16 * kQuads));
    const int steps = 2000;
    for (int waves_per_simd : {4, 8}) {
        const int blocks = cus * waves_per_simd;          // 256 threads = one wave per SIMD of a CU
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_gather<kQuads>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, 200, d_out);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather<kQuads>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, d_out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double recs = (double)blocks * 256 * steps;
        const double rate = recs / (ms * 1e-3);
        printf("%s  {\"pattern\": \"%s\", \"record_bytes\": %d, \"table_mb\": %.1f, \"waves_per_simd\": %d, \"ms\": %.3f, "
               "\"records_per_s\": %.4g, \"tb_per_s\": %.3f, \"bytes_per_clk_per_cu_at_2p4ghz\": %.2f}",
               first ? "" : ",\n", what, 16 * kQuads, table_bytes / 1e6, waves_per_simd, ms, rate, rate * 16 * kQuads / 1e12,
               rate * 16 * kQuads / cus / 2.4e9);
        first = false;
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t max_bytes = 64u << 20;
    std::vector<uint32_t> h(max_bytes / 4);
    uint32_t x = 12345;
    for (uint32_t &v : h) { x = x * 1664525u + 1013904223u; v = x; }
    uint4 *d_table; uint32_t *d_out;
    CHECK(hipMalloc((void **)&d_table, max_bytes));
    CHECK(hipMemcpy(d_table, h.data(), max_bytes, hipMemcpyHostToDevice));
    CHECK(hipMalloc((void **)&d_out, (size_t)cus * 8 * 256 * 4));
    printf("{\"device\": \"%s\", \"cus\": %d, \"results\": [\n", prop.gcnArchName, cus);
    bool first = true;
    // 2 MB: the bench scene's node array (fits the 4 MB L2 of every XCD); 16 MB: nodes + triangles (Infinity Cache, partly L2)
    for (size_t bytes : {(size_t)15360, (size_t)245760, (size_t)2 << 20, (size_t)16 << 20}) {
        run<5>("node 80 B", d_table, bytes, d_out, cus, first);
        run<6>("node 96 B", d_table, bytes, d_out, cus, first);
        run<7>("node 112 B", d_table, bytes, d_out, cus, first);
        run<8>("node 128 B", d_table, bytes, d_out, cus, first);
        run<3>("triangle 48 B", d_table, bytes, d_out, cus, first);
    }
    printf("\n]}\n");
    return 0;
}
