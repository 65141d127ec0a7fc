// Probe: how fast can a kernel gather 2 KiB rows out of pinned host memory over PCIe on MI355X?
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_probe.hip -o /tmp/pcie_probe && /tmp/pcie_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one 2 KiB row = 128 lanes x 16 B.  ROWS_PER_WG rows per 256-thread workgroup, all loads issued before any store.
template <int MODE, int ROWS_PER_WG>
__global__ __launch_bounds__(256) void gather(const u32x4* __restrict__ host, u32x4* __restrict__ dev, const int* __restrict__ ids, int nrows) {
    const int tid = threadIdx.x, unit = tid & 127, rsub = tid >> 7;
    u32x4 v[ROWS_PER_WG / 2];
#pragma unroll
    for (int k = 0; k < ROWS_PER_WG / 2; ++k) {
        int i = blockIdx.x * ROWS_PER_WG + k * 2 + rsub;
        if (i < nrows) {
            const u32x4* p = host + (size_t)ids[i] * 128 + unit;
            if (MODE == 0) v[k] = *p;
            else if (MODE == 1) v[k] = __builtin_nontemporal_load(p);
            else { // sc1 sc0 (system-scope) load
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[k]) : "v"(p) : "memory");
            }
        }
    }
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < ROWS_PER_WG / 2; ++k) {
        int i = blockIdx.x * ROWS_PER_WG + k * 2 + rsub;
        if (i < nrows) dev[(size_t)i * 128 + unit] = v[k];
    }
}

template <int MODE, int RPW>
float run(const u32x4* host, u32x4* dev, const int* ids, int nrows, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int grid = (nrows + RPW - 1) / RPW;
    hipLaunchKernelGGL((gather<MODE, RPW>), dim3(grid), dim3(256), 0, 0, host, dev, ids, nrows);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gather<MODE, RPW>), dim3(grid), dim3(256), 0, 0, host, dev, ids, nrows);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const size_t table_rows = 1 << 20;  // 2 GiB host table
    u32x4* host; CK(hipHostMalloc((void**)&host, table_rows * 2048, hipHostMallocDefault));
    for (size_t i = 0; i < table_rows * 128; i += 997) host[i] = (u32x4){(uint32_t)i, 1, 2, 3};
    for (int nrows : {680, 2048, 8192, 65536}) {
        std::vector<int> h(nrows); srand(7);
        for (int i = 0; i < nrows; ++i) h[i] = (int)(((size_t)rand() * 7919 + i) % table_rows);
        int* ids; CK(hipMalloc(&ids, nrows * 4)); CK(hipMemcpy(ids, h.data(), nrows * 4, hipMemcpyHostToDevice));
        u32x4* dev; CK(hipMalloc(&dev, (size_t)nrows * 2048));
        double mb = nrows * 2048.0 / 1e6;
        int it = nrows >= 65536 ? 5 : 50;
        float t;
        t = run<0, 8>(host, dev, ids, nrows, it);  printf("rows %6d (%.2f MB)  plain  8 rows/WG : %8.1f us  %6.1f GB/s\n", nrows, mb, t * 1e3, mb / t);
        t = run<0, 2>(host, dev, ids, nrows, it);  printf("rows %6d (%.2f MB)  plain  2 rows/WG : %8.1f us  %6.1f GB/s\n", nrows, mb, t * 1e3, mb / t);
        t = run<0, 16>(host, dev, ids, nrows, it); printf("rows %6d (%.2f MB)  plain 16 rows/WG : %8.1f us  %6.1f GB/s\n", nrows, mb, t * 1e3, mb / t);
        t = run<1, 8>(host, dev, ids, nrows, it);  printf("rows %6d (%.2f MB)  nt     8 rows/WG : %8.1f us  %6.1f GB/s\n", nrows, mb, t * 1e3, mb / t);
        t = run<2, 8>(host, dev, ids, nrows, it);  printf("rows %6d (%.2f MB)  sc0sc1 8 rows/WG : %8.1f us  %6.1f GB/s\n", nrows, mb, t * 1e3, mb / t);
        CK(hipFree(ids)); CK(hipFree(dev));
    }
    // DMA ceiling
    void* d; size_t n = 256u << 20; CK(hipMalloc(&d, n));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemcpyAsync(d, host, n, hipMemcpyHostToDevice, 0)); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) CK(hipMemcpyAsync(d, host, n, hipMemcpyHostToDevice, 0));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemcpyAsync H2D 256 MiB: %.1f GB/s\n", n / 1e6 / (ms / 5));
    // small DMA copies: 2 KiB x 680 individually enqueued
    CK(hipEventRecord(a)); for (int i = 0; i < 680; ++i) CK(hipMemcpyAsync((char*)d + i * 2048, (char*)host + (size_t)i * 7 * 2048, 2048, hipMemcpyHostToDevice, 0));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("680 x hipMemcpyAsync(2 KiB): %.1f us total\n", ms * 1e3);
    return 0;
}
