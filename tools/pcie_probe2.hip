// Probe: does the SIZE of the pinned host table (8 GB at bs 1, 197 GB at bs 24) or its page size change the rate at which a
// kernel gathers random 2 KiB rows over PCIe?  Same gather as tools/pcie_probe.hip (8 rows per 256-thread workgroup, all
// loads of a workgroup issued before its stores), 16,384 rows (33.5 MB: about the misses of one bs-24 layer) per launch.
//   tables: hipHostMalloc (what the cache uses), and anonymous memory with MADV_HUGEPAGE registered with hipHostRegister
//   (2 MiB pages where transparent huge pages are granted) - the GPU reads both through its own page tables.
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_probe2.hip -o tools/bin/pcie_probe2 -lpthread
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <algorithm>
#include <thread>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ROWS_PER_WG>
__global__ __launch_bounds__(256) void gather(const u32x4* __restrict__ host, u32x4* __restrict__ dev, const long long* __restrict__ ids, int nrows) {
    const int tid = threadIdx.x, unit = tid & 127, rsub = tid >> 7;
    u32x4 v[ROWS_PER_WG / 2];
    long long src[ROWS_PER_WG / 2];
#pragma unroll
    for (int k = 0; k < ROWS_PER_WG / 2; ++k) {
        const int i = min(blockIdx.x * ROWS_PER_WG + k * 2 + rsub, nrows - 1);
        src[k] = ids[i];
    }
#pragma unroll
    for (int k = 0; k < ROWS_PER_WG / 2; ++k) v[k] = host[(size_t)src[k] * 128 + unit];
#pragma unroll
    for (int k = 0; k < ROWS_PER_WG / 2; ++k) {
        const int i = blockIdx.x * ROWS_PER_WG + k * 2 + rsub;
        if (i < nrows) dev[(size_t)i * 128 + unit] = v[k];
    }
}

static void touch(char* p, size_t bytes) {      // first touch + a recognisable pattern, 16 threads
    const int T = 16;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([=] {
            const size_t lo = bytes / T * t, hi = t == T - 1 ? bytes : bytes / T * (t + 1);
            for (size_t o = lo; o < hi; o += 4096) memset(p + o, (int)(o >> 12) & 0x7f, 4096);
        });
    for (auto& x : th) x.join();
}

static long anon_huge_kb() {
    FILE* f = fopen("/proc/self/smaps_rollup", "r");
    if (!f) return -1;
    char line[256]; long kb = -1;
    while (fgets(line, sizeof line, f)) if (sscanf(line, "AnonHugePages: %ld kB", &kb) == 1) break;
    fclose(f);
    return kb;
}

template <int RPW>
static float run(const u32x4* host, u32x4* dev, const long long* ids, int nrows, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = (nrows + RPW - 1) / RPW;
    hipLaunchKernelGGL((gather<RPW>), dim3(grid), dim3(256), 0, 0, host, dev, ids, nrows);
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gather<RPW>), dim3(grid), dim3(256), 0, 0, host, dev, ids, nrows);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms / iters);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

static void bench(const char* what, const u32x4* dptr, size_t table_rows, u32x4* dev, long long* ids_d, int nrows) {
    std::vector<long long> h(nrows);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < nrows; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (long long)(s % table_rows); }
    CK(hipMemcpy(ids_d, h.data(), nrows * sizeof(long long), hipMemcpyHostToDevice));
    const double mb = nrows * 2048.0 / 1e6;
    const float t8 = run<8>(dptr, dev, ids_d, nrows, 8), t16 = run<16>(dptr, dev, ids_d, nrows, 8);
    printf("%-58s 8 rows/WG %7.1f us %5.1f GB/s | 16 rows/WG %7.1f us %5.1f GB/s\n", what, t8 * 1e3, mb / t8, t16 * 1e3, mb / t16);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const size_t big_gb = argc > 1 ? (size_t)atoi(argv[1]) : 96;
    const int nrows = 16384;
    {
        FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        char line[128] = "?";
        if (f) { if (!fgets(line, sizeof line, f)) strcpy(line, "?"); fclose(f); }
        printf("transparent_hugepage/enabled: %s", line);
    }
    u32x4* dev; CK(hipMalloc(&dev, (size_t)nrows * 2048));
    long long* ids_d; CK(hipMalloc(&ids_d, nrows * sizeof(long long)));
    for (size_t gb : {(size_t)8, big_gb}) {
        const size_t bytes = gb << 30, rows = bytes / 2048;
        char name[128];
        {   // hipHostMalloc
            void* p; CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
            touch((char*)p, bytes);
            snprintf(name, sizeof name, "hipHostMalloc %zu GiB, random rows of the whole table", gb);
            bench(name, (const u32x4*)p, rows, dev, ids_d, nrows);
            snprintf(name, sizeof name, "hipHostMalloc %zu GiB, random rows of its first 2 GiB", gb);
            bench(name, (const u32x4*)p, (size_t)(2ull << 30) / 2048, dev, ids_d, nrows);
            CK(hipHostFree(p));
        }
        {   // anonymous memory, MADV_HUGEPAGE, registered
            void* p = mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (p == MAP_FAILED) { printf("mmap failed\n"); continue; }
            char* al = (char*)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
            const int adv = madvise(al, bytes, MADV_HUGEPAGE);
            touch(al, bytes);
            const long huge = anon_huge_kb();
            hipError_t e = hipHostRegister(al, bytes, hipHostRegisterMapped);
            if (e != hipSuccess) { printf("hipHostRegister(%zu GiB) failed: %s\n", gb, hipGetErrorString(e)); (void)hipGetLastError(); munmap(p, bytes + (2 << 20)); continue; }
            void* d; CK(hipHostGetDevicePointer(&d, al, 0));
            snprintf(name, sizeof name, "mmap+MADV_HUGEPAGE(rc %d, %ld MB huge)+hipHostRegister %zu GiB", adv, huge / 1024, gb);
            bench(name, (const u32x4*)d, rows, dev, ids_d, nrows);
            CK(hipHostUnregister(al));
            munmap(p, bytes + (2 << 20));
        }
    }
    return 0;
}
