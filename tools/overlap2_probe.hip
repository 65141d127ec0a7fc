// Can a PCIe-bound fetch kernel confined to a few CUs run beside an HBM-bound streaming kernel without slowing it?
// (design question behind pipelining two half-batches: one half's fetch under the other half's dense work.)
//   kernel H: streams `hbm_mb` MB from HBM (grid fills the chip, like the GEMVs)
//   kernel P: `nwg` persistent workgroups x 256 threads pull `host_mb` MB from pinned host memory with 16-B loads
//             (4 in flight per thread) and store them to HBM; `lds_kb` of LDS per workgroup (160: nothing else fits its CU)
//   hipcc --offload-arch=gfx950 -O3 tools/overlap2_probe.hip -o /tmp/overlap2_probe ; /tmp/overlap2_probe <nwg> <lds_kb> <host_mb>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kH(const u32x4* __restrict__ src, unsigned* __restrict__ out, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void kP(const u32x4* __restrict__ host, u32x4* __restrict__ dst, size_t n16) {
    extern __shared__ unsigned char lds[];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a = host[i], b = host[i + stride], c = host[i + 2 * stride], d = host[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    if (n16 == 1) lds[threadIdx.x] = 1;
}
int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 64, lds_kb = argc > 2 ? atoi(argv[2]) : 0;
    const size_t host_mb = argc > 3 ? atoi(argv[3]) : 16, hbm_mb = 2048;
    u32x4 *src, *dst, *host; unsigned* out;
    hipMalloc(&src, hbm_mb << 20); hipMemset(src, 1, hbm_mb << 20); hipMalloc(&dst, host_mb << 20); hipMalloc(&out, 64);
    hipHostMalloc(&host, host_mb << 20, hipHostMallocMapped); 
    for (size_t i = 0; i < (host_mb << 20) / 16; ++i) host[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
    hipFuncSetAttribute((const void*)kP, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipEvent_t e0, e1, ea, eb; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&ea); hipEventCreate(&eb);
    auto run = [&](bool h, bool p) {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0); hipStreamWaitEvent(sa, e0, 0); hipStreamWaitEvent(sb, e0, 0);
        // H: 8 launches of 256 MB (~45 us each at 5.7 TB/s); P: one launch
        if (p) hipLaunchKernelGGL(kP, dim3(nwg), dim3(256), (size_t)lds_kb * 1024, sb, host, dst, (host_mb << 20) / 16);
        if (h) for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(kH, dim3(2048), dim3(256), 0, sa, src + (size_t)k * (16 << 20), out, (size_t)(256 << 20) / 16);
        hipEventRecord(ea, sa); hipEventRecord(eb, sb);
        hipStreamWaitEvent(0, ea, 0); hipStreamWaitEvent(0, eb, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms, ma, mb; hipEventElapsedTime(&ms, e0, e1); hipEventElapsedTime(&ma, e0, ea); hipEventElapsedTime(&mb, e0, eb);
        printf("  %s%s: total %.1f us", h ? "H" : "", p ? "P" : "", ms * 1e3);
        if (h) printf(" | H done at %.1f us (%.2f TB/s)", ma * 1e3, 8 * 256.0 / 1e6 * 1.048576 / (ma * 1e-3) );
        if (p) printf(" | P done at %.1f us (%.1f GB/s)", mb * 1e3, host_mb * 1.048576e-3 / (mb * 1e-3));
        printf("\n");
    };
    printf("fetch: %d workgroups, %d KB LDS each, %zu MB from host\n", nwg, lds_kb, host_mb);
    run(true, true); run(true, false); run(false, true); run(true, true); run(true, true);
    return 0;
}
