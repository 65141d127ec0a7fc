// Round 3 follow-up of tools/overlap2_probe.hip: round 2 measured that an HBM-streaming kernel runs at about half speed
// while host-memory (PCIe) reads are in flight anywhere on the chip.  Is that a property of the whole memory system, or of
// the L2 / fabric port of the XCD whose CUs issue the host reads?  Here the PCIe-pulling workgroups CONFINE themselves to
// the XCDs of a mask (every workgroup reads HW_REG_XCC_ID; the ones elsewhere exit at once; the ones inside claim a worker
// id and pull 16 KB pieces from a shared counter until the buffer is done), the HBM-streaming kernel runs beside them.
//   hipcc --offload-arch=gfx950 -O3 tools/overlap3_probe.hip -o tools/bin/overlap3_probe ; tools/bin/overlap3_probe <host_mb>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}
// H: streams n16 16-byte words from HBM.  skip_mask: workgroups on those XCDs do nothing (their share is NOT redistributed:
// used only to see what the streaming rate of the remaining XCDs is)
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
// P: ctl[0] = workers claimed, ctl[1] = next piece, ctl[2..9] = pieces done per XCD (statistics)
__global__ __launch_bounds__(256) void kP(const u32x4* __restrict__ host, u32x4* __restrict__ dst, unsigned npieces, unsigned xcd_mask,
                                          unsigned max_workers, unsigned* __restrict__ ctl) {
    __shared__ unsigned s_piece;
    const unsigned x = xcc_id();
    if (!((xcd_mask >> x) & 1u)) return;
    if (threadIdx.x == 0) s_piece = atomicAdd(&ctl[0], 1u);
    __syncthreads();
    if (s_piece >= max_workers) return;
    __syncthreads();
    unsigned done = 0;
    for (;;) {
        if (threadIdx.x == 0) s_piece = atomicAdd(&ctl[1], 1u);
        __syncthreads();
        const unsigned p = s_piece;
        __syncthreads();
        if (p >= npieces) break;
        const size_t base = (size_t)p * 1024 + threadIdx.x;        // 1,024 16-byte words per piece
        u32x4 a = host[base], b = host[base + 256], c = host[base + 512], d = host[base + 768];
        dst[base] = a; dst[base + 256] = b; dst[base + 512] = c; dst[base + 768] = d;
        ++done;
    }
    if (threadIdx.x == 0) atomicAdd(&ctl[2 + x], done);
}
int main(int argc, char** argv) {
    const size_t host_mb = argc > 1 ? atoi(argv[1]) : 16, hbm_mb = 2048;
    u32x4 *src, *dst, *host; unsigned *out, *ctl;
    CK(hipMalloc(&src, hbm_mb << 20)); CK(hipMemset(src, 1, hbm_mb << 20)); CK(hipMalloc(&dst, host_mb << 20)); CK(hipMalloc(&out, 64));
    CK(hipMalloc(&ctl, 64));
    CK(hipHostMalloc(&host, host_mb << 20, hipHostMallocMapped));
    for (size_t i = 0; i < (host_mb << 20) / 16; ++i) host[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    hipEvent_t e0, e1, ea, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const unsigned npieces = (unsigned)((host_mb << 20) / 16384);
    auto run = [&](bool h, bool p, unsigned mask, unsigned workers, const char* what) {
        float best[3] = {1e9f, 1e9f, 1e9f};
        unsigned stat[16];
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(ctl, 0, 64));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0)); CK(hipStreamWaitEvent(sa, e0, 0)); CK(hipStreamWaitEvent(sb, e0, 0));
            if (p) hipLaunchKernelGGL(kP, dim3(2048), dim3(256), 0, sb, host, dst, npieces, mask, workers, ctl);
            if (h) for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(kH, dim3(2048), dim3(256), 0, sa, src + (size_t)k * (16 << 20), out, (size_t)(256 << 20) / 16);
            CK(hipEventRecord(ea, sa)); CK(hipEventRecord(eb, sb));
            CK(hipStreamWaitEvent(0, ea, 0)); CK(hipStreamWaitEvent(0, eb, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms, ma, mb; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&ma, e0, ea)); CK(hipEventElapsedTime(&mb, e0, eb));
            if (rep > 0 && ms < best[0]) { best[0] = ms; best[1] = ma; best[2] = mb; CK(hipMemcpy(stat, ctl, 64, hipMemcpyDeviceToHost)); }
        }
        printf("%-46s total %7.1f us", what, best[0] * 1e3);
        if (h) printf(" | H done %7.1f us (%.2f TB/s)", best[1] * 1e3, 8 * 256.0 * 1.048576e-6 / (best[1] * 1e-3));
        if (p) {
            printf(" | P done %7.1f us (%4.1f GB/s), workers %u, pieces per XCD:", best[2] * 1e3, host_mb * 1.048576e-3 / (best[2] * 1e-3), stat[0] < workers ? stat[0] : workers);
            for (int x = 0; x < 8; ++x) printf(" %u", stat[2 + x]);
        }
        printf("\n");
    };
    printf("host buffer %zu MB (%u pieces of 16 KB), HBM stream 2 GB in 8 launches\n", host_mb, npieces);
    run(true, false, 0, 0, "H alone");
    run(false, true, 0xff, 64, "P alone, all XCDs, 64 workers");
    run(false, true, 0x01, 64, "P alone, XCD 0 only, up to 64 workers");
    run(false, true, 0x01, 32, "P alone, XCD 0 only, up to 32 workers");
    run(false, true, 0x03, 64, "P alone, XCDs 0-1, up to 64 workers");
    run(true, true, 0xff, 64, "H + P, all XCDs, 64 workers");
    run(true, true, 0xff, 16, "H + P, all XCDs, 16 workers");
    run(true, true, 0x01, 64, "H + P, P on XCD 0 only, up to 64 workers");
    run(true, true, 0x01, 32, "H + P, P on XCD 0 only, up to 32 workers");
    run(true, true, 0x01, 16, "H + P, P on XCD 0 only, up to 16 workers");
    run(true, true, 0x03, 64, "H + P, P on XCDs 0-1, up to 64 workers");
    run(true, false, 0, 0, "H alone (again)");
    return 0;
}
