// Ablation probe for the landmark scan: full kernel vs loads-only vs no-statistics-tail.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I shadowkv_amd/csrc tools/score_probe.hip -o /tmp/score_probe
#include "../shadowkv_amd/csrc/skv_select.hip"
#include <stdio.h>
#include <vector>
template <int ABL>
float run(const bf16_t* q, std::vector<bf16_t*>& tabs, bf16_t* D, float* pm, float* ps, int B, int N, int T) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (auto t : tabs) hipLaunchKernelGGL((skv_score_tile_kernel<4, ABL>), dim3(T, B), dim3(64 * SKV_SCORE_WAVES), 0, 0, q, t, D, pm, ps, N, T, 0.088f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int it = 0; it < 5; ++it)
        for (auto t : tabs) hipLaunchKernelGGL((skv_score_tile_kernel<4, ABL>), dim3(T, B), dim3(64 * SKV_SCORE_WAVES), 0, 0, q, t, D, pm, ps, N, T, 0.088f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / (5 * tabs.size());
}
int main() {
    const int B = 8, N = 15560, T = (N + 255) / 256;
    std::vector<bf16_t*> tabs(32);
    for (auto& t : tabs) { hipMalloc(&t, (size_t)B * N * 256); hipMemset(t, 0x3c, (size_t)B * N * 256); }
    bf16_t *q, *D; float *pm, *ps;
    hipMalloc(&q, B * 4 * 256); hipMemset(q, 0x3c, B * 4 * 256); hipMalloc(&D, (size_t)B * 4 * N * 2); hipMalloc(&pm, B * T * 16); hipMalloc(&ps, B * T * 16);
    float full = run<0>(q, tabs, D, pm, ps, B, N, T), mem = run<1>(q, tabs, D, pm, ps, B, N, T), notail = run<2>(q, tabs, D, pm, ps, B, N, T);
    double mb = (double)B * N * 256 / 1e6;
    printf("waves/tile %d: full %.2f us (%.2f TB/s) | loads only %.2f us (%.2f TB/s) | no stats tail %.2f us\n", SKV_SCORE_WAVES, full, mb / full, mem, mb / mem, notail);
    return 0;
}
