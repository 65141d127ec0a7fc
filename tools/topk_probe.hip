// Diagnostic build of the top-k + diff kernel with phase stamps (see TOPK_STAMP in skv_select.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DSKV_TOPK_STAMPS] -I shadowkv_amd/csrc tools/topk_probe.hip -o /tmp/topk_probe
// (without the define: the shipped kernel, event-timed back to back only)
#include "../shadowkv_amd/csrc/skv_select.hip"
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char** argv) {
    const bool inplace = argc > 1 && !strcmp(argv[1], "inplace");
    // TOPK_PROBE_N / TOPK_PROBE_B: another shape (GLM-4 at 200K: N 25544, B 4; the 1M-token context: N 131056, B 8)
    const int B = getenv("TOPK_PROBE_B") ? atoi(getenv("TOPK_PROBE_B")) : 8, N = getenv("TOPK_PROBE_N") ? atoi(getenv("TOPK_PROBE_N")) : 15560;
    const int S = 256, stride = (N + 7) & ~7;
    const int R = argc > 6 ? atoi(argv[6]) : S;
    const bool nolm = getenv("TOPK_PROBE_NOLM") != nullptr;   // identity slot -> id map: no id gathers at all     // argv[6]: resident slots per head (> S: least-recently-selected replacement)
    std::vector<uint16_t> sc((size_t)B * stride);
    std::vector<int64_t> lm((size_t)B * N), cached((size_t)B * R);
    srand(3);
    // scores shaped like the decode step's softmax probabilities: log-normal, sigma from argv (0 = flat +-40 %)
    const float sigma = argc > 4 ? atof(argv[4]) : 0.f;
    for (auto& v : sc) {
        float p;
        if (sigma == 0.f) p = 6.4e-5f * (0.6f + 0.8f * rand() / RAND_MAX);
        else {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
            p = 6.4e-5f * expf(sigma * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
        }
        uint32_t u; memcpy(&u, &p, 4); v = u >> 16;
    }
    if (argc > 5 && atoi(argv[5]) > 0) {   // argv[5] = run length: the top scores sit in runs of that many consecutive landmark slots (attention locality)
        const int runlen = atoi(argv[5]), nruns = (S + runlen - 1) / runlen;
        for (int b = 0; b < B; ++b)
            for (int r = 0; r < nruns; ++r) {
                const int start = (int)((size_t)rand() % (N - runlen));
                for (int j = start; j < start + runlen; ++j) {
                    float p = 1e-3f * (1.0f + 0.5f * rand() / RAND_MAX);
                    uint32_t u; memcpy(&u, &p, 4); sc[(size_t)b * stride + j] = u >> 16;
                }
            }
    }
    for (int b = 0; b < B; ++b) for (int j = 0; j < N; ++j) lm[(size_t)b * N + j] = j + j / 300;
    // resident ids: random chunk ids (an arithmetic progression like (61 j) mod N is a worst case of the multiplicative
    // hash at table sizes 1024 / 2048 - stride 0.70007 of the table - and doubles the hash-build and lookup phases)
    for (int b = 0; b < B; ++b) for (int j = 0; j < R; ++j) cached[(size_t)b * R + j] = (argc > 7 ? (j * 61) % N : rand() % N);
    int32_t* dage; hipMalloc(&dage, (size_t)B * R * 4); hipMemset(dage, 0, (size_t)B * R * 4);
    uint16_t* dsc; int64_t *dlm, *dc, *dsel; int32_t *doff, *dcnt, *dslot;
    hipMalloc(&dsc, sc.size() * 2); hipMalloc(&dlm, lm.size() * 8); hipMalloc(&dc, cached.size() * 8); hipMalloc(&dsel, cached.size() * 8);
    hipMalloc(&doff, B * S * 4); hipMalloc(&dcnt, B * 4); hipMalloc(&dslot, B * S * 4);
    hipMemcpy(dsc, sc.data(), sc.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dlm, lm.data(), lm.size() * 8, hipMemcpyHostToDevice);
    char* flush; const size_t flush_bytes = (size_t)640 << 20; hipMalloc(&flush, flush_bytes);   // > L2 + MALL
    const bool cold = argc > 2 && !strcmp(argv[2], "cold");
    for (int it = 0; it < 4; ++it) {
        hipMemcpy(dc, cached.data(), cached.size() * 8, hipMemcpyHostToDevice);
        if (cold) { hipMemset(flush, it, flush_bytes); hipDeviceSynchronize(); }   // evicts data AND the kernel's code
        if (argc > 3 && !strcmp(argv[3], "warmcode")) {   // same kernel on OTHER buffers first: code warm, data cold
            static uint16_t* dsc2 = nullptr; static int64_t *dlm2, *dc2; static int32_t *doff2, *dcnt2;
            if (!dsc2) {
                hipMalloc(&dsc2, sc.size() * 2); hipMalloc(&dlm2, lm.size() * 8); hipMalloc(&dc2, cached.size() * 8);
                hipMalloc(&doff2, B * S * 4); hipMalloc(&dcnt2, B * 4);
                hipMemcpy(dsc2, sc.data(), sc.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dlm2, lm.data(), lm.size() * 8, hipMemcpyHostToDevice);
                hipMemcpy(dc2, cached.data(), cached.size() * 8, hipMemcpyHostToDevice);
                hipMemset(flush, 9, flush_bytes); hipDeviceSynchronize();
            }
            skv_launch_topk_reorder(dsc2, stride, dlm2, nullptr, dc2, doff2, dcnt2, nullptr, nullptr, B, N, S, 0);
            hipDeviceSynchronize();
        }
        int rc = skv_launch_topk_resident(dsc, stride, nolm ? nullptr : dlm, nullptr, dc, doff, dcnt, nullptr, inplace ? dslot : nullptr, B, N, S, R, R > S ? dage : nullptr, 0, nullptr, nullptr, 0);
        hipDeviceSynchronize();
#ifdef SKV_TOPK_STAMPS
        unsigned long long st[24]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_topk_stamps), sizeof(st));
#ifdef SKV_TOPK_V1
        const char* names[] = {"stage+hist1(+hash build)", "select1+zero", "hist2", "select2", "count", "scan", "assign+gather", "-", "lookup", "scan2", "sorted-vote", "write"};
        const int ns = 12;
#else
        const char* names[] = {"loads+init+max", "hist(+hash build)", "fold+scan+thr", "count+scan", "assign", "id-gather", "lookup", "scan3", "sorted-vote", "write"};
        const int ns = 10;
#endif
        printf("run %d rc=%d total %.2f us :", it, rc, (st[ns] - st[0]) / 100.0);
        for (int i = 0; i < ns; ++i) printf(" %s=%.2f", names[i], (st[i + 1] - st[i]) / 100.0);
        printf("\n");
#else
        (void)rc;
#endif
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) skv_launch_topk_resident(dsc, stride, nolm ? nullptr : dlm, nullptr, dc, doff, dcnt, nullptr, inplace ? dslot : nullptr, B, N, S, R, R > S ? dage : nullptr, 0, nullptr, nullptr, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("back-to-back launches: %.2f us per launch (event time, includes launch boundary)\n", ms * 1e3 / 50);
    return 0;
}
