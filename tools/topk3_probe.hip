// Probe for the fused selection (round 4): scan (keys + slot-major logits) -> top-k with the logit-domain prefilter, against
// the three-launch chain scan -> normalise -> top-k, on synthetic landmarks with a walking query (so that the second step
// runs the fast path: good log-normalisers from the first).  With -DSKV_TOPK_STAMPS: phase stamps of the fused top-k launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DSKV_TOPK_STAMPS] -I shadowkv_amd/csrc tools/topk3_probe.hip -o /tmp/topk3_probe
//   TOPK_PROBE_B / _G / _N / _S: shape (default 8 / 4 / 15560 / 256)
#include "../shadowkv_amd/csrc/skv_select.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float gauss() { float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
int main() {
    const int B = getenv("TOPK_PROBE_B") ? atoi(getenv("TOPK_PROBE_B")) : 8, G = getenv("TOPK_PROBE_G") ? atoi(getenv("TOPK_PROBE_G")) : 4;
    const int N = getenv("TOPK_PROBE_N") ? atoi(getenv("TOPK_PROBE_N")) : 15560, S = getenv("TOPK_PROBE_S") ? atoi(getenv("TOPK_PROBE_S")) : 256;
    const int T = (N + 255) / 256, stride = (N + 7) & ~7;
    srand(5);
    std::vector<uint16_t> lm((size_t)B * N * 128), q((size_t)B * G * 128);
    for (auto& v : lm) v = f2b(gauss());
    std::vector<float> qf(q.size());
    for (auto& v : qf) v = 2.0f * gauss();
    std::vector<int64_t> idx((size_t)B * N), cached((size_t)B * S);
    for (int b = 0; b < B; ++b) for (int j = 0; j < N; ++j) idx[(size_t)b * N + j] = j + j / 300;
    for (auto& c : cached) c = rand() % N;
    uint16_t *dlm, *dq, *dD, *dscore; float *dpm, *dps, *dct; int64_t *didx, *dc; int32_t *doff, *dcnt, *dslot;
    hipMalloc(&dlm, lm.size() * 2); hipMalloc(&dq, q.size() * 2); hipMalloc(&dD, (size_t)B * G * N * 2); hipMalloc(&dscore, (size_t)B * stride * 2);
    hipMalloc(&dpm, B * T * G * 4); hipMalloc(&dps, B * T * G * 4); hipMalloc(&dct, B * G * 4); hipMemset(dct, 0, B * G * 4);
    int* dlvl; hipMalloc(&dlvl, B * 12); hipMemset(dlvl, 0, B * 12);
    hipMalloc(&didx, idx.size() * 8); hipMalloc(&dc, cached.size() * 8); hipMalloc(&doff, B * S * 4); hipMalloc(&dcnt, B * 4); hipMalloc(&dslot, B * S * 4);
    hipMemcpy(dlm, lm.data(), lm.size() * 2, hipMemcpyHostToDevice); hipMemcpy(didx, idx.data(), idx.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, cached.data(), cached.size() * 8, hipMemcpyHostToDevice);
    FusedSel fs{dct, dscore, dD, stride};
    FusedTop ft{dD, dpm, dps, dct, dlvl, dlvl + B, T};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto new_q = [&](float step) {
        for (size_t i = 0; i < q.size(); ++i) { qf[i] += step * gauss(); q[i] = f2b(qf[i]); }
        hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
    };
    for (int step = 0; step < 5; ++step) {
        new_q(step ? 0.3f : 0.f);
        skv_launch_score(dq, dlm, dD, dpm, dps, B, G, N, 0.0884f, 0, nullptr, &fs);
        hipDeviceSynchronize();
        int rc = skv_launch_topk_resident(dscore, stride, didx, nullptr, dc, doff, dcnt, nullptr, dslot, B, N, S, S, nullptr, 0, nullptr, &ft, G);
        hipDeviceSynchronize();
#ifdef SKV_TOPK_STAMPS
        unsigned long long st[24]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_topk_stamps), sizeof(st));
        printf("step %d rc=%d total %.2f us | loads+finals %.2f level guess %.2f search1 (if any) %.2f level out %.2f gather+exact %.2f search2 %.2f place %.2f | F..classify %.2f lookup %.2f scan3 %.2f vote %.2f write %.2f\n",
               step, rc, (st[10] - st[0]) / 100.0, (st[12] - st[0]) / 100.0, (st[13] - st[12]) / 100.0, (st[14] - st[13]) / 100.0, (st[15] - st[14]) / 100.0,
               (st[16] - st[15]) / 100.0, (st[17] - st[16]) / 100.0, (st[18] - st[17]) / 100.0, (st[6] - st[18]) / 100.0, (st[7] - st[6]) / 100.0,
               (st[8] - st[7]) / 100.0, (st[9] - st[8]) / 100.0, (st[10] - st[9]) / 100.0);
#else
        (void)rc;
#endif
    }
    // event-timed pairs, back to back: fused (scan + top-k) vs three launches
    float ms;
    hipEventRecord(e0);
    for (int it = 0; it < 40; ++it) {
        skv_launch_score(dq, dlm, dD, dpm, dps, B, G, N, 0.0884f, 0, nullptr, &fs);
        skv_launch_topk_resident(dscore, stride, didx, nullptr, dc, doff, dcnt, nullptr, dslot, B, N, S, S, nullptr, 0, nullptr, &ft, G);
    }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("fused  scan + top-k:             %.2f us per pair\n", ms * 1e3 / 40);
    hipEventRecord(e0);
    for (int it = 0; it < 40; ++it) {
        skv_launch_score(dq, dlm, dD, dpm, dps, B, G, N, 0.0884f, 0, nullptr, nullptr);
        skv_launch_normalize_groupmax(dD, dpm, dps, nullptr, dscore, stride, B, G, N, 0, nullptr);
        skv_launch_topk_resident(dscore, stride, didx, nullptr, dc, doff, dcnt, nullptr, dslot, B, N, S, S, nullptr, 0, nullptr, nullptr, 0);
    }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("three  scan + normalise + top-k: %.2f us per triple\n", ms * 1e3 / 40);
    hipEventRecord(e0);
    for (int it = 0; it < 40; ++it) skv_launch_topk_resident(dscore, stride, didx, nullptr, dc, doff, dcnt, nullptr, dslot, B, N, S, S, nullptr, 0, nullptr, nullptr, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("       top-k alone (scores in):  %.2f us\n", ms * 1e3 / 40);
    skv_launch_score(dq, dlm, dD, dpm, dps, B, G, N, 0.0884f, 0, nullptr, &fs);
    hipEventRecord(e0);
    for (int it = 0; it < 40; ++it) skv_launch_topk_resident(dscore, stride, didx, nullptr, dc, doff, dcnt, nullptr, dslot, B, N, S, S, nullptr, 0, nullptr, &ft, G);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("       fused top-k alone:        %.2f us\n", ms * 1e3 / 40);
    return 0;
}
