// Probe for the landmark scan (skv_score_tile_kernel): request depth per wave (PD) x waves per tile, against "all loads
// up front" (PD = ITERS, the round-1/2 kernel) and the loads-only / no-statistics ablations, at the two BASELINE shapes
// (Llama 122K: B = 8, G = 4, N = 15,560; GLM 200K: B = 4, G = 8, N = 25,544).  Every variant cycles over 32 landmark
// tables (1 GB: far beyond the Infinity Cache), interleaved rounds in one process, median of the rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I shadowkv_amd/csrc tools/score_probe.hip -o /tmp/score_probe
#include "../shadowkv_amd/csrc/skv_select.hip"
#include <stdio.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

struct Variant {
    std::string name;
    std::function<void(const bf16_t*)> launch;
    std::vector<float> us;
};

template <int G, int ABL, int WAVES, int PD>
Variant make(const char* name, const bf16_t* q, bf16_t* D, float* pm, float* ps, int B, int N, int T) {
    return Variant{name, [=](const bf16_t* tab) {
        hipLaunchKernelGGL((skv_score_tile_kernel<G, ABL, WAVES, PD>), dim3(T, B), dim3(64 * WAVES), 0, 0, q, tab, D, pm, ps, N, T, 0.088f, EarlyHooks{}, FusedSel{});
    }, {}};
}

static void bench(std::vector<Variant>& vs, std::vector<bf16_t*>& tabs, double mb, const char* title) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (auto& v : vs) for (auto t : tabs) v.launch(t);
    hipDeviceSynchronize();
    for (int round = 0; round < 7; ++round)
        for (auto& v : vs) {
            hipEventRecord(a);
            for (int it = 0; it < 3; ++it) for (auto t : tabs) v.launch(t);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            v.us.push_back(ms * 1e3f / (3 * tabs.size()));
        }
    printf("%s (%.2f MB per launch)\n", title, mb);
    for (auto& v : vs) {
        std::sort(v.us.begin(), v.us.end());
        const float med = v.us[v.us.size() / 2];
        printf("  %-44s median %6.2f us (%5.2f TB/s)  min %6.2f\n", v.name.c_str(), med, mb / med, v.us[0]);
    }
}

int main() {
    {
        const int B = 8, G = 4, N = 15560, T = (N + 255) / 256;
        std::vector<bf16_t*> tabs(32);
        for (auto& t : tabs) { hipMalloc(&t, (size_t)B * N * 256); hipMemset(t, 0x3c, (size_t)B * N * 256); }
        bf16_t *q, *D; float *pm, *ps;
        hipMalloc(&q, B * G * 256); hipMemset(q, 0x3c, B * G * 256); hipMalloc(&D, (size_t)B * G * N * 2);
        hipMalloc(&pm, B * T * G * 4); hipMalloc(&ps, B * T * G * 4);
        std::vector<Variant> vs;
        vs.push_back(make<4, 0, 16, 4>("G4 16 waves, all 4 loads up front (r2)", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 16, 1>("G4 16 waves, depth 1", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 16, 2>("G4 16 waves, depth 2", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 16, 3>("G4 16 waves, depth 3", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 8, 2>("G4  8 waves, depth 2", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 8, 3>("G4  8 waves, depth 3", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 8, 4>("G4  8 waves, depth 4", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 4, 4>("G4  4 waves, depth 4", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 4, 6>("G4  4 waves, depth 6", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 1, 16, 4>("G4 16 waves, loads only, all up front", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 1, 16, 2>("G4 16 waves, loads only, depth 2", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 2, 16, 2>("G4 16 waves, depth 2, no statistics tail", q, D, pm, ps, B, N, T));
        bench(vs, tabs, (double)B * N * 256 / 1e6, "Llama-3.1-8B 122K: B 8, G 4, N 15560, 488 tiles");
        for (auto t : tabs) hipFree(t);
        hipFree(q); hipFree(D); hipFree(pm); hipFree(ps);
    }
    {
        const int B = 4, G = 8, N = 25544, T = (N + 255) / 256;
        std::vector<bf16_t*> tabs(40);
        for (auto& t : tabs) { hipMalloc(&t, (size_t)B * N * 256); hipMemset(t, 0x3c, (size_t)B * N * 256); }
        bf16_t *q, *D; float *pm, *ps;
        hipMalloc(&q, B * G * 256); hipMemset(q, 0x3c, B * G * 256); hipMalloc(&D, (size_t)B * G * N * 2);
        hipMalloc(&pm, B * T * G * 4); hipMalloc(&ps, B * T * G * 4);
        std::vector<Variant> vs;
        vs.push_back(make<8, 0, 8, 8>("G8  8 waves, all 8 loads up front (r2)", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 8, 2>("G8  8 waves, depth 2", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 8, 3>("G8  8 waves, depth 3", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 8, 4>("G8  8 waves, depth 4", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 8, 6>("G8  8 waves, depth 6", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 16, 2>("G8 16 waves (one tile per CU), depth 2", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 16, 4>("G8 16 waves (one tile per CU), all up front", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 4, 4>("G8  4 waves, depth 4", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 4, 8>("G8  4 waves, depth 8", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 1, 8, 8>("G8  8 waves, loads only, all up front", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 1, 8, 3>("G8  8 waves, loads only, depth 3", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 2, 8, 3>("G8  8 waves, depth 3, no statistics tail", q, D, pm, ps, B, N, T));
        bench(vs, tabs, (double)B * N * 256 / 1e6, "GLM-4-9B 200K: B 4, G 8, N 25544, 400 tiles");
    }
    {   // Yi-9B-200K at 122K / Llama 60K-style shapes: fewer tiles than CUs - one tile's latency chain
        const int B = 4, G = 8, N = 15560, T = (N + 255) / 256;
        std::vector<bf16_t*> tabs(48);
        for (auto& t : tabs) { hipMalloc(&t, (size_t)B * N * 256); hipMemset(t, 0x3c, (size_t)B * N * 256); }
        bf16_t *q, *D; float *pm, *ps;
        hipMalloc(&q, B * G * 256); hipMemset(q, 0x3c, B * G * 256); hipMalloc(&D, (size_t)B * G * N * 2);
        hipMalloc(&pm, B * T * G * 4); hipMalloc(&ps, B * T * G * 4);
        std::vector<Variant> vs;
        vs.push_back(make<8, 0, 8, 8>("G8  8 waves, all up front (shipped)", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 16, 4>("G8 16 waves, all up front", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 1, 8, 8>("G8  8 waves, loads only", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 1, 16, 4>("G8 16 waves, loads only", q, D, pm, ps, B, N, T));
        bench(vs, tabs, (double)B * N * 256 / 1e6, "Yi-9B 122K: B 4, G 8, N 15560, 244 tiles");
        for (auto t : tabs) hipFree(t);
        hipFree(q); hipFree(D); hipFree(pm); hipFree(ps);
    }
    {
        const int B = 8, G = 4, N = 7672, T = (N + 255) / 256;
        std::vector<bf16_t*> tabs(64);
        for (auto& t : tabs) { hipMalloc(&t, (size_t)B * N * 256); hipMemset(t, 0x3c, (size_t)B * N * 256); }
        bf16_t *q, *D; float *pm, *ps;
        hipMalloc(&q, B * G * 256); hipMemset(q, 0x3c, B * G * 256); hipMalloc(&D, (size_t)B * G * N * 2);
        hipMalloc(&pm, B * T * G * 4); hipMalloc(&ps, B * T * G * 4);
        std::vector<Variant> vs;
        vs.push_back(make<4, 0, 16, 4>("G4 16 waves, all up front (shipped)", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 8, 8>("G4  8 waves, all up front", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 1, 16, 4>("G4 16 waves, loads only", q, D, pm, ps, B, N, T));
        bench(vs, tabs, (double)B * N * 256 / 1e6, "Llama 60K: B 8, G 4, N 7672, 240 tiles");
    }
    return 0;
}
