// Probe for the landmark scan (skv_score_tile_kernel): request depth per wave (PD) x waves per tile, against "all loads
// up front" (PD = ITERS, the round-1/2 kernel) and the loads-only / no-statistics ablations, at the two BASELINE shapes
// (Llama 122K: B = 8, G = 4, N = 15,560; GLM 200K: B = 4, G = 8, N = 25,544).  Every variant cycles over 32 landmark
// tables (1 GB: far beyond the Infinity Cache), interleaved rounds in one process, median of the rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I shadowkv_amd/csrc tools/score_probe.hip -o /tmp/score_probe
#ifdef PROBE_STAMPS
#define SKV_SCORE_STAMPS
#endif
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

#ifdef PROBE_STAMPS
// per-wave phase stamps of one launch of variant v: where the launch's time goes (us since the first wave's start)
static void stamp_report(Variant& v, const bf16_t* tab, int n_wg, int waves, const char* title) {
    unsigned long long* d; hipMalloc(&d, (size_t)n_wg * 16 * 8 * 8); hipMemset(d, 0, (size_t)n_wg * 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_score_stamps), &d, sizeof(d));
    for (int w = 0; w < 3; ++w) { v.launch(tab); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h((size_t)n_wg * 16 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long* z = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(g_score_stamps), &z, sizeof(z)); hipFree(d);
    unsigned long long t0 = ~0ull;
    for (int g = 0; g < n_wg; ++g) for (int w = 0; w < waves; ++w) { auto x = h[((size_t)g * 16 + w) * 8]; if (x && x < t0) t0 = x; }
    const char* names[6] = {"loads issued", "first row group's data", "last row group's data", "dot products done", "behind the barrier", "statistics done"};
    printf("%s: per-wave stamps, us since the first wave's start (min / median / p90 / max over %d waves)\n", title, n_wg * waves);
    for (int i = 0; i < 6; ++i) {
        std::vector<float> a;
        for (int g = 0; g < n_wg; ++g) for (int w = 0; w < waves; ++w) { auto x = h[((size_t)g * 16 + w) * 8 + i]; if (x) a.push_back((x - t0) / 100.0f); }
        if (a.empty()) continue;
        std::sort(a.begin(), a.end());
        printf("   %-26s %6.2f / %6.2f / %6.2f / %6.2f\n", names[i], a[0], a[a.size() / 2], a[a.size() * 9 / 10], a.back());
    }
    // per workgroup: when its LAST wave has its last data, when it passes the barrier, when its last wave ends
    std::vector<float> wl, wb, we;
    for (int g = 0; g < n_wg; ++g) {
        unsigned long long l = 0, bb = 0, e = 0;
        for (int w = 0; w < waves; ++w) { l = std::max(l, h[((size_t)g * 16 + w) * 8 + 2]); bb = std::max(bb, h[((size_t)g * 16 + w) * 8 + 4]); e = std::max(e, h[((size_t)g * 16 + w) * 8 + 5]); }
        wl.push_back((l - t0) / 100.0f); wb.push_back((bb - t0) / 100.0f); we.push_back((e - t0) / 100.0f);
    }
    auto pr = [](const char* n, std::vector<float>& a) { std::sort(a.begin(), a.end()); printf("   per workgroup: %-22s %6.2f / %6.2f / %6.2f / %6.2f\n", n, a[0], a[a.size() / 2], a[a.size() * 9 / 10], a.back()); };
    pr("last data", wl); pr("barrier passed", wb); pr("last wave done", we);
}

#else
static void stamp_report(Variant&, const bf16_t*, int, int, const char*) {}
#endif

// Pure stream of `total` bytes: workgroups of WAVES waves, every wave LOADS coalesced 1-KiB loads issued up front, folded and dropped.
// Same bytes, different numbers of workgroups / waves: what part of the scan's "loads-only" time is dispatch and what is bandwidth.
template <int WAVES, int LOADS>
__global__ __launch_bounds__(64 * WAVES) void probe_stream_kernel(const u32x4* __restrict__ src, size_t n_vec, uint32_t* sink) {
    const size_t base = ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * LOADS * 64 + (threadIdx.x & 63);
    u32x4 x[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        size_t v = base + (size_t)i * 64;
        x[i] = __builtin_nontemporal_load(src + (v < n_vec ? v : n_vec - 1));
    }
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) f ^= x[i][0] ^ x[i][1] ^ x[i][2] ^ x[i][3];
    if (f == 0x12345u) *sink = f;
}
template <int WAVES, int LOADS>
static Variant make_stream(const char* name, size_t bytes, uint32_t* sink) {
    const size_t n_vec = bytes / 16, per_wg = (size_t)WAVES * LOADS * 64;
    const int grid = (int)((n_vec + per_wg - 1) / per_wg);
    char buf[128]; snprintf(buf, sizeof buf, "%s (%d workgroups, %d waves)", name, grid, grid * WAVES);
    return Variant{buf, [=](const bf16_t* tab) {
        hipLaunchKernelGGL((probe_stream_kernel<WAVES, LOADS>), dim3(grid), dim3(64 * WAVES), 0, 0, (const u32x4*)tab, n_vec, sink);
    }, {}};
}

int main() {
    {
        const size_t bytes = (size_t)8 * 15560 * 256;
        std::vector<bf16_t*> tabs(32);
        for (auto& t : tabs) { hipMalloc(&t, bytes); hipMemset(t, 0x3c, bytes); }
        uint32_t* sink; hipMalloc(&sink, 4);
        std::vector<Variant> vs;
        vs.push_back(make_stream<16, 4>("16 waves x 4 KiB", bytes, sink));
        vs.push_back(make_stream<16, 8>("16 waves x 8 KiB", bytes, sink));
        vs.push_back(make_stream<16, 16>("16 waves x 16 KiB", bytes, sink));
        vs.push_back(make_stream<8, 8>(" 8 waves x 8 KiB", bytes, sink));
        vs.push_back(make_stream<8, 16>(" 8 waves x 16 KiB", bytes, sink));
        vs.push_back(make_stream<4, 16>(" 4 waves x 16 KiB", bytes, sink));
        vs.push_back(make_stream<4, 32>(" 4 waves x 32 KiB", bytes, sink));
        vs.push_back(make_stream<4, 4>(" 4 waves x 4 KiB", bytes, sink));
        vs.push_back(make_stream<1, 16>(" 1 wave  x 16 KiB", bytes, sink));
        bench(vs, tabs, bytes / 1e6, "pure stream of 31.87 MB");
        for (auto t : tabs) hipFree(t);
    }
    {
        const int B = 8, G = 4, N = 15560, T = (N + 255) / 256;
        std::vector<bf16_t*> tabs(32);
        for (auto& t : tabs) { hipMalloc(&t, (size_t)B * N * 256); hipMemset(t, 0x3c, (size_t)B * N * 256); }
        bf16_t *q, *D; float *pm, *ps;
        hipMalloc(&q, B * G * 256); hipMemset(q, 0x3c, B * G * 256); hipMalloc(&D, (size_t)B * G * N * 2);
        hipMalloc(&pm, B * T * G * 4); hipMalloc(&ps, B * T * G * 4);
        std::vector<Variant> vs;
        vs.push_back(make<4, 0, 16, 4>("G4 16 waves x 4 row groups (shipped)", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 8, 8>("G4  8 waves x 8 row groups, 128 VGPRs", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 4, 16>("G4  4 waves x 16 row groups, 256 VGPRs", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 2, 32>("G4  2 waves x 32 row groups", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 1, 16, 4>("G4 16 waves, loads only", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 1, 8, 8>("G4  8 waves, loads only", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 1, 4, 16>("G4  4 waves, loads only", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 1, 2, 32>("G4  2 waves, loads only", q, D, pm, ps, B, N, T));
        bench(vs, tabs, (double)B * N * 256 / 1e6, "Llama-3.1-8B 122K: B 8, G 4, N 15560, 488 tiles");
        stamp_report(vs[0], tabs[5], T * B, 16, "G4 16 waves");
        stamp_report(vs[2], tabs[5], T * B, 4, "G4 4 waves");
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
        vs.push_back(make<8, 0, 8, 8>("G8  8 waves x 8 row groups (shipped)", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 4, 16>("G8  4 waves x 16 row groups, 256 VGPRs", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 2, 32>("G8  2 waves x 32 row groups", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 1, 8, 8>("G8  8 waves, loads only", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 1, 4, 16>("G8  4 waves, loads only", q, D, pm, ps, B, N, T));
        bench(vs, tabs, (double)B * N * 256 / 1e6, "GLM-4-9B 200K: B 4, G 8, N 25544, 400 tiles");
        stamp_report(vs[0], tabs[5], T * B, 8, "G8 8 waves");
        stamp_report(vs[1], tabs[5], T * B, 4, "G8 4 waves");
        for (auto t : tabs) hipFree(t);
        hipFree(q); hipFree(D); hipFree(pm); hipFree(ps);
    }
    {
        const int B = 4, G = 8, N = 15560, T = (N + 255) / 256;
        std::vector<bf16_t*> tabs(48);
        for (auto& t : tabs) { hipMalloc(&t, (size_t)B * N * 256); hipMemset(t, 0x3c, (size_t)B * N * 256); }
        bf16_t *q, *D; float *pm, *ps;
        hipMalloc(&q, B * G * 256); hipMemset(q, 0x3c, B * G * 256); hipMalloc(&D, (size_t)B * G * N * 2);
        hipMalloc(&pm, B * T * G * 4); hipMalloc(&ps, B * T * G * 4);
        std::vector<Variant> vs;
        vs.push_back(make<8, 0, 8, 8>("G8  8 waves (shipped)", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 16, 4>("G8 16 waves", q, D, pm, ps, B, N, T));
        vs.push_back(make<8, 0, 4, 16>("G8  4 waves", q, D, pm, ps, B, N, T));
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
        vs.push_back(make<4, 0, 16, 4>("G4 16 waves (shipped)", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 8, 8>("G4  8 waves", q, D, pm, ps, B, N, T));
        vs.push_back(make<4, 0, 4, 16>("G4  4 waves", q, D, pm, ps, B, N, T));
        bench(vs, tabs, (double)B * N * 256 / 1e6, "Llama 60K: B 8, G 4, N 7672, 240 tiles");
    }
    return 0;
}
