// Probe: how does v_mfma_f32_16x16x32_bf16 round?  Compares the hardware result with candidate
// CPU models so the oracle can restate it.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void probe(const uint16_t* A /*[16][32]*/, const uint16_t* B /*[16 cols][32 k]*/, const float* C, float* D) {
    int l = threadIdx.x;
    u32x4 a = *(const u32x4*)(A + (l & 15) * 32 + 8 * (l >> 4));
    u32x4 b = *(const u32x4*)(B + (l & 15) * 32 + 8 * (l >> 4));
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) * 4 + r) * 16 + (l & 15)];
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = d[r];
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float rnd(float scale) { return scale * ((float)rand() / RAND_MAX * 2.f - 1.f); }

#include <fenv.h>
#include <vector>
struct Model { const char* name; std::vector<std::vector<int>> groups; int rtz; int cfirst; };
static float to_f32(long double s, int rtz) {
    float f = (float)s;
    if (rtz == 1) { if (fabsl((long double)f) > fabsl(s)) f = nextafterf(f, 0.0f); }           // toward zero
    if (rtz == 2) { if ((long double)f > s) f = nextafterf(f, -INFINITY); }                      // toward -inf
    if (rtz == 3) { if ((long double)f < s) f = nextafterf(f, INFINITY); }                       // toward +inf
    return f;
}
int main() {
    srand(1);
    uint16_t hA[16 * 32], hB[16 * 32]; float hC[256], hD[256];
    uint16_t *dA, *dB; float *dC, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dC, sizeof hC); (void)hipMalloc(&dD, sizeof hD);
    std::vector<Model> models;
    auto nat = [](int gs) { std::vector<std::vector<int>> g; for (int i = 0; i < 32; i += gs) { std::vector<int> v; for (int k = i; k < i + gs; ++k) v.push_back(k); g.push_back(v); } return g; };
    for (int gs : {1, 2, 4, 8, 16, 32}) for (int rtz : {0, 1, 2, 3}) for (int cf : {0, 1}) models.push_back({"natural", nat(gs), rtz, cf});
    { std::vector<std::vector<int>> g; for (int half = 0; half < 2; ++half) for (int h = 0; h < 4; ++h) { std::vector<int> v; for (int j = 0; j < 4; ++j) v.push_back(8 * h + 4 * half + j); g.push_back(v); }
      for (int rtz : {0, 1, 2, 3}) for (int cf : {0, 1}) models.push_back({"g4 (half,h)", g, rtz, cf}); }
    { std::vector<std::vector<int>> g; for (int half = 0; half < 2; ++half) { std::vector<int> v; for (int h = 0; h < 4; ++h) for (int j = 0; j < 4; ++j) v.push_back(8 * h + 4 * half + j); g.push_back(v); }
      for (int rtz : {0, 1, 2, 3}) for (int cf : {0, 1}) models.push_back({"g16 by half(j<4|j>=4)", g, rtz, cf}); }
    { std::vector<std::vector<int>> g; for (int j = 0; j < 8; ++j) { std::vector<int> v; for (int h = 0; h < 4; ++h) v.push_back(8 * h + j); g.push_back(v); }
      for (int rtz : {0, 1, 2, 3}) for (int cf : {0, 1}) models.push_back({"g4 by j (k=8h+j over h)", g, rtz, cf}); }
    { std::vector<std::vector<int>> g; for (int jj = 0; jj < 4; ++jj) { std::vector<int> v; for (int h = 0; h < 4; ++h) for (int j = 2*jj; j < 2*jj+2; ++j) v.push_back(8 * h + j); g.push_back(v); }
      for (int rtz : {0, 1, 2, 3}) for (int cf : {0, 1}) models.push_back({"g8 by j-pairs", g, rtz, cf}); }
    std::vector<long> match(models.size(), 0); long total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        float sc = (trial % 4 == 0) ? 1.f : (trial % 4 == 1 ? 100.f : (trial % 4 == 2 ? 0.01f : 1.f));
        for (int i = 0; i < 512; ++i) { hA[i] = f2bf(rnd(1.f) * (rand() % 7 == 0 ? 50.f : 1.f)); hB[i] = f2bf(rnd(1.f)); }
        for (int i = 0; i < 256; ++i) hC[i] = (trial % 3 == 0) ? 0.f : rnd(sc);
        (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float c = hC[i * 16 + j]; long double p[32];
            for (int k = 0; k < 32; ++k) p[k] = (long double)bf2f(hA[i * 32 + k]) * (long double)bf2f(hB[j * 32 + k]);
            for (size_t q = 0; q < models.size(); ++q) {
                const Model& M = models[q];
                float acc = M.cfirst ? c : 0.f;
                for (auto& g : M.groups) { long double s = acc; for (int k : g) s += p[k]; acc = to_f32(s, M.rtz); }
                if (!M.cfirst) acc = to_f32((long double)acc + (long double)c, M.rtz);
                match[q] += (memcmp(&acc, &hD[i * 16 + j], 4) == 0);
            }
            ++total;
        }
    }
    for (size_t q = 0; q < models.size(); ++q)
        printf("%-26s groups=%2zu rtz=%d cfirst=%d  %.3f%%\n", models[q].name, models[q].groups.size(), models[q].rtz, models[q].cfirst, 100.0 * match[q] / total);
    return 0;
}
