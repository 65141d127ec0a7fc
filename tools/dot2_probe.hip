// Rounding behaviour of v_dot2c_f32_bf16 on gfx950 (could it replace unpack + two FMAs in the landmark scan without changing
// a bit of the arithmetic contract?).  Compares the instruction with four host models on random operands.
//   hipcc --offload-arch=gfx950 -O2 tools/dot2_probe.hip -o /tmp/dot2_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, const float* c, float* d, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a[i]), __builtin_bit_cast(bf2, b[i]), c[i], false);
}
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t rnd_bf(int spread) {   // random bf16 with exponent in a window, random sign
    uint16_t e = 127 - spread / 2 + rand() % (spread + 1);
    return (uint16_t)((rand() & 1) << 15 | e << 7 | (rand() & 127));
}
int main(int argc, char** argv) {
    const int n = 1 << 20, spread = argc > 1 ? atoi(argv[1]) : 6;
    std::vector<uint32_t> a(n), b(n); std::vector<float> c(n), d(n);
    srand(5);
    for (int i = 0; i < n; ++i) {
        a[i] = rnd_bf(spread) | (uint32_t)rnd_bf(spread) << 16;
        b[i] = rnd_bf(spread) | (uint32_t)rnd_bf(spread) << 16;
        float m = (float)(rand() % (1 << 24)) / (1 << 20) - 8.f;
        c[i] = (i % 7 == 0) ? 0.f : ldexpf(m, rand() % (spread + 1) - spread / 2);
    }
    uint32_t *da, *db; float *dc, *dd;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dc, dd, n);
    hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    long mA = 0, mB = 0, mC = 0, mD = 0, mE = 0;
    int shown = 0;
    for (int i = 0; i < n; ++i) {
        const float a0 = bf(a[i] & 0xffff), a1 = bf(a[i] >> 16), b0 = bf(b[i] & 0xffff), b1 = bf(b[i] >> 16);
        const float A = fmaf(a1, b1, fmaf(a0, b0, c[i]));                 // element 0 first, then element 1
        const float B = fmaf(a0, b0, fmaf(a1, b1, c[i]));                 // element 1 first
        const float C = (float)((double)a0 * b0 + (double)a1 * b1 + (double)c[i]);   // one rounding (double is exact here for small spreads)
        const float D = (a0 * b0 + a1 * b1) + c[i];                       // products summed in f32 (a0*b0 exact, sum rounded), then + c
        const float E = fmaf(a0, b0, a1 * b1) + c[i];
        mA += A == d[i]; mB += B == d[i]; mC += C == d[i]; mD += D == d[i]; mE += E == d[i];
        if (A != d[i] && shown < 4) { printf("  ex: a=(%g,%g) b=(%g,%g) c=%g  gpu=%.9g A=%.9g B=%.9g C=%.9g\n", a0, a1, b0, b1, c[i], d[i], A, B, C); ++shown; }
    }
    printf("spread %d: n=%d  match A(fma e0 then e1)=%ld  B(fma e1 then e0)=%ld  C(single rounding)=%ld  D=%ld  E=%ld\n", spread, n, mA, mB, mC, mD, mE);
    return 0;
}
