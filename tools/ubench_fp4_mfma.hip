// r03: can the 2-bit genotypes feed the matrix pipe without a conversion?  A nibble 00cc IS the FP4 (E2M1) number c/2, and gfx950's
// v_mfma_scale_f32_16x16x128_f8f6f4 takes FP4 / FP6 / FP8 operands with one E8M0 scale per lane (= per 32 K-elements of a row / column).
// This probe (1) finds the operand layout -- which lane and which bit field holds element k of A (FP4) and of B (FP6 E2M3 / FP8 E4M3), and
// what the scale operands do -- by one-hot experiments, (2) checks a random product against a host model of that layout, and (3) times the
// instruction alone and between VALU work for the operand formats of interest.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fp4_mfma.hip -o tools/bin/ubfp4 && tools/bin/ubfp4
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITER 2048

// one wave: D = A . B with per-lane scales; a[64][8], b[64][8] registers, sa[64], sb[64] scale words (byte 0 used)
template <int FA, int FB>
__global__ void k_one(const int* a, const int* b, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    for (int r = 0; r < 8; ++r) { av[r] = a[l * 8 + r]; bv[r] = b[l * 8 + r]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, FA, FB, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

static float dec_fp4(int c) { static const float t[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f}; return (c & 8 ? -1.f : 1.f) * t[c & 7]; }
static float dec_fp6(int c) {                      // E2M3, bias 1
    const int e = (c >> 3) & 3, m = c & 7; const float s = c & 32 ? -1.f : 1.f;
    return s * (e == 0 ? m * 0.125f : ldexpf(1.f + m * 0.125f, e - 1));
}
static float dec_fp8(int c) {                      // E4M3 (OCP), bias 7
    const int e = (c >> 3) & 15, m = c & 7; const float s = c & 128 ? -1.f : 1.f;
    if (e == 15 && m == 7) return NAN;
    return s * (e == 0 ? ldexpf(m * 0.125f, -6) : ldexpf(1.f + m * 0.125f, e - 7));
}
static int bits_of(int fmt) { return fmt == 4 ? 4 : (fmt >= 2 ? 6 : 8); }
static float dec(int fmt, int c) { return fmt == 4 ? dec_fp4(c) : (fmt >= 2 ? dec_fp6(c) : dec_fp8(c)); }
// assumed layout: element e (0..31) of a lane at bits [w*e, w*e + w) of the lane's registers read as one little-endian bit string
static void put(int* regs, int fmt, int e, int code) {
    const int w = bits_of(fmt), bit = w * e;
    for (int k = 0; k < w; ++k) if (code >> k & 1) regs[(bit + k) >> 5] |= 1 << ((bit + k) & 31);
}
static int get(const int* regs, int fmt, int e) {
    const int w = bits_of(fmt), bit = w * e; int c = 0;
    for (int k = 0; k < w; ++k) if (regs[(bit + k) >> 5] >> ((bit + k) & 31) & 1) c |= 1 << k;
    return c;
}
template <int FA, int FB> static void run_one(const int* a, const int* b, const int* sa, const int* sb, float* d) {
    int *da, *db, *dsa, *dsb; float* dd;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dd, 1024);
    (void)hipMemcpy(da, a, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, b, 2048, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_one<FA, FB>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    (void)hipMemcpy(d, dd, 1024, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dsa); (void)hipFree(dsb); (void)hipFree(dd);
}
// host model: A[i][k]: lane i + 16 (k / 32), element k % 32; B[k][j]: lane j + 16 (k / 32), element k % 32; D[4 (l >> 4) + r][l & 15] in lane l, register r
template <int FA, int FB> static double check_random(unsigned seed, bool scales) {
    srand(seed);
    int a[512] = {0}, b[512] = {0}, sa[64], sb[64]; float d[256];
    for (int l = 0; l < 64; ++l) {
        for (int e = 0; e < 32; ++e) {
            int ca = rand() & ((1 << bits_of(FA)) - 1), cb = rand() & ((1 << bits_of(FB)) - 1);
            if (FA < 2 && (ca & 0x7f) == 0x7f) ca = 0;
            if (FB < 2 && (cb & 0x7f) == 0x7f) cb = 0;
            put(a + 8 * l, FA, e, ca); put(b + 8 * l, FB, e, cb);
        }
        sa[l] = scales ? 127 + (rand() % 9) - 4 : 127; sb[l] = scales ? 127 + (rand() % 9) - 4 : 127;
        sa[l] |= 0x55aa3300; sb[l] |= 0x7f7f7f00;           // the other bytes must not matter (opsel = 0)
    }
    run_one<FA, FB>(a, b, sa, sb, d);
    double worst = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0, mag = 0;
        for (int k = 0; k < 128; ++k) {
            const int la = i + 16 * (k / 32), lb = j + 16 * (k / 32);
            const double t = (double)dec(FA, get(a + 8 * la, FA, k % 32)) * ldexp(1.0, (sa[la] & 255) - 127) * dec(FB, get(b + 8 * lb, FB, k % 32)) * ldexp(1.0, (sb[lb] & 255) - 127);
            ref += t; mag += fabs(t);
        }
        const float got = d[(j + 16 * (i / 4)) * 4 + (i & 3)];
        const double err = fabs(got - ref) / (mag + 1e-30);
        if (err > worst) worst = err;
    }
    return worst;
}
// one-hot: A element (qa, ea) = 1.0 in every row; B element (qb, eb) carries a value that names (qb, eb): which B element meets it?
template <int FB> static void one_hot() {
    int bad = 0;
    for (int qa = 0; qa < 4; ++qa) for (int ea = 0; ea < 32; ++ea) {
        int a[512] = {0}, b[512] = {0}, sa[64], sb[64]; float d[256];
        for (int l = 16 * qa; l < 16 * qa + 16; ++l) put(a + 8 * l, 4, ea, 2);
        for (int l = 0; l < 64; ++l) {
            sa[l] = 127; sb[l] = 127 + 8 * (l >> 4);                         // B's K-block q scaled by 2^(8q)
            for (int e = 0; e < 32; ++e) put(b + 8 * l, FB, e, e < 31 ? e + 1 : (FB >= 2 ? 0x21 : 0x81));   // element e -> a value naming e
        }
        run_one<4, FB>(a, b, sa, sb, d);
        const float got = d[0];
        const float want = dec(FB, ea < 31 ? ea + 1 : (FB >= 2 ? 0x21 : 0x81)) * ldexpf(1.f, 8 * qa);
        if (got != want) { if (bad < 8) printf("   A(q=%d, e=%d): D = %g, identity layout predicts %g\n", qa, ea, got, want); ++bad; }
    }
    printf("one-hot, A FP4 x B %s: %d of 128 positions differ from the identity K mapping (lane block q <-> q, element e <-> e)\n", FB >= 2 ? "FP6" : "FP8", bad);
}

template <int FA, int FB, int NV>
__global__ __launch_bounds__(256) void k_time(float* out, int iters) {
    f32x4 acc[4]; i32x8 a, b; float v[8];
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 8; ++j) { a[j] = 0x22222222 ^ (threadIdx.x * 0x01010101 & 0x11111111); b[j] = 0x08080808 + j; v[j] = threadIdx.x + j; }
    float c = 1.0000001f; int s = 127; asm volatile("" : "+v"(c), "+v"(s));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[m & 3], FA, FB, 0, s, 0, s);
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(c));
        }
    }
    float t = 0;
    for (int j = 0; j < 4; ++j) t += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    for (int j = 0; j < 8; ++j) t += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int FA, int FB, int NV> static void timeit(float* out, int wps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_time<FA, FB, NV>), dim3(256 * wps), dim3(256), 0, 0, out, ITER);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_time<FA, FB, NV>), dim3(256 * wps), dim3(256), 0, 0, out, ITER);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double per = best * 1e-3 * 2.4e9 / ITER / wps / 8.0;
    printf("16x16x128 A fmt %d x B fmt %d + %2d v_fma_f32 per MFMA   wps=%d  %7.3f ms  %6.2f cycles per group  (MFMA share %6.2f at 2.95 per fma)\n", FA, FB, NV, wps, best, per, per - 2.95 * NV);
}
int main() {
    one_hot<0>(); one_hot<2>();
    printf("random product vs host model, worst |err| / sum|terms|:  FP4 x FP8 %.2e (scales 1) %.2e (random scales)   FP4 x FP6 %.2e / %.2e   FP4 x FP4 %.2e / %.2e\n",
           check_random<4, 0>(1, false), check_random<4, 0>(2, true), check_random<4, 2>(3, false), check_random<4, 2>(4, true), check_random<4, 4>(5, false), check_random<4, 4>(6, true));
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int wps = 3; wps <= 4; ++wps) {
        timeit<4, 4, 0>(out, wps); timeit<4, 2, 0>(out, wps); timeit<4, 0, 0>(out, wps); timeit<0, 0, 0>(out, wps);
        timeit<4, 2, 4>(out, wps); timeit<4, 0, 4>(out, wps); timeit<4, 2, 8>(out, wps); timeit<4, 0, 8>(out, wps); timeit<4, 2, 14>(out, wps); timeit<4, 0, 14>(out, wps);
    }
    return 0;
}
