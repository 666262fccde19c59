// r03: two questions about pass 2's tile loop.
//  (1) Does v_pk_mul_f32 with the VOP3P clamp bit saturate both halves to [0, 1] (+inf -> 1, negative / -inf -> 0, NaN -> 0), and what does
//      it cost against the two v_med3_f32 it would replace?
//  (2) Is v_mfma_f32_16x16x16_bf16 (K = 16, two operand registers) cheaper to issue than v_mfma_f32_16x16x32_bf16 -- alone and between VALU
//      work -- so that the half-empty second MFMA of R = P Q^T could shrink?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_clamp_mfma16.hip -o /tmp/ubcm && /tmp/ubcm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#define ITER 4096
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k_sem(const float* in, float* out, float b) {
    f32x2 v = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]}, r, bb = {b, b};
    asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(v), "v"(bb));
    out[2 * threadIdx.x] = r.x; out[2 * threadIdx.x + 1] = r.y;
}

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define VKERNEL(NAME, BODY)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                       \
        f32x2 a[8]; f32x2 c = {1.0000001f, 0.9999999f};                                                        \
        for (int j = 0; j < 8; ++j) a[j] = (f32x2){threadIdx.x * 0.001f + j + 1.5f, threadIdx.x * 0.002f + j}; \
        asm volatile("" : "+v"(c));                                                                            \
        for (int i = 0; i < iters; ++i) { REP32(BODY) }                                                        \
        float acc = 0;                                                                                         \
        for (int j = 0; j < 8; ++j) acc += a[j].x + a[j].y;                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                             \
    }
#define B_PKMUL(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
#define B_PKMULC(j) asm volatile("v_pk_mul_f32 %0, %0, %1 clamp" : "+v"(a[j]) : "v"(c));
#define B_PKMULS(j) asm volatile("v_pk_mul_f32 %0, %0, s[20:21] clamp" : "+v"(a[j]) : : "s20", "s21");
#define B_MED3(j) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(a[j].x) : "v"(c.x));
VKERNEL(k_pkmul, B_PKMUL) VKERNEL(k_pkmulc, B_PKMULC) VKERNEL(k_pkmuls, B_PKMULS) VKERNEL(k_med3, B_MED3)

// MFMA kernels: NV fast-class VALU instructions between MFMAs; 4 independent accumulators
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    f32x4 acc[4]; s16x8 a8, b8; s16x4 a4, b4; float v[8];
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 8; ++j) { a8[j] = (short)(0x3f80 + threadIdx.x + j); b8[j] = (short)(0x3f00 + j); v[j] = threadIdx.x + j; }
    for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
    float c = 1.0000001f; asm volatile("" : "+v"(c));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 32) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a8), "v"(b8));
            else            asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a4), "v"(b4));
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(c));
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef void (*kern_t)(float*, int);
static float time_kernel(kern_t fn, float* out, int wps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fn, dim3(256 * wps), dim3(256), 0, 0, out, ITER);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(256 * wps), dim3(256), 0, 0, out, ITER);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
static void runv(const char* name, kern_t fn, float* out, int wps) {
    const float ms = time_kernel(fn, out, wps);
    printf("%-44s wps=%d  %7.3f ms  %6.2f cycles per instruction (2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ITER / wps / 32.0);
}
template <int KIND, int NV> static void runm(float* out, int wps) {
    const float ms = time_kernel(k_mfma<KIND, NV>, out, wps);
    const double per = ms * 1e-3 * 2.4e9 / ITER / wps / 8.0;            // cycles per (MFMA + NV fma) group per wave
    printf("16x16x%-2d bf16 + %2d v_fma_f32 per MFMA           wps=%d  %7.3f ms  %6.2f cycles per group  (MFMA share %6.2f at 2.95 per fma)\n", KIND, NV, wps, ms, per, per - 2.95 * NV);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    float h[128]; const float vals[16] = {INFINITY, -INFINITY, -3.f, 0.5f, 2.f, NAN, 1e-13f, 0.f, -0.f, 1.f, 0.999f, 1e30f, -1e-30f, 1e-45f, 3e12f, 7e11f};
    for (int i = 0; i < 128; ++i) h[i] = vals[i % 16];
    float *din, *dout; (void)hipMalloc(&din, 512); (void)hipMalloc(&dout, 512);
    (void)hipMemcpy(din, h, 512, hipMemcpyHostToDevice);
    for (float b : {1.0f, 1e-12f}) {
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, din, dout, b);
        float r[128]; (void)hipMemcpy(r, dout, 512, hipMemcpyDeviceToHost);
        printf("v_pk_mul_f32 clamp, second factor %g:\n", b);
        for (int i = 0; i < 16; ++i) printf("   %-12g -> %-12g\n", vals[i], r[i]);
    }
    for (int wps = 1; wps <= 3; wps += 2) {
        runv("v_pk_mul_f32 v, v, v", k_pkmul, out, wps); runv("v_pk_mul_f32 v, v, v clamp", k_pkmulc, out, wps);
        runv("v_pk_mul_f32 v, v, s[20:21] clamp", k_pkmuls, out, wps); runv("v_med3_f32 v, v, 0, v", k_med3, out, wps);
        runm<32, 0>(out, wps); runm<16, 0>(out, wps); runm<32, 4>(out, wps); runm<16, 4>(out, wps);
        runm<32, 8>(out, wps); runm<16, 8>(out, wps); runm<32, 14>(out, wps); runm<16, 14>(out, wps);
    }
    return 0;
}
