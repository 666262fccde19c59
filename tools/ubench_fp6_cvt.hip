// r03: v_cvt_scalef32_2xpk16_fp6_f32 (32 floats -> 32 FP6 E2M3 in 6 registers) and v_cvt_scalef32_pk32_f32_fp6 (back): element order,
// direction of the scale, rounding, saturation, and what they cost.  Used to cut V into FP6 pieces for the FP4 x FP6 matrix instruction
// (tools/ubench_fp4_mfma.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fp6_cvt.hip -o tools/bin/ubfp6 && tools/bin/ubfp6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
__global__ void k(const float* in, unsigned* out, float* back, float scale) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
    f32x32 f = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r, scale);
    for (int i = 0; i < 32; ++i) back[threadIdx.x * 32 + i] = f[i];
}
__global__ __launch_bounds__(256) void k_time(float* out, int iters, float scale) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.01f + i; b[i] = threadIdx.x * 0.02f - i; }
    u32x6 acc = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
            asm volatile("" : "+v"(r));
            f32x32 f = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r, scale);
            asm volatile("" : "+v"(f));
            a[m] += f[m]; acc ^= r;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[7] + (float)(acc[0] ^ acc[5]);
}
static float dec_fp6(int c) { const int e = (c >> 3) & 3, m = c & 7; const float s = c & 32 ? -1.f : 1.f; return s * (e == 0 ? m * 0.125f : ldexpf(1.f + m * 0.125f, e - 1)); }
static int get6(const unsigned* regs, int e) { int c = 0; for (int k = 0; k < 6; ++k) if (regs[(6 * e + k) >> 5] >> ((6 * e + k) & 31) & 1) c |= 1 << k; return c; }
int main() {
    float h[64]; unsigned o[12]; float bk[64];
    float *din, *dback; unsigned* dout;
    (void)hipMalloc(&din, 256); (void)hipMalloc(&dout, 48); (void)hipMalloc(&dback, 256);
    auto run = [&](float scale) {
        (void)hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(2), 0, 0, din, dout, dback, scale);
        (void)hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost); (void)hipMemcpy(bk, dback, 256, hipMemcpyDeviceToHost);
    };
    // element order: input e carries the e-th positive FP6 value (e = 31: -0.125)
    for (int e = 0; e < 32; ++e) h[e] = h[32 + e] = dec_fp6(e < 31 ? e + 1 : 0x21);
    run(1.0f);
    int bad = 0, badb = 0;
    for (int e = 0; e < 32; ++e) { if (get6(o, e) != (e < 31 ? e + 1 : 0x21)) { if (bad < 6) printf("  in[%d] -> field %d holds code 0x%x\n", e, e, get6(o, e)); ++bad; } if (bk[e] != h[e]) ++badb; }
    printf("order: %d of 32 fields differ from (a[0..15], b[0..15]) -> fields 0..31 little-endian; round trip differs in %d\n", bad, badb);
    // scale direction, rounding, saturation
    const float tv[16] = {3.0f, 0.3f, 0.0625f, 0.06f, 0.07f, 0.1875f, 4.25f, 4.75f, 7.6f, 7.8f, 100.f, -100.f, 1.0625f, 1.1875f, 1e-30f, 5.25f};
    for (int e = 0; e < 32; ++e) h[e] = tv[e & 15];
    for (float sc : {1.0f, 2.0f, 0.25f, 3.0f}) {
        run(sc);
        printf("scale %g:", sc);
        for (int e = 0; e < 16; ++e) printf("  %g->%g(back %g)", tv[e], dec_fp6(get6(o, e)), bk[e]);
        printf("\n");
    }
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps += 3) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_time, dim3(256 * wps), dim3(256), 0, 0, out, 1024, 1.0f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("cvt to FP6 + cvt back + 2 VALU, wps=%d: %.3f ms  %.1f cycles per pair of conversions (2.4 GHz)\n", wps, best, best * 1e-3 * 2.4e9 / 1024 / wps / 8.0);
    }
    return 0;
}
