// Do f32-input MFMA and f32 VALU overlap on one SIMD?  Block = 8 waves (2 per SIMD): waves 0-3 run a loop of
// independent v_mfma_f32_16x16x4_f32 (or bf16 16x16x32), waves 4-7 a loop of independent v_fma_f32.
// Compare t(both) with t(mfma only) and t(valu only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define ITER 2048
template <int MODE, bool BF16>   // MODE bit0: mfma waves active, bit1: valu waves active
__global__ __launch_bounds__(512) void k(float* out) {
    const int wave = threadIdx.x >> 6;
    float acc = 0.f;
    if (wave < 4) {
        if (MODE & 1) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            const float a = threadIdx.x * 1e-3f, b = 1.0001f;
            const bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
            for (int i = 0; i < ITER; ++i) {
                if (BF16) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c3, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                }
            }
            acc = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (MODE & 2) {
            float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
            const float c = 1.0000001f, d = 0.5f;
            for (int i = 0; i < ITER * 8; ++i) {     // 8 fma per iteration, sized to take about as long as the mfma loop
                a0 = fmaf(a0, c, d); a1 = fmaf(a1, c, d); a2 = fmaf(a2, c, d); a3 = fmaf(a3, c, d);
                a4 = fmaf(a4, c, d); a5 = fmaf(a5, c, d); a6 = fmaf(a6, c, d); a7 = fmaf(a7, c, d);
            }
            acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int MODE, bool BF16>
float run(float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, BF16>), dim3(256), dim3(512), 0, 0, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BF16>), dim3(256), dim3(512), 0, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    printf("f32 mfma 16x16x4 : mfma only %.3f ms, valu only %.3f ms, both %.3f ms\n", run<1, false>(out), run<2, false>(out), run<3, false>(out));
    printf("bf16 mfma 16x16x32: mfma only %.3f ms, valu only %.3f ms, both %.3f ms\n", run<1, true>(out), run<2, true>(out), run<3, true>(out));
    return 0;
}
