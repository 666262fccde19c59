// Same-wave interleave: each wave runs {1 MFMA + NV independent v_fma_f32} per iteration.  Does VALU hide under MFMA?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define ITER 4096
template <int MODE, bool BF16, int NV>   // MODE bit0: mfma, bit1: valu
__global__ __launch_bounds__(256) void k(float* out) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    const float c = 1.0000001f, d = 0.5f;
    for (int i = 0; i < ITER; ++i) {
        if (MODE & 1) {
            if (BF16) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c0, 0, 0, 0); }
            else { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); }
        }
        if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], c, d);
        }
        if (MODE & 1) {
            if (BF16) { c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c1, 0, 0, 0); }
            else { c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0); }
        }
        if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], c, d);
        }
    }
    float acc = c0[0] + c1[1];
    for (int j = 0; j < 8; ++j) acc += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE, bool BF16, int NV>
float run(float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, BF16, NV>), dim3(256), dim3(256), 0, 0, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BF16, NV>), dim3(256), dim3(256), 0, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <bool BF16, int NV> void row(float* out) {
    printf("%s mfma + %2d fma per mfma (1 wave/SIMD): mfma only %.3f  valu only %.3f  both %.3f ms\n", BF16 ? "bf16 16x16x32" : "f32  16x16x4 ",
           NV, run<1, BF16, NV>(out), run<2, BF16, NV>(out), run<3, BF16, NV>(out));
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    row<false, 4>(out); row<false, 8>(out); row<false, 16>(out);
    row<true, 2>(out); row<true, 4>(out); row<true, 8>(out); row<true, 16>(out);
    return 0;
}
