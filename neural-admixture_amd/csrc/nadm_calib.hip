// Box fingerprint for measurements (bench.py "box"; include/nadm.h, measurement helpers).  NOT on the training path.
//
// One box of the pool sustains another shader clock than the next under the same load (r05: the driver's bench landed on a box
// whose pass 2 ran 9.6 % slower than the round's own collections, every memory-bound kernel of the step unchanged), and nothing
// in the bench line could say so.  This kernel runs a FIXED instruction stream of pass 2's kind -- packed-f32 VALU chains with
// a 16x16x32 bf16 matrix instruction in between, three waves per SIMD on every CU -- and reads two counters around it in
// every block: s_memtime (ticks = shader cycles, MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units") and s_memrealtime (a
// constant-rate clock, hipDeviceAttributeWallClockRate).  cycles / constant ticks x rate = the shader clock the box sustained
// under that load; the launch's duration for the fixed work is the second half of the fingerprint.
#include "nadm_host.h"
#include <stdint.h>

namespace nadm {
typedef float cal_f32x2 __attribute__((ext_vector_type(2)));
typedef float cal_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cal_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void calib_clock_kernel(int iters, uint64_t* __restrict__ out,
                                                                                                     float* __restrict__ sink) {
    const uint64_t c0 = __builtin_readcyclecounter();          // s_memtime
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const float s = 1.0f + 1e-7f * (float)threadIdx.x;
    cal_f32x2 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (cal_f32x2){s + j, s - j};
    const cal_f32x2 m = {0.999999f, 1.000001f}, c = {1e-6f, -1e-6f};
    cal_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    cal_bf16x8 A, B;
#pragma unroll
    for (int j = 0; j < 8; ++j) { A[j] = (__bf16)(0.5f + j); B[j] = (__bf16)(1.0f / (1 + j)); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_elementwise_fma(a[j], m, c);       // 8 independent v_pk_fma_f32 chains
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(a[j]));                      // keep the stream as written
    }
    float t = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) t += a[j].x + a[j].y;
    const uint64_t c1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (t == 12345.678f) sink[0] = t;                                                    // never true: keeps the arithmetic alive
}
}  // namespace nadm

extern "C" int64_t nadm_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}

extern "C" int nadm_calib_clock(int32_t iters, uint64_t* out, int32_t max_blocks, float* sink, void* stream) {
    using namespace nadm;
    if (iters < 1 || out == nullptr || sink == nullptr || max_blocks < 1) return fail("nadm_calib_clock: bad arguments");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        return fail("nadm_calib_clock: no device");
    const int blocks = 3 * cus < max_blocks ? 3 * cus : max_blocks;                     // 3 blocks x 4 waves per CU = 3 waves per SIMD
    hipLaunchKernelGGL(calib_clock_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out, sink);
    const int rc = check_launch("calib_clock");
    return rc ? -rc : blocks;                                                           // > 0: the number of blocks that report
}
