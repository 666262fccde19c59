// r03: does the issue cost of a VALU instruction on gfx950 depend on the number of waves per SIMD?  (tools/ubench_valu_asm.hip
// measured everything at 3 waves per SIMD, the occupancy of pass 2.)  Each kernel = a loop of 32 inline-asm instructions of one
// form on 8 independent registers; the grid puts `wps` waves on every SIMD of the chip; cycles per instruction are derived from
// the wall time AND from the shader-clock counter read inside the kernel (s_memtime runs at a constant 100 MHz: the ratio of
// both gives the real shader clock under this load).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_occupancy.hip -o /tmp/ubo && /tmp/ubo
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define KERNEL(NAME, BODY)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                       \
        float a[8]; f32x2 p[8]; unsigned u[8];                                                                 \
        for (int j = 0; j < 8; ++j) { a[j] = threadIdx.x * 0.001f + j + 1.5f; p[j] = (f32x2){a[j], a[j] + 0.25f}; u[j] = threadIdx.x * 977u + j; } \
        float c1 = 1.0000001f, c2 = 0.5f;                                                                      \
        f32x2 q = {1.0000001f, 0.999999f};                                                                     \
        asm volatile("" : "+v"(c1), "+v"(c2), "+v"(q));                                                        \
        for (int i = 0; i < iters; ++i) { REP32(BODY) }                                                        \
        float acc = 0;                                                                                         \
        for (int j = 0; j < 8; ++j) acc += a[j] + p[j].x + p[j].y + (float)u[j];                               \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                             \
    }
#define B_FMA(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
#define B_PKFMA(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(q));
#define B_MAX3(j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
#define B_LOG(j) asm volatile("v_log_f32 %0, %0" : "+v"(a[j]));
#define B_CVTBF(j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
// the instruction mix of pass 2 per pair of genotypes: 11 v_pk, 2 max3, 2 rcp, 4 log, 3 cvt, 2 add, 2 shift/and
#define B_MIX(j) asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_add_f32 %0, %0, %2\n v_max3_f32 %1, %1, %3, %4\n v_rcp_f32 %1, %1\n" \
                              "v_pk_mul_f32 %0, %0, %2\n v_log_f32 %1, %1\n v_cvt_pk_bf16_f32 %1, %1, %3\n v_pk_fma_f32 %0, %0, %2, %2\n" \
                              "v_log_f32 %1, %1\n v_add_f32 %1, %1, %3\n v_pk_add_f32 %0, %0, %2\n v_and_b32 %1, 0xffff0000, %1" \
                              : "+v"(p[j]), "+v"(a[j]) : "v"(q), "v"(c1), "v"(c2));
KERNEL(k_fma, B_FMA) KERNEL(k_pkfma, B_PKFMA) KERNEL(k_max3, B_MAX3) KERNEL(k_log, B_LOG) KERNEL(k_cvtbf, B_CVTBF) KERNEL(k_mix, B_MIX)

// MFMA + VALU mix: 2 MFMAs (independent accumulators) + 24 v_pk_fma per iteration
__global__ __launch_bounds__(256) void k_mfma_mix(float* out, int iters) {
    f32x2 p[8]; f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    for (int j = 0; j < 8; ++j) p[j] = (f32x2){threadIdx.x * 0.001f + j, 1.f};
    f32x2 q = {1.0000001f, 0.999999f};
    bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
    asm volatile("" : "+v"(q), "+v"(av), "+v"(bv));
    for (int i = 0; i < iters; ++i) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, d1, 0, 0, 0);
#define PK(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(q));
        REP8(PK) REP8(PK) REP8(PK)
    }
    float acc = d0[0] + d1[1];
    for (int j = 0; j < 8; ++j) acc += p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kern_t)(float*, int);
static void run(const char* name, kern_t fn, float* out, int wps, int insts_per_iter, double ghz) {
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
    printf("%-34s wps=%d  %7.3f ms  %6.2f cycles/inst (at %.2f GHz)  %6.2f ns/inst/SIMD\n", name, wps, best,
           best * 1e-3 * ghz * 1e9 / ITER / wps / insts_per_iter, ghz, best * 1e6 / ITER / wps / insts_per_iter);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 16 * 256 * 4);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk * 1e-6;
    printf("clock attribute %.3f GHz\n", ghz);
    const int wlist[] = {1, 2, 3, 4, 5, 6, 8};
    for (int wi = 0; wi < 7; ++wi) {
        const int wps = wlist[wi];
        run("v_fma_f32", k_fma, out, wps, 32, ghz);
        run("v_pk_fma_f32", k_pkfma, out, wps, 32, ghz);
        run("v_max3_f32", k_max3, out, wps, 32, ghz);
        run("v_log_f32", k_log, out, wps, 32, ghz);
        run("v_cvt_pk_bf16_f32", k_cvtbf, out, wps, 32, ghz);
        run("pass-2 mix (12 inst x 8)", k_mix, out, wps, 96 * 4, ghz);
        run("2 mfma + 24 v_pk_fma", k_mfma_mix, out, wps, 26, ghz);
    }
    return 0;
}
