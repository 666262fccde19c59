// r03: is v_cndmask_b32 really a 23-cycle instruction on gfx950 -- and what do the integer multiplies of the row addressing cost (profiles/r02_ubench_valu_asm.txt)?  Independent destination registers,
// mask in VCC / in an SGPR pair, against v_bfi_b32 and v_and_b32 doing the same selection with a full-width mask register.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_cndmask.hip -o /tmp/ubc && /tmp/ubc
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define KERNEL(NAME, BODY)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                       \
        float a[8]; unsigned u[8]; unsigned long long w[8];                                                    \
        for (int j = 0; j < 8; ++j) { a[j] = threadIdx.x * 0.001f + j + 1.5f; u[j] = threadIdx.x * 977u + j; w[j] = u[j]; } \
        float c1 = 1.0000001f; unsigned mk = (threadIdx.x & 1) ? 0xFFFFFFFFu : 0u;                             \
        asm volatile("" : "+v"(c1), "+v"(mk));                                                                 \
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" : : "v"(a[0]), "v"(c1) : "vcc", "s20", "s21");  \
        for (int i = 0; i < iters; ++i) { REP32(BODY) }                                                        \
        float acc = 0;                                                                                         \
        for (int j = 0; j < 8; ++j) acc += a[j] + (float)u[j] + (float)w[j];                                   \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                             \
    }
#define B_CND_VCC(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(c1));
#define B_CND_SG(j) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[j]) : "v"(c1));
#define B_CND_0(j) asm volatile("v_cndmask_b32_e64 %0, 0, %0, s[20:21]" : "+v"(a[j]));
#define B_BFI(j) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[j]) : "v"(mk), "v"(c1));
#define B_AND(j) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[j]) : "v"(mk));
#define B_FMA(j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c1));
#define B_MULLO(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j]) : "v"(mk));
#define B_MUL24(j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[j]) : "v"(mk));
#define B_MULHI24(j) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(u[j]) : "v"(mk));
#define B_MAD64(j) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(w[j]) : "v"(u[j]), "v"(mk) : "s20", "s21");
#define B_LSHLADD64(j) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(w[j]) : "v"(w[(j + 1) & 7]));
KERNEL(k_cnd_vcc, B_CND_VCC) KERNEL(k_cnd_sg, B_CND_SG) KERNEL(k_cnd_0, B_CND_0) KERNEL(k_bfi, B_BFI) KERNEL(k_and, B_AND) KERNEL(k_fma, B_FMA) KERNEL(k_mullo, B_MULLO) KERNEL(k_mul24, B_MUL24) KERNEL(k_mulhi24, B_MULHI24) KERNEL(k_mad64, B_MAD64) KERNEL(k_lshladd64, B_LSHLADD64)
typedef void (*kern_t)(float*, int);
static void run(const char* name, kern_t fn, float* out, int wps) {
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
    printf("%-44s wps=%d  %7.3f ms  %6.2f cycles per instruction (2.4 GHz)\n", name, wps, best, best * 1e-3 * 2.4e9 / ITER / wps / 32.0);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int wps = 1; wps <= 3; wps += 2) {
        run("v_cndmask_b32 v, v, v, vcc", k_cnd_vcc, out, wps); run("v_cndmask_b32_e64 v, v, v, s[20:21]", k_cnd_sg, out, wps);
        run("v_cndmask_b32_e64 v, 0, v, s[20:21]", k_cnd_0, out, wps); run("v_bfi_b32 v, mask, v, v", k_bfi, out, wps);
        run("v_and_b32 v, mask, v", k_and, out, wps); run("v_fma_f32", k_fma, out, wps);
        run("v_mul_lo_u32", k_mullo, out, wps); run("v_mul_u32_u24", k_mul24, out, wps); run("v_mul_hi_u32_u24", k_mulhi24, out, wps);
        run("v_mad_u64_u32", k_mad64, out, wps); run("v_lshl_add_u64", k_lshladd64, out, wps);
    }
    return 0;
}
