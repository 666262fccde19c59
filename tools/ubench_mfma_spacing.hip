// r03: what does a v_mfma_f32_16x16x32_bf16 cost a SIMD that is otherwise saturated with VALU work, as a function of WHERE the MFMAs
// sit in the instruction stream?  Every kernel issues 24 v_pk_fma_f32 + N MFMAs per loop iteration (inline asm, order pinned):
//   dep2_b2b    2 MFMAs on the same accumulator, back to back            (how pass 2 writes its R / dQ / dP pairs)
//   ind2_b2b    2 MFMAs on different accumulators, back to back
//   ind2_spaced MFMA, 12 v_pk_fma, MFMA, 12 v_pk_fma
//   dep2_spaced the same with one accumulator
//   one         1 MFMA + 24 v_pk_fma
//   none        24 v_pk_fma
//   ind4_spaced 4 MFMAs (2 accumulators) each followed by 6 v_pk_fma
// at 1 / 2 / 3 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_spacing.hip -o /tmp/ubm && /tmp/ubm
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PK(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(q));
#define PK6 PK(0) PK(1) PK(2) PK(3) PK(4) PK(5)
#define PK12 PK6 PK(6) PK(7) PK(0) PK(1) PK(2) PK(3)
#define PK24 PK12 PK(4) PK(5) PK(6) PK(7) PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7)
#define MF(d) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(av), "v"(bv));
#define KERNEL(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                            \
        f32x2 p[8]; f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};                                     \
        for (int j = 0; j < 8; ++j) p[j] = (f32x2){threadIdx.x * 0.001f + j, 1.f};                  \
        f32x2 q = {1.0000001f, 0.999999f};                                                          \
        u32x4 av = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bv = av;                   \
        asm volatile("" : "+v"(q), "+v"(av), "+v"(bv));                                             \
        for (int i = 0; i < iters; ++i) { BODY }                                                    \
        float acc = d0[0] + d1[1];                                                                  \
        for (int j = 0; j < 8; ++j) acc += p[j].x + p[j].y;                                         \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                  \
    }
KERNEL(k_none, PK24)
KERNEL(k_one, MF(d0) PK24)
KERNEL(k_dep2_b2b, MF(d0) MF(d0) PK24)
KERNEL(k_ind2_b2b, MF(d0) MF(d1) PK24)
KERNEL(k_ind2_spaced, MF(d0) PK12 MF(d1) PK12)
KERNEL(k_dep2_spaced, MF(d0) PK12 MF(d0) PK12)
KERNEL(k_ind4_spaced, MF(d0) PK6 MF(d1) PK6 MF(d0) PK6 MF(d1) PK6)
KERNEL(k_ind4_b2b, MF(d0) MF(d1) MF(d0) MF(d1) PK24)
KERNEL(k_mfma_only4, MF(d0) MF(d1) MF(d0) MF(d1))

typedef void (*kern_t)(float*, int);
static double run(const char* name, kern_t fn, float* out, int wps, int n_mfma, double base_ns) {
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
    const double ns_iter = best * 1e6 / ITER / wps;          // SIMD time per loop iteration of one wave
    printf("%-14s wps=%d  %7.3f ms  %7.1f ns/iter/wave", name, wps, best, ns_iter);
    if (n_mfma && base_ns > 0) printf("   -> %6.1f ns = %5.1f cycles(2.4 GHz) per MFMA on top of the VALU work", (ns_iter - base_ns) / n_mfma, (ns_iter - base_ns) / n_mfma * 2.4);
    printf("\n");
    return ns_iter;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 16 * 256 * 4);
    for (int wps = 1; wps <= 3; ++wps) {
        const double base = run("none", k_none, out, wps, 0, 0);
        run("one", k_one, out, wps, 1, base);
        run("dep2_b2b", k_dep2_b2b, out, wps, 2, base);
        run("ind2_b2b", k_ind2_b2b, out, wps, 2, base);
        run("ind2_spaced", k_ind2_spaced, out, wps, 2, base);
        run("dep2_spaced", k_dep2_spaced, out, wps, 2, base);
        run("ind4_spaced", k_ind4_spaced, out, wps, 4, base);
        run("ind4_b2b", k_ind4_b2b, out, wps, 4, base);
        run("mfma_only4", k_mfma_only4, out, wps, 4, 1e-9);
    }
    return 0;
}
