// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for plain f32 ops, packed f32 ops and
// transcendentals, dependent vs independent, at 1/2/4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_ITER 4096
template <int MODE>
__global__ void k(float* out, long long* cyc, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 1.0000001f, d = 0.5f;
    const f32x2 c2 = {c, c}, d2 = {d, d};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) {          // 8 independent v_fma_f32
            a0 = fmaf(a0, c, d); a1 = fmaf(a1, c, d); a2 = fmaf(a2, c, d); a3 = fmaf(a3, c, d);
            a4 = fmaf(a4, c, d); a5 = fmaf(a5, c, d); a6 = fmaf(a6, c, d); a7 = fmaf(a7, c, d);
        } else if (MODE == 1) {   // 8 dependent v_fma_f32
            a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d);
            a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d);
        } else if (MODE == 2) {   // 4 independent v_pk_fma_f32 (8 FMAs)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c2), "v"(d2));
        } else if (MODE == 3) {   // 8 independent v_rcp_f32
            a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
            a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
        } else if (MODE == 4) {   // 8 independent v_cndmask (cmp hoisted)
            bool m = a7 > 3.f;
            a0 = m ? a0 : a1; a1 = m ? a1 : a2; a2 = m ? a2 : a3; a3 = m ? a3 : a4;
            a4 = m ? a4 : a5; a5 = m ? a5 : a6; a6 = m ? a6 : a0; a7 = a7 + 1.f;
        } else if (MODE == 5) {   // 8 independent v_max_f32
            a0 = fmaxf(a0, c); a1 = fmaxf(a1, d); a2 = fmaxf(a2, c); a3 = fmaxf(a3, d);
            a4 = fmaxf(a4, c); a5 = fmaxf(a5, d); a6 = fmaxf(a6, c); a7 = fmaxf(a7, d);
            a0 += 1.f; a1 += 1.f; a2 += 1.f; a3 += 1.f; a4 += 1.f; a5 += 1.f; a6 += 1.f; a7 += 1.f;   // 16 ops total
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int ops_per_iter, float* out, long long* cyc) {
    for (int wpb : {256, 512, 1024}) {     // 1, 2, 4 waves per SIMD (one block per CU)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double winst = (double)N_ITER * ops_per_iter;                 // per wave
        double waves_per_simd = wpb / 256.0;
        printf("%-28s waves/SIMD %.0f: %.3f ms, s_memtime/instr %.2f, wall-cycles(2.4GHz)/instr/SIMD %.2f\n", name, waves_per_simd, ms,
               (double)c / winst, ms * 1e-3 * 2.4e9 / (winst * waves_per_simd));
    }
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32 indep x8", 8, out, cyc);
    run<1>("v_fma_f32 dependent x8", 8, out, cyc);
    run<2>("v_pk_fma_f32 indep x4", 4, out, cyc);
    run<3>("v_rcp_f32 indep x8", 8, out, cyc);
    run<4>("v_cndmask indep x8", 8, out, cyc);
    run<5>("v_max+v_add x16", 16, out, cyc);
    return 0;
}
