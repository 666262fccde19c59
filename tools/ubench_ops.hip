// Price list of the VALU instructions the pass-2 kernel uses: cycles per wave64 instruction (s_memtime), 8 independent
// registers, compiler-scheduled straight-line code (no asm statement boundaries), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float c, float d, int sh) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + 0.1f + 0.01f * j;
    unsigned u[8];
    for (int j = 0; j < 8; ++j) u[j] = threadIdx.x * 2654435761u + j;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) v[j] = fmaf(v[j], c, d);
            if (MODE == 1) v[j] = v[j] * c;
            if (MODE == 2) v[j] = v[j] + d;
            if (MODE == 3) v[j] = fmaxf(v[j], d);
            if (MODE == 4) v[j] = __builtin_amdgcn_fmed3f(v[j], c, d);
            if (MODE == 5) v[j] = __builtin_amdgcn_rcpf(v[j]);
            if (MODE == 6) v[j] = __builtin_amdgcn_logf(v[j]);
            if (MODE == 7) v[j] = (v[j] == c) ? d : v[j];                      // v_cmp + v_cndmask
            if (MODE == 8) u[j] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){__uint_as_float(u[j]), c}, bf16x2));
            if (MODE == 9) u[j] = (u[j] >> sh) & 3u;                          // v_bfe_u32 (sh runtime)
            if (MODE == 10) v[j] = (float)(__float_as_uint(v[j]) & 0xffu);    // v_cvt_f32_ubyte0
            if (MODE == 11) u[j] = u[j] << sh;
            if (MODE == 12) v[j] = 1.0f - v[j];
            if (MODE == 13) { f32x2 r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(u[j], 1.0f, 1); u[j] = __float_as_uint(r.x) + __float_as_uint(r.y); }   // + 1 v_add
            if (MODE == 14) u[j] = __builtin_amdgcn_perm(u[j], u[(j + 1) & 7], 0x07060302u);
            if (MODE == 15) u[j] = u[j] + (unsigned)sh;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int j = 0; j < 8; ++j) acc += v[j] + __uint_as_float(u[j]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int ninstr, float* out, long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0000001f, 0.5f, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0000001f, 0.5f, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s s_memtime %.2f /slot ; wall %.3f ms = %.2f cycles@2.4GHz /slot (8 slots per iteration)\n", name, (double)c / ((double)ITER * ninstr), ms, ms * 1e-3 * 2.4e9 / ((double)ITER * ninstr));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", 8, out, cyc); run<1>("v_mul_f32", 8, out, cyc); run<2>("v_add_f32", 8, out, cyc); run<3>("v_max_f32", 8, out, cyc);
    run<4>("v_med3_f32", 8, out, cyc); run<5>("v_rcp_f32", 8, out, cyc); run<6>("v_log_f32", 8, out, cyc); run<7>("v_cmp_eq + v_cndmask (2 instr)", 8, out, cyc);
    run<8>("v_cvt_pk_bf16_f32", 8, out, cyc); run<9>("v_lshr + v_and / v_bfe", 8, out, cyc); run<10>("v_and + v_cvt_f32_ubyte0", 8, out, cyc);
    run<11>("v_lshlrev_b32", 8, out, cyc); run<12>("v_sub_f32 (1 - x)", 8, out, cyc);
    run<13>("v_cvt_scalef32_pk_f32_fp4 + v_add_u32", 8, out, cyc); run<14>("v_perm_b32", 8, out, cyc); run<15>("v_add_u32", 8, out, cyc);
    return 0;
}
