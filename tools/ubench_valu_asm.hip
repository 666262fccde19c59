// Issue cost of individual gfx950 VALU instruction forms with a saturated SIMD (3 waves per SIMD, 256 CUs), inline asm so the
// compiler cannot fold, pack or re-associate anything.  Each loop iteration = 32 instructions of one form on 8 independent
// destination registers.  Prints SIMD cycles per instruction (wall time x clock / instructions issued per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_asm.hip -o /tmp/ubv && /tmp/ubv
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

#define B_ADD_VV(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
#define B_ADD_SV(j) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[j]));
#define B_MUL_VV(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
#define B_MAX_VV(j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
#define B_MAX_SV(j) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[j]));
#define B_FMA_VVV(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
#define B_FMAC(j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
#define B_MAX3(j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
#define B_MED3(j) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(a[j]));
#define B_MAXCL(j) asm volatile("v_max_f32_e64 %0, %0, %0 clamp" : "+v"(a[j]));
#define B_AND(j) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[j]));
#define B_SHL(j) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[j]));
#define B_PERM(j) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(c1), "v"(c2));
#define B_CNDMASK(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(c1) : "vcc");
#define B_CMP(j) asm volatile("v_cmp_eq_f32 vcc, %0, %1" : : "v"(a[j]), "v"(c1) : "vcc");
#define B_CMPS(j) asm volatile("v_cmp_eq_f32 s[20:21], %0, %1" : : "v"(a[j]), "v"(c1) : "s20", "s21");
#define B_PKADD(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(q));
#define B_PKMUL(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(q));
#define B_PKFMA(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(q));
#define B_CVTBF(j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
#define B_FP4(j) asm volatile("v_cvt_scalef32_pk_f32_fp4 %0, %1, 1.0" : "=v"(p[j]) : "v"(u[j]));
#define B_LOG(j) asm volatile("v_log_f32 %0, %0" : "+v"(a[j]));
#define B_RCP(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
#define B_MOV(j) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(c1));
#define B_LOGMIX(j) asm volatile("v_log_f32 %0, %0\n v_mul_f32 %1, %1, %2\n v_mul_f32 %3, %3, %2" : "+v"(a[j]), "+v"(p[j].x), "+v"(c1), "+v"(p[j].y));
#define B_BFE(j) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u[j]));
#define B_ADDU(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(c1));
#define B_MULU24(j) asm volatile("v_mul_u32_u24 %0, 0x10000, %0" : "+v"(u[j]));
#define B_ADDABS(j) asm volatile("v_add_f32_e64 %0, %0, |%0|" : "+v"(a[j]));
#define B_LSHLADD(j) asm volatile("v_lshl_add_u32 %0, %0, 16, %1" : "+v"(u[j]) : "v"(c1));
#define B_ANDOR(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(c1), "v"(c2));
#define B_OR(j) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[j]) : "v"(c1));
#define B_BFI(j) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[j]) : "v"(c1), "v"(c2));
#define B_ALIGNBIT(j) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(u[j]) : "v"(c1));
#define B_MIN(j) asm volatile("v_min_f32 %0, 1.0, %0" : "+v"(a[j]));
#define B_CVTI(j) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[j]));
#define B_FREXPM(j) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(a[j]));
#define B_LOGABS(j) asm volatile("v_log_f32_e64 %0, |%0|" : "+v"(a[j]));
#define B_MULLEG(j) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
#define B_PKMOV(j) asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(p[j]) : "v"(q));
#define B_MULLO(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j]) : "v"(c1));
#define B_MULHI(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[j]) : "v"(c1));
#define B_MAD64(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[j]) : "v"(u[j]), "v"(c1) : "vcc");
#define B_LSHLADD64(j) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(p[j]) : "v"(q));
#define B_LSHL64(j) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(p[j]));
#define B_MULU24(j2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[j2]) : "v"(c1));
#define B_MADU24(j) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[j]) : "v"(c1), "v"(c2));
#define B_SUBREV(j) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(a[j]));

KERNEL(k_add_vv, B_ADD_VV) KERNEL(k_add_sv, B_ADD_SV) KERNEL(k_mul_vv, B_MUL_VV) KERNEL(k_max_vv, B_MAX_VV) KERNEL(k_max_sv, B_MAX_SV)
KERNEL(k_fma_vvv, B_FMA_VVV) KERNEL(k_fmac, B_FMAC) KERNEL(k_max3, B_MAX3) KERNEL(k_med3, B_MED3) KERNEL(k_maxcl, B_MAXCL) KERNEL(k_and, B_AND)
KERNEL(k_shl, B_SHL) KERNEL(k_perm, B_PERM) KERNEL(k_cndmask, B_CNDMASK) KERNEL(k_cmp, B_CMP) KERNEL(k_cmps, B_CMPS) KERNEL(k_pkadd, B_PKADD)
KERNEL(k_pkmul, B_PKMUL) KERNEL(k_pkfma, B_PKFMA) KERNEL(k_cvtbf, B_CVTBF) KERNEL(k_fp4, B_FP4) KERNEL(k_log, B_LOG) KERNEL(k_rcp, B_RCP)
KERNEL(k_mulu24, B_MULU24) KERNEL(k_addabs, B_ADDABS) KERNEL(k_lshladd, B_LSHLADD) KERNEL(k_andor, B_ANDOR) KERNEL(k_or, B_OR) KERNEL(k_bfi, B_BFI)
KERNEL(k_alignbit, B_ALIGNBIT) KERNEL(k_min, B_MIN) KERNEL(k_cvti, B_CVTI) KERNEL(k_frexpm, B_FREXPM) KERNEL(k_logabs, B_LOGABS) KERNEL(k_mulleg, B_MULLEG) KERNEL(k_pkmov, B_PKMOV)
KERNEL(k_mullo, B_MULLO) KERNEL(k_mulhi, B_MULHI) KERNEL(k_mad64, B_MAD64) KERNEL(k_lshladd64, B_LSHLADD64) KERNEL(k_lshl64, B_LSHL64) KERNEL(k_madu24, B_MADU24)
KERNEL(k_mov, B_MOV) KERNEL(k_logmix, B_LOGMIX) KERNEL(k_bfe, B_BFE) KERNEL(k_addu, B_ADDU) KERNEL(k_subrev, B_SUBREV)

static double g_ghz = 2.4;
typedef void (*kern_t)(float*, int);
static void run(const char* name, kern_t fn, float* out, int wps, int per_body) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(256 * wps), dim3(256), 0, 0, out, ITER);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(256 * wps), dim3(256), 0, 0, out, ITER);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s wps=%d  %7.3f ms  %6.2f cycles per instruction\n", name, wps, best, best * 1e-3 * g_ghz * 1e9 / ITER / wps / (32.0 * per_body));
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    g_ghz = clk * 1e-6;
    printf("clock attribute %.3f GHz\n", g_ghz);
    for (int wps = 3; wps <= 3; ++wps) {
        run("v_add_f32 v, v, v", k_add_vv, out, wps, 1); run("v_add_f32 v, 1.0, v", k_add_sv, out, wps, 1); run("v_sub_f32 v, 1.0, v", k_subrev, out, wps, 1);
        run("v_mul_f32 v, v, v", k_mul_vv, out, wps, 1);
        run("v_max_f32 v, v, v", k_max_vv, out, wps, 1); run("v_max_f32 v, 0, v", k_max_sv, out, wps, 1); run("v_max_f32_e64 v, v, v clamp", k_maxcl, out, wps, 1);
        run("v_fma_f32 v, v, v, v", k_fma_vvv, out, wps, 1); run("v_fmac_f32 v, v, v", k_fmac, out, wps, 1);
        run("v_max3_f32 v, v, v, v", k_max3, out, wps, 1); run("v_med3_f32 v, v, 0, 1.0", k_med3, out, wps, 1);
        run("v_and_b32 v, lit, v", k_and, out, wps, 1); run("v_lshlrev_b32 v, 16, v", k_shl, out, wps, 1); run("v_bfe_u32", k_bfe, out, wps, 1);
        run("v_add_u32 v, v, v", k_addu, out, wps, 1); run("v_perm_b32 v, v, v, v", k_perm, out, wps, 1); run("v_mov_b32 v, v", k_mov, out, wps, 1);
        run("v_cndmask_b32 v, v, v, vcc", k_cndmask, out, wps, 1); run("v_cmp_eq_f32 vcc, v, v", k_cmp, out, wps, 1); run("v_cmp_eq_f32 s[20:21], v, v", k_cmps, out, wps, 1);
        run("v_pk_add_f32", k_pkadd, out, wps, 1); run("v_pk_mul_f32", k_pkmul, out, wps, 1); run("v_pk_fma_f32", k_pkfma, out, wps, 1);
        run("v_cvt_pk_bf16_f32", k_cvtbf, out, wps, 1); run("v_cvt_scalef32_pk_f32_fp4", k_fp4, out, wps, 1);
        run("v_log_f32", k_log, out, wps, 1); run("v_rcp_f32", k_rcp, out, wps, 1);
        run("v_log_f32 + 2 v_mul_f32 (per instruction)", k_logmix, out, wps, 3);
        run("v_mul_u32_u24 v, 0x10000, v", k_mulu24, out, wps, 1); run("v_add_f32_e64 v, v, |v|", k_addabs, out, wps, 1); run("v_lshl_add_u32", k_lshladd, out, wps, 1);
        run("v_and_or_b32", k_andor, out, wps, 1); run("v_or_b32", k_or, out, wps, 1); run("v_bfi_b32", k_bfi, out, wps, 1); run("v_alignbit_b32", k_alignbit, out, wps, 1);
        run("v_min_f32 v, 1.0, v", k_min, out, wps, 1); run("v_cvt_f32_i32", k_cvti, out, wps, 1); run("v_frexp_mant_f32", k_frexpm, out, wps, 1);
        run("v_mul_lo_u32 v, v, v", k_mullo, out, wps, 1); run("v_mul_hi_u32 v, v, v", k_mulhi, out, wps, 1); run("v_mad_u64_u32 v[2], vcc, v, v, v[2]", k_mad64, out, wps, 1);
        run("v_lshl_add_u64 v[2], v[2], 2, v[2]", k_lshladd64, out, wps, 1); run("v_lshlrev_b64 v[2], 3, v[2]", k_lshl64, out, wps, 1); run("v_mad_u32_u24 v, v, v, v", k_madu24, out, wps, 1);
        run("v_log_f32_e64 v, |v|", k_logabs, out, wps, 1); run("v_mul_legacy_f32", k_mulleg, out, wps, 1); run("v_pk_mov_b32", k_pkmov, out, wps, 1);
    }
    return 0;
}
