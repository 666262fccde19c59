// How do v_mfma_f32_16x16x32_bf16 and VALU instructions share one SIMD's issue slots?  (gfx950)
// Every wave runs: per iteration NM MFMAs (independent accumulators, or one dependent chain) + NV independent VALU ops of one
// kind.  Grid = 256 CUs x WPS waves per SIMD.  Reported: shader cycles per iteration per SIMD (s_memtime is 100 MHz wall
// clock; clock64() = s_memrealtime?  we use wall time of the whole launch and the measured sclk instead).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define ITER 4096

enum { K_FMA = 0, K_PKFMA = 1, K_LOG = 2, K_RCP = 3, K_MAX3 = 4, K_CVT = 5, K_ADD2 = 6, K_FMA3 = 7, K_MAXC = 8, K_AND = 9, K_SHL = 10,
       K_CNDMASK = 11, K_CMP = 12, K_FP4 = 13, K_PKADD2 = 14, K_PERM = 15, K_MED3 = 16, K_MAX2 = 17, K_MULC = 18, K_SQRT = 19, K_EXP = 20 };

template <int NM, int NV, int KIND, bool CHAIN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
    float a[8];
    f32x2 p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = threadIdx.x + j + 1.5f; p[j] = (f32x2){a[j], a[j] + 0.25f}; }
    const float cc = 1.0000001f, dd = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int q = CHAIN ? 0 : (m & 3);
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c[q], 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int j = v & 7;
            if (KIND == K_FMA) a[j] = fmaf(a[j], cc, dd);
            else if (KIND == K_PKFMA) p[j] = __builtin_elementwise_fma(p[j], (f32x2){cc, cc}, (f32x2){dd, dd});
            else if (KIND == K_LOG) a[j] = __builtin_amdgcn_logf(a[j]);
            else if (KIND == K_RCP) a[j] = __builtin_amdgcn_rcpf(a[j]);
            else if (KIND == K_MAX3) a[j] = __builtin_fmaxf(__builtin_fmaxf(a[j], cc), a[(j + 1) & 7]);
            else if (KIND == K_ADD2) a[j] = a[j] + a[(j + 3) & 7];
            else if (KIND == K_FMA3) a[j] = fmaf(a[j], a[(j + 3) & 7], a[(j + 5) & 7]);
            else if (KIND == K_MAXC) a[j] = __builtin_fmaxf(a[j], cc);
            else if (KIND == K_MAX2) a[j] = __builtin_fmaxf(a[j], a[(j + 3) & 7]);
            else if (KIND == K_MULC) a[j] = a[j] * cc;
            else if (KIND == K_AND) a[j] = __uint_as_float(__float_as_uint(a[j]) & 0xFFFF0000u);
            else if (KIND == K_SHL) a[j] = __uint_as_float(__float_as_uint(a[j]) << 1);
            else if (KIND == K_CNDMASK) a[j] = (threadIdx.x & (1 << (v & 3))) ? a[j] : a[(j + 3) & 7];     // mask is loop-invariant: v_cndmask only
            else if (KIND == K_CMP) a[j] = (a[(j + 1) & 7] > cc) ? a[j] : a[(j + 3) & 7];                 // v_cmp + v_cndmask
            else if (KIND == K_FP4) { const f32x2 r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(__float_as_uint(a[j]), 1.0f, 0); a[j] = r.x + 1.f; a[(j + 1) & 7] = r.y; }
            else if (KIND == K_PKADD2) p[j] = p[j] + p[(j + 3) & 7];
            else if (KIND == K_PERM) a[j] = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(a[j]), __float_as_uint(a[(j + 3) & 7]), 0x07060302u));
            else if (KIND == K_MED3) a[j] = __builtin_amdgcn_fmed3f(a[j], 0.f, 1.f);
            else if (KIND == K_SQRT) a[j] = __builtin_amdgcn_sqrtf(a[j]);
            else if (KIND == K_EXP) a[j] = __builtin_amdgcn_exp2f(a[j]);
            else if (KIND == K_CVT) {
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                const bf2 r = __builtin_convertvector((f32x2){a[j], a[(j + 1) & 7]}, bf2);
                a[j] = __uint_as_float(__builtin_bit_cast(unsigned, r));
            }
        }
        asm volatile("" ::: "memory");
    }
    float acc = c[0][0] + c[1][1] + c[2][2] + c[3][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] + p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static double g_ghz = 2.4;

template <int NM, int NV, int KIND, bool CHAIN>
void run(const char* name, float* out, int wps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wps;                       // 256-thread blocks = 1 wave per SIMD each
    hipLaunchKernelGGL((k<NM, NV, KIND, CHAIN>), dim3(blocks), dim3(256), 0, 0, out, ITER);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, NV, KIND, CHAIN>), dim3(blocks), dim3(256), 0, 0, out, ITER);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * g_ghz * 1e9 / ITER / wps;        // cycles per iteration per wave slot on a SIMD
    printf("%-34s wps=%d  %7.3f ms  %7.1f cycles/iter/wave   (additive model %d, overlap model %d)\n", name, wps, best, cyc,
           NM * 16 + NV * 4, (NM * 16 > NV * 4 + NM * 4) ? NM * 16 : NV * 4 + NM * 4);
    if (NM == 0) printf("%-34s        -> %.2f cycles per instruction\n", "", cyc / NV);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    g_ghz = clk * 1e-6;
    printf("device clock rate attribute: %.3f GHz (cycles below assume it)\n", g_ghz);
    for (int wps = 1; wps <= 3; wps += 2) {
        run<0, 32, K_FMA, false>("32 v_fma (v, s, s)", out, wps);
        run<0, 32, K_FMA3, false>("32 v_fma (v, v, v)", out, wps);
        run<0, 32, K_ADD2, false>("32 v_add (v, v)", out, wps);
        run<0, 32, K_MULC, false>("32 v_mul (v, s)", out, wps);
        run<0, 32, K_MAXC, false>("32 v_max (v, s)", out, wps);
        run<0, 32, K_MAX2, false>("32 v_max (v, v)", out, wps);
        run<0, 32, K_MED3, false>("32 v_med3 (v, 0, 1)", out, wps);
        run<0, 32, K_MAX3, false>("32 v_max3", out, wps);
        run<0, 32, K_AND, false>("32 v_and", out, wps);
        run<0, 32, K_SHL, false>("32 v_lshlrev", out, wps);
        run<0, 32, K_PERM, false>("32 v_perm", out, wps);
        run<0, 32, K_CNDMASK, false>("32 v_cndmask", out, wps);
        run<0, 32, K_CMP, false>("32 v_cmp + v_cndmask", out, wps);
        run<0, 32, K_FP4, false>("32 cvt_scalef32_pk_f32_fp4 + v_add", out, wps);
        run<0, 32, K_PKFMA, false>("32 v_pk_fma (v, s, s)", out, wps);
        run<0, 32, K_PKADD2, false>("32 v_pk_add (v, v)", out, wps);
        run<0, 32, K_CVT, false>("32 v_cvt_pk_bf16", out, wps);
        run<0, 32, K_LOG, false>("32 v_log", out, wps);
        run<0, 32, K_RCP, false>("32 v_rcp", out, wps);
        run<0, 32, K_SQRT, false>("32 v_sqrt", out, wps);
        run<0, 32, K_EXP, false>("32 v_exp", out, wps);
        run<4, 0, K_FMA, true>("4 mfma (chain)", out, wps);
        run<4, 64, K_FMA, true>("4 mfma(chain) + 64 v_fma", out, wps);
        run<4, 32, K_PKFMA, true>("4 mfma(chain) + 32 v_pk_fma", out, wps);
        run<4, 32, K_LOG, true>("4 mfma(chain) + 32 v_log", out, wps);
        run<0, 64, K_FMA, false>("64 v_fma", out, wps);
    }
    return 0;
}
