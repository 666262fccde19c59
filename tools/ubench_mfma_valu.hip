// Do matrix-pipe time and VALU time overlap on one SIMD when they come from DIFFERENT waves (wave specialisation)?
//
// r02's tools/ubench_issue.hip answered the question for waves that each issue both kinds (time = VALU + MFMA); this one puts
// MFMA-only waves beside VALU-only waves on the same SIMD (the block's wave w sits on SIMD w % 4 -- printed from HW_ID below), the
// shape a wave-specialised pass 2 would have: one matrix wave per SIMD (recon, dQ, dP from LDS-staged operands) and the BCE
// algebra in the others.  Every case reports t(MFMA waves only), t(VALU waves only), t(both) and both / max, both / sum.
//
//   A  1 MFMA wave + 1 VALU wave per SIMD, v_mfma_f32_16x16x4_f32 (the f32 MFMA runs on the vector ALUs: the control)
//   B  1 + 1, v_mfma_f32_16x16x32_bf16, VALU = independent v_fma_f32
//   C  1 + 3 (a 1024-thread block), bf16, VALU = v_fma_f32, VALU work per MFMA swept: from MFMA-bound to pass 2's ratio (VALU : MFMA = 4 : 1)
//   D  1 + 2 and 1 + 3, bf16, VALU = pass 2's real per-genotype mix (decode_bce_bf16_kernel's tile body: fp4 conversion, den, rcp,
//      clamp-multiply, the one-log-per-pair loss, the bf16 hi + lo split) -- the stream that would sit in the BCE waves
//   E  the same-wave form for reference (every wave issues both, 3 waves per SIMD): pass 2 as built
//   F  packed against scalar f32 next to MFMAs in ONE wave (MI355X_MICROARCH.md prices v_pk_*_f32 beside MFMAs at +22..26 cycles per
//      instruction against two scalar ones): {1 MFMA + N v_pk_fma_f32} against {1 MFMA + 2N v_fma_f32}, 1 and 3 waves per SIMD
//
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma_valu tools/ubench_mfma_valu.hip     (-> profiles/r05_ubench_mfma_valu.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int ITER = 4096;

template <bool BF16>
__device__ __forceinline__ float mfma_loop(int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < iters; ++i) {                                          // 4 independent accumulators: issue-bound, not latency-bound
        if (BF16) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
}

// NV independent v_fma_f32 per iteration
template <int NV>
__device__ __forceinline__ float fma_loop(int iters) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    const float c = 1.0000001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], c, d);
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    }
    float acc = 0.f;
    for (int j = 0; j < 8; ++j) acc += v[j];
    return acc;
}

// pass 2's per-pair-of-genotypes VALU work (csrc/nadm_genotype_passes.hip: bce_grad2, bce_loss_prod2, the hi + lo split), on
// register data: one call = two genotypes.  Returns through the accumulators so nothing is dead.
__device__ __forceinline__ void bce_pair(const f32x2 d, const uint32_t cw, const float eps, float& lacc, uint32_t& hsum, uint32_t& lsum) {
    const f32x2 x = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, 1.0f, 0);
    const f32x2 den = __builtin_elementwise_fma(-d, d, d);
    const f32x2 inv = {__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(den.x) * eps, 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(den.y) * eps, 0.f, 1.f)};
    const f32x2 dR = (d - x) * inv;
    const f32x2 h = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4((cw & ~(cw >> 1)) & 0x11111111u, 2.0f, 0);
    const f32x2 o = {__builtin_amdgcn_fmed3f(1.f - d.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - d.y, 0.f, 1.f)};
    const f32x2 q = o - x;
    const f32x2 qq = q * q;
    const f32x2 f = __builtin_elementwise_fma(h, den - qq, qq);
    lacc += __builtin_amdgcn_logf(f.x * f.y);
    const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(dR, bf16x2_t));
    const f32x2 rem = dR - (f32x2){__uint_as_float(hp << 16), __uint_as_float(hp & 0xFFFF0000u)};
    const uint32_t lp = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, bf16x2_t));
    hsum ^= hp; lsum ^= lp;
}
// NP pairs per iteration (a 16 x 16 tile is 2 pairs per lane: pass 2 runs 4 MFMAs per 2 pairs, i.e. NP = 2 per 4 MFMAs)
template <int NP>
__device__ __forceinline__ float bce_loop(int iters) {
    float eps_v = 1e-12f;
    asm volatile("" : "+s"(eps_v));
    float lacc = 0.f;
    uint32_t hs = 0, ls = 0;
    f32x2 d[4];
    uint32_t cw = threadIdx.x * 2654435761u;
    for (int j = 0; j < 4; ++j) d[j] = (f32x2){0.1f + 1e-3f * (threadIdx.x & 63) + 0.01f * j, 0.7f - 1e-3f * (threadIdx.x & 63) - 0.01f * j};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NP; ++j) bce_pair(d[j & 3], (cw >> (j & 3)) & 0x33333333u, eps_v, lacc, hs, ls);
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(cw));     // loop-variant to the compiler: no hoisting
    }
    return lacc + __uint_as_float((hs ^ ls) & 0x007FFFFFu);
}

// MODE bit 0: MFMA waves active, bit 1: VALU waves active.  Waves 0..3 = the MFMA wave of SIMD 0..3, the other WPB - 4 are VALU waves.
// VKIND 0: NV v_fma per iteration, 1: NV bce pairs per iteration.  MIT / VIT: iterations of the two loops
template <int MODE, bool BF16, int WPB, int VKIND, int NV>
__global__ __launch_bounds__(64 * WPB) void k_spec(float* out, int mit, int vit, uint32_t* simd_of_wave) {
    const int wave = threadIdx.x >> 6;
    float acc = 0.f;
    if (wave < 4) {
        if (MODE & 1) acc = mfma_loop<BF16>(mit);
    } else if (MODE & 2) {
        acc = VKIND == 0 ? fma_loop<NV>(vit) : bce_loop<NV>(vit);
    }
    if (simd_of_wave && blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        simd_of_wave[wave] = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);       // HW_REG_HW_ID, SIMD_ID = bits [5:4] (the field arrives at bit 0)
    out[blockIdx.x * (64 * WPB) + threadIdx.x] = acc;
}

// same-wave form: every wave runs {NM MFMAs + its VALU work} per iteration; WPS waves per SIMD
template <int MODE, int WPS, int VKIND, int NV, int NM>
__global__ __launch_bounds__(256 * WPS) void k_same(float* out, int iters) {
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    f32x2 pv[8];
    for (int j = 0; j < 8; ++j) pv[j] = (f32x2){(float)threadIdx.x + j, 1.f + j};
    const float cc = 1.0000001f, dd = 0.5f;
    const f32x2 pc = {cc, cc}, pd = {dd, dd};
    float eps_v = 1e-12f;
    asm volatile("" : "+s"(eps_v));
    float lacc = 0.f;
    uint32_t hs = 0, ls = 0, cw = threadIdx.x * 2654435761u;
    f32x2 d[4];
    for (int j = 0; j < 4; ++j) d[j] = (f32x2){0.1f + 1e-3f * (threadIdx.x & 63) + 0.01f * j, 0.7f - 1e-3f * (threadIdx.x & 63) - 0.01f * j};
    for (int i = 0; i < iters; ++i) {
        if (MODE & 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) c[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c[m & 3], 0, 0, 0);
        }
        if (MODE & 2) {
            if (VKIND == 0) {                                                   // scalar f32
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], cc, dd);
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            } else if (VKIND == 2) {                                            // packed f32: NV v_pk_fma_f32
#pragma unroll
                for (int j = 0; j < NV; ++j) pv[j & 7] = __builtin_elementwise_fma(pv[j & 7], pc, pd);
                asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]));
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j) bce_pair(d[j & 3], (cw >> (j & 3)) & 0x33333333u, eps_v, lacc, hs, ls);
                asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(cw));
            }
        }
    }
    float acc = c[0][0] + c[1][1] + c[2][2] + c[3][3] + lacc + __uint_as_float((hs ^ ls) & 0x007FFFFFu);
    for (int j = 0; j < 8; ++j) acc += v[j] + pv[j].x + pv[j].y;
    out[blockIdx.x * (256 * WPS) + threadIdx.x] = acc;
}

static float* g_out;
static uint32_t* g_simd;
template <typename F> float timed(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();                                                        // warm (clock ramp)
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}
void verdict(const char* name, float tm, float tv, float tb) {
    const float mx = tm > tv ? tm : tv;
    printf("%-64s mfma %.3f  valu %.3f  both %.3f ms   both/max %.2f  both/sum %.2f\n", name, tm, tv, tb, tb / mx, tb / (tm + tv));
}
template <bool BF16, int WPB, int VKIND, int NV> void spec(const char* name, int mit, int vit, bool print_simd = false) {
    const float tm = timed([&] { hipLaunchKernelGGL((k_spec<1, BF16, WPB, VKIND, NV>), dim3(256), dim3(64 * WPB), 0, 0, g_out, mit, vit, (uint32_t*)nullptr); });
    const float tv = timed([&] { hipLaunchKernelGGL((k_spec<2, BF16, WPB, VKIND, NV>), dim3(256), dim3(64 * WPB), 0, 0, g_out, mit, vit, (uint32_t*)nullptr); });
    const float tb = timed([&] { hipLaunchKernelGGL((k_spec<3, BF16, WPB, VKIND, NV>), dim3(256), dim3(64 * WPB), 0, 0, g_out, mit, vit, g_simd); });
    verdict(name, tm, tv, tb);
    if (print_simd) {
        uint32_t h[16]; hipMemcpy(h, g_simd, sizeof(h), hipMemcpyDeviceToHost);
        printf("    SIMD of the block's waves 0..%d:", WPB - 1);
        for (int w = 0; w < WPB; ++w) printf(" %u", h[w] & 3);
        printf("   (waves 0-3 issue the MFMAs)\n");
    }
}
template <int WPS, int VKIND, int NV, int NM> void same(const char* name, int iters) {
    const float tm = timed([&] { hipLaunchKernelGGL((k_same<1, WPS, VKIND, NV, NM>), dim3(256), dim3(256 * WPS), 0, 0, g_out, iters); });
    const float tv = timed([&] { hipLaunchKernelGGL((k_same<2, WPS, VKIND, NV, NM>), dim3(256), dim3(256 * WPS), 0, 0, g_out, iters); });
    const float tb = timed([&] { hipLaunchKernelGGL((k_same<3, WPS, VKIND, NV, NM>), dim3(256), dim3(256 * WPS), 0, 0, g_out, iters); });
    verdict(name, tm, tv, tb);
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4);
    hipMalloc(&g_simd, 64);
    hipMemset(g_simd, 0, 64);
    printf("grid = 256 blocks (one per CU); every time is the best of 3 after 2 warm-up launches\n");
    printf("--- A/B: 1 MFMA wave + 1 VALU wave per SIMD (512-thread block), VALU = 32 v_fma per iteration, MFMA = 4 per iteration\n");
    spec<false, 8, 0, 32>("A f32 16x16x4  (VALU-pipe MFMA: the control, must be additive)", ITER / 2, ITER, true);
    spec<true, 8, 0, 32>("B bf16 16x16x32", ITER, ITER, true);
    printf("--- C: 1 MFMA wave + 3 VALU waves per SIMD (1024-thread block), bf16; MFMA wave: 4 MFMA x %d; VALU waves: NV v_fma x %d each\n", ITER, ITER);
    spec<true, 16, 0, 8>("C NV =  8 (VALU:MFMA time ~ 1:1)", ITER, ITER, true);
    spec<true, 16, 0, 16>("C NV = 16 (~2:1)", ITER, ITER);
    spec<true, 16, 0, 32>("C NV = 32 (~4:1, pass 2's ratio)", ITER, ITER);
    printf("--- D: VALU = pass 2's BCE mix (one 'pair' = 2 genotypes: fp4 cvt, den, 2 rcp, 2 clamp-mul, loss with 1 log per pair, hi+lo bf16 split)\n");
    printf("       pass 2 issues 4 MFMAs per 2 pairs and wave; a matrix wave serving W BCE waves issues 4 MFMAs per 2 W pairs\n");
    spec<true, 12, 1, 2>("D 1 + 2 waves/SIMD: BCE waves 2 pairs/iter, MFMA wave 8 MFMA/iter", 2 * ITER, ITER, true);
    spec<true, 16, 1, 2>("D 1 + 3 waves/SIMD: BCE waves 2 pairs/iter, MFMA wave 12 MFMA/iter", 3 * ITER, ITER, true);
    printf("--- E: the same-wave form (every wave: 4 MFMAs + 2 BCE pairs per iteration), 3 waves per SIMD = pass 2 as built\n");
    same<3, 1, 2, 4>("E 3 waves/SIMD, 4 MFMA + 2 pairs", ITER);
    same<2, 1, 2, 4>("E 2 waves/SIMD, 4 MFMA + 2 pairs", ITER);
    printf("--- F: packed against scalar f32 beside MFMAs in ONE wave: per iteration 2 MFMAs + {8 v_pk_fma_f32 | 16 v_fma_f32} (same flops)\n");
    same<1, 2, 8, 2>("F 1 wave/SIMD  packed  (8 v_pk_fma)", ITER);
    same<1, 0, 16, 2>("F 1 wave/SIMD  scalar (16 v_fma)", ITER);
    same<3, 2, 8, 2>("F 3 waves/SIMD packed  (8 v_pk_fma)", ITER);
    same<3, 0, 16, 2>("F 3 waves/SIMD scalar (16 v_fma)", ITER);
    return 0;
}
