// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nadm {

// ---- genotype decode -------------------------------------------------------------------------
// 2-bit code -> x in {0, .5, 1}; 3 (missing) -> 0   (reference: neural_admixture.py:169-170).
// One 64-bit shift through a packed table of bf16 bit patterns {0x0000,0x3F00,0x3F80,0x0000}.
__device__ __forceinline__ float decode_x(uint32_t code2) {
    const uint64_t tab = 0x00003F803F000000ull;
    return __uint_as_float(static_cast<uint32_t>(tab >> (code2 << 4)) << 16);
}

// ---- wave64 reductions with DPP (no LDS traffic) -------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Sum over the 64 lanes; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);       // row_half_mirror
    v += dpp_f<0x140>(v);       // row_mirror       -> every lane of a 16-row holds the row sum
    v += dpp_f<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
    v += dpp_f<0x143, 0xc>(v);  // row_bcast31 into rows 2,3 -> lane 63 holds the wave sum
    return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
    v = wave_sum_lane63(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_sum_all_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- BCE element: clamp, loss term, gradient w.r.t. the pre-clamp reconstruction ------------
// dR = (r - x) / max(r(1-r), 1e-12) masked to 0 <= r_raw <= 1 (inclusive, on the PRE-clamp value);
// ATen binary_cross_entropy(_backward) + clamp_ backward, behind neural_admixture.py:97,288,410.
__device__ __forceinline__ float bce_grad(float r_raw, float x, float& r_out) {
    const float r = fminf(fmaxf(r_raw, 0.f), 1.f);
    const float den = fmaxf(fmaf(-r, r, r), 1e-12f);
    const float g = (r - x) * __builtin_amdgcn_rcpf(den);
    r_out = r;
    return (r == r_raw) ? g : 0.f;
}
// -[x*max(log r,-100) + (1-x)*max(log(1-r),-100)] with log(1-r) evaluated as log1p(-r).
__device__ __forceinline__ float bce_loss(float r, float x) {
    const float lr = fmaxf(__logf(r), -100.f);
    const float u = 1.f - r;
    float l1;
    if (u == 1.f) l1 = -r;                       // |r| < 2^-24: log1p(-r) = -r
    else l1 = __logf(u) * (-r * __builtin_amdgcn_rcpf(u - 1.f));   // log(u) * (-r)/(u-1)
    l1 = (r == 1.f) ? -100.f : fmaxf(l1, -100.f);
    return -(x * lr + (1.f - x) * l1);
}

}  // namespace nadm
