// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nadm.h"

namespace nadm {

// ---- genotype decode -------------------------------------------------------------------------
// 2-bit code -> x in {0, .5, 1}; 3 (missing) -> 0   (reference: neural_admixture.py:169-170).
// One 64-bit shift through a packed table of bf16 bit patterns {0x0000,0x3F00,0x3F80,0x0000}.
__device__ __forceinline__ float decode_x(uint32_t code2) {
    const uint64_t tab = 0x00003F803F000000ull;
    return __uint_as_float(static_cast<uint32_t>(tab >> (code2 << 4)) << 16);
}

// ---- lane masks without compares -----------------------------------------------------------------
// A select on a lane-dependent condition becomes v_cndmask_b32; hipcc prefers its VOP2 form with the mask in VCC, and THAT form
// costs 23 cycles per wave64 instruction on gfx950 against 4.6 with the mask in an SGPR pair and 2.8 / 4.5 for v_and / v_bfi with a
// mask register (profiles/r03_ubench_cndmask.txt).  Prologues made of dozens of such selects (operand blends by lane group, "row
// past M -> 0") were 14 % of a pass-1 block's life.  Masks are therefore formed arithmetically -- from a wrapping subtraction, so that
// LLVM cannot fold them back into a compare + select -- and applied with bit operations.
__device__ __forceinline__ uint32_t lt_mask(int a, int b) {            // a < b ? 0xFFFFFFFF : 0   (|a - b| < 2^31)
    uint32_t m = (uint32_t)((int)((uint32_t)a - (uint32_t)b) >> 31);
    asm("" : "+v"(m));                  // opaque: with known operand ranges LLVM proves "m is sext(a < b)" and selects again
    return m;
}
__device__ __forceinline__ uint32_t lt_mask64(int64_t a, int64_t b) {  // the same for 64-bit operands
    uint32_t m = (uint32_t)((int64_t)((uint64_t)a - (uint64_t)b) >> 63);
    asm("" : "+v"(m));
    return m;
}
__device__ __forceinline__ float keepf(float x, uint32_t m) { return __uint_as_float(__float_as_uint(x) & m); }
__device__ __forceinline__ uint32_t blend(uint32_t if_set, uint32_t if_clear, uint32_t m) { return (if_set & m) | (if_clear & ~m); }   // v_bfi_b32

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

// ---- Adam element (torch.optim.Adam closed form, betas .9/.95, eps 1e-8; neural_admixture.py:187-204) + restrict_P ----
// inv_bc2 = 1 / sqrt(1 - 0.95^t), step_size = lr / (1 - 0.9^t): nadm_host.h adam_scalars
// One definition for the stand-alone kernel and for the epilogues of passes 2 and 3 that apply the update to the rows whose
// gradient they have just completed: the same operations in the same order, hence the same bits.
struct AdamFused {            // m == nullptr: no fused update
    float* m;
    float* v;
    float step_size, inv_bc2, grad_scale;
};
__device__ __forceinline__ float adam_element(float p, float g, float& m, float& v, float step_size, float inv_bc2,
                                              float grad_scale, bool clamp01) {
    const float b2 = 0.95f, eps = 1e-8f;
    const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.95);
    const float gr = g * grad_scale;
    m = m + (gr - m) * omb1;
    v = v * b2 + gr * gr * omb2;
    // sqrt(v) / sqrt(1 - b2^t) + eps and m / den with the hardware's 1-ulp square root and reciprocal (v_sqrt_f32, v_rcp_f32) and
    // the bias correction as a multiplication: the IEEE forms cost ten-instruction division sequences and a fixed-up square root
    // per element -- 5 % of pass 2's and a quarter of pass 3's wave time in their Adam epilogues (r03) -- for a step that differs
    // by < 4e-7 relative (the update is <= lr = 2e-3: 1e-9 on a parameter).  v >= 0 always; a denormal v reads as 0, den = eps.
    const float den = __builtin_amdgcn_sqrtf(v) * inv_bc2 + eps;
    float np_ = p - step_size * (m * __builtin_amdgcn_rcpf(den));
    if (clamp01) np_ = fminf(fmaxf(np_, 0.f), 1.f);
    return np_;
}
__device__ __forceinline__ void adam_float4(float* __restrict__ p, const float4 G4, float* __restrict__ m, float* __restrict__ v,
                                            float step_size, float inv_bc2, float grad_scale, bool clamp01) {
    const float4 P4 = *reinterpret_cast<const float4*>(p);
    const float4 M4 = *reinterpret_cast<const float4*>(m);
    const float4 V4 = *reinterpret_cast<const float4*>(v);
    float pp[4] = {P4.x, P4.y, P4.z, P4.w}, gg[4] = {G4.x, G4.y, G4.z, G4.w};
    float mm[4] = {M4.x, M4.y, M4.z, M4.w}, vv[4] = {V4.x, V4.y, V4.z, V4.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) pp[q] = adam_element(pp[q], gg[q], mm[q], vv[q], step_size, inv_bc2, grad_scale, clamp01);
    *reinterpret_cast<float4*>(p) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *reinterpret_cast<float4*>(m) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *reinterpret_cast<float4*>(v) = make_float4(vv[0], vv[1], vv[2], vv[3]);
}

// ---- bf16 pieces of an fp32 value (round-to-nearest-even conversions, v_cvt_pk_bf16_f32) ----
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float lo_half, float hi_half) {      // RNE, one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo_half, hi_half}, bf16x2_t));
}
// 3-way bf16 split of one fp32 value: 16-bit patterns of hi, mid, lo
__device__ __forceinline__ void split3(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = pk_bf16(v, 0.f) & 0xFFFFu;
    const float r1 = v - __uint_as_float(h << 16);
    m = pk_bf16(r1, 0.f) & 0xFFFFu;
    const float r2 = r1 - __uint_as_float(m << 16);
    l = pk_bf16(r2, 0.f) & 0xFFFFu;
}

// ---- dZ as the operand image of pass 3 on the FP4 x FP6 matrix instruction (C <= 8; encode_bwd_fp4_kernel) --------------------------
// Per tile of 128 samples (= K of v_mfma_scale_f32_16x16x128_f8f6f4) 7 x 64 uint4: uint4 k = 0..5 of lane l hold the lane's 24 operand
// dwords (row group rg = 0..3: dwords 6 rg .. 6 rg + 5 = 32 FP6 codes), uint4 k = 6 holds in .x the four E8M0 scale bytes (byte rg).
// Lane l = 16 q + 8 parity + c is row (parity, column c) and K-block q of the instruction; row group rg carries pieces 2 rg + parity.
// A lane's 32 values (one column, 32 consecutive samples) are cut into eight pieces of four bits: |v| in fixed point below 2^(E0 + 4)
// > the block's largest magnitude, n = |v| / 2^(E0 - 28) < 2^32, piece p = hexadecimal digit p of n with the sign of v.  Digit h is the
// FP6 (E2M3) number h / 8 exactly, so piece p enters with the scale 2^(E0 - 4p + 3).  An element within 2^-8 of the block maximum is
// carried exactly, a smaller one to an absolute error < 2^-31 of that maximum.
constexpr int DZI_TS = 128;
constexpr int DZI_TILE_U4 = 7 * 64;
// sample (within its tile) of element e = 0..31 of K-block q: the order pass 3's bit transposition produces (see there)
__host__ __device__ constexpr int dzi_sample(int q, int e) { return 32 * q + 8 * (e >> 3) + 4 * (e & 1) + ((e & 7) >> 1); }
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x6_t __attribute__((ext_vector_type(6)));
// piece p (0..7) of 32 values that share one scale block: the 32 FP6 codes (6 dwords, element e at bits [6e, 6e + 6)) and the E8M0 scale
__device__ __forceinline__ void fp6_piece(const float (&v)[32], int p, u32x6_t& codes, int& scale_byte) {
    float mx = 0.f;
    bool finite = true;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        mx = __builtin_fmaxf(mx, __builtin_fabsf(v[e]));
        finite = finite && (__builtin_fabsf(v[e]) < __builtin_inff());       // (a NaN does not survive the maximum)
    }
    // mx = m 2^ex, m in [1/2, 1): every |v| < 2^ex = 2^(E0 + 4)
    const int E0 = __builtin_amdgcn_frexp_expf(mx) - 4;
    int sb = 127 + E0 - 4 * p + 3;
    const bool dead = !(mx > 0.f) || !finite || sb < 1;     // (a piece below 2^-126 carries nothing an fp32 sum would see)
    if (dead) sb = finite ? 127 : 255;                      // inf / NaN in the block: the E8M0 scale NaN, so that what multiplies it becomes NaN
    f32x16_t fa, fb;
    const uint32_t digit = dead ? 0u : 15u;                 // ONE select: per element it is a v_cndmask with its mask in VCC, 23 cycles x 32 on the tail of the MLP backward
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const uint32_t n = (uint32_t)__builtin_ldexpf(__builtin_fabsf(v[e]), 28 - E0);
        const uint32_t h = (n >> (28 - 4 * p)) & digit;
        const float f = __builtin_copysignf((float)h, v[e]);
        if (e & 1) fb[e >> 1] = f; else fa[e >> 1] = f;      // the conversion interleaves its two sources: field 2i = a[i], 2i + 1 = b[i]
    }
    codes = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(fa, fb, 8.0f);      // h -> the code of h / 8, exact
    scale_byte = sb;
}
// where piece p of (column c, K-block q) goes in a tile image: consumer lane 16 q + 8 parity + c, row group p >> 1
__device__ __forceinline__ void fp6_piece_store(uint4* tile, int q, int c, int p, const u32x6_t& codes, int scale_byte) {
    const int l = 16 * q + 8 * (p & 1) + c, rg = p >> 1;
    uint32_t* base = reinterpret_cast<uint32_t*>(tile);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = 6 * rg + i;
        base[((d >> 2) * 64 + l) * 4 + (d & 3)] = codes[i];
    }
    reinterpret_cast<uint8_t*>(base)[((6 * 64 + l) * 4) * 4 + rg] = (uint8_t)scale_byte;
}
// one thread: piece p of column c of K-block q of tile T.  GROUP = false: dZ is the [b, CP] matrix in global memory; true: dZ points at
// the K-block's 32 samples x 8 columns in LDS (rows past the batch and columns past CP zeroed by the caller)
template <bool GROUP>
__device__ __forceinline__ void dzi_build_piece(const float* dZ, int b, int CP, uint4* img, int T, int q, int c, int p) {
    float v[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        if constexpr (GROUP) {
            v[e] = dZ[(dzi_sample(q, e) - 32 * q) * 8 + c];
        } else {
            const int s = T * DZI_TS + dzi_sample(q, e);
            const float x = dZ[(int64_t)(s < b ? s : b - 1) * CP + (c < CP ? c : 0)];
            v[e] = (s < b && c < CP) ? x : 0.f;
        }
    }
    u32x6_t codes;
    int sb;
    fp6_piece(v, p, codes, sb);
    fp6_piece_store(img + (int64_t)T * DZI_TILE_U4, q, c, p, codes, sb);
}

// ---- Q as the MFMA operand images of the bf16 pass 2 (K <= 16), one image per head and 64-sample tile --------------------------
// Every block of pass 2 (1954 of them at M = 500k) needs the batch's Q as bf16 pieces laid out as its MFMA operands; built
// from the fp32 Q inside pass 2 that is two split3 and 16 two-byte LDS stores per thread and tile, 3.5 % of the kernel without
// the loss value.  The MLP forward, which has each Q element in a register once per step, can write the images instead
// (nadm_mlp_fwd_images); pass 2 then copies 10-12 KB per tile with 16-byte loads.  Layout of one tile image (uint4 units; a
// uint4 = the 8 bf16 of one lane of one operand):
//   [0, 512)            R^T operands  [sample tile st = 0..3][operand 0 / 1][lane]
//   [512, 512 + 128 W2) dP operands   [sample pair 0..1][operand 0 .. W2-1][lane]          W2 = 1 for K <= 8, 2 for K 9..16
// with the slot assignment of decode_bce_bf16_kernel (see there).  Slots that hold no piece are zero and are never written:
// the buffer must be zero-filled once.  Rows b .. 64*ceil(b/64)-1 of the last tile are written as zeros by the producer.
constexpr int QI_TS = 64;                                  // samples per tile (= NADM_BF_TS of the pass-2 build)
constexpr int QI_TILE_U4 = 768;                            // uint4 per tile image in global memory (K <= 8 uses the first 640)
__device__ __host__ constexpr int qi_tile_u4(int kp) { return kp > 8 ? 768 : 640; }
// element (sample qr of the tile, column qk of the head) with value v -> its pieces in the tile image `img` (uint16 view)
__device__ __forceinline__ void q_image_put(uint16_t* __restrict__ img, const int kp, const int qr, const int qk, const float v) {
    uint32_t h, md, lo;
    split3(v, h, md, lo);
    const int st = qr >> 4, i = qr & 15;
    const int pair = qr >> 5, within = qr & 31, q8 = within >> 3, e = within & 7;
    if (kp > 8) {
        const int sl = qk >> 3, kk = qk & 7;                                              // k slot and position inside it
        uint16_t* r1 = img + ((st * 2 + 0) * 64) * 8 + kk;                               // + lane * 8
        uint16_t* r2 = img + ((st * 2 + 1) * 64) * 8 + kk;
        r1[(i + 16 * sl) * 8] = (uint16_t)h;   r1[(i + 16 * (2 + sl)) * 8] = (uint16_t)md;   // [Qh Qh' Qm Qm']
        r2[(i + 16 * sl) * 8] = (uint16_t)lo;  r2[(i + 16 * (2 + sl)) * 8] = (uint16_t)h;    // [Ql Ql' Qh Qh']
        uint16_t* d1 = img + (512 + (pair * 2 + 0) * 64) * 8 + e;
        uint16_t* d2 = img + (512 + (pair * 2 + 1) * 64) * 8 + e;
        d1[(q8 * 16 + qk) * 8] = (uint16_t)h;                                             // column qk of Qh / Qm
        d2[(q8 * 16 + qk) * 8] = (uint16_t)md;
    } else if (kp <= 4) {                                                                 // one R^T instruction: slots [Qh|Qm Qh|Qm Ql|Qh 0], four k each
        uint16_t* r1 = img + ((st * 2 + 0) * 64) * 8 + qk;
        r1[(i) * 8] = (uint16_t)h;        r1[(i) * 8 + 4] = (uint16_t)md;
        r1[(i + 16) * 8] = (uint16_t)h;   r1[(i + 16) * 8 + 4] = (uint16_t)md;
        r1[(i + 32) * 8] = (uint16_t)lo;  r1[(i + 32) * 8 + 4] = (uint16_t)h;
        uint16_t* d1 = img + (512 + pair * 64) * 8 + e;
        d1[(q8 * 16 + qk) * 8] = (uint16_t)h;                                             // columns 0..7: Qh
        d1[(q8 * 16 + qk + 8) * 8] = (uint16_t)md;                                        // columns 8..15: Qm
    } else {
        uint16_t* r1 = img + ((st * 2 + 0) * 64) * 8 + qk;
        uint16_t* r2 = img + ((st * 2 + 1) * 64) * 8 + qk;
        r1[(i) * 8] = (uint16_t)h;        r1[(i + 32) * 8] = (uint16_t)h;                 // slots 0,2: Qh
        r1[(i + 16) * 8] = (uint16_t)md;  r1[(i + 48) * 8] = (uint16_t)md;                // slots 1,3: Qm
        r2[(i) * 8] = (uint16_t)lo;       r2[(i + 16) * 8] = (uint16_t)h;                 // slots 0,1: Ql, Qh
        uint16_t* d1 = img + (512 + pair * 64) * 8 + e;
        d1[(q8 * 16 + qk) * 8] = (uint16_t)h;                                             // columns 0..7: Qh
        d1[(q8 * 16 + qk + 8) * 8] = (uint16_t)md;                                        // columns 8..15: Qm
    }
}
// one Q element of the batch (sample s, column j of head hh, value v) into the head's images.  The rows from b to the end of
// the last tile must read as zeros (a shorter batch than the previous one leaves no stale rows): `first_of_block` threads (one
// per block and column) clear rows b + blockIdx.x, b + blockIdx.x + gridDim.x, ... of their column -- spread over the blocks,
// because one block clearing up to 63 rows with dependent two-byte stores made it the kernel's critical path (+4.6 us)
__device__ __forceinline__ void q_image_store(uint4* __restrict__ qimg, const int64_t head_stride_u4, const int hh, const int kp,
                                              const int s, const int j, const float v, const int b, const bool first_of_block) {
    if (qimg == nullptr || kp > 16) return;
    uint16_t* head = reinterpret_cast<uint16_t*>(qimg + hh * head_stride_u4);
    q_image_put(head + (int64_t)(s / QI_TS) * QI_TILE_U4 * 8, kp, s % QI_TS, j, v);
    if (first_of_block)
        for (int r = b + (int)blockIdx.x; r < (b + QI_TS - 1) / QI_TS * QI_TS; r += (int)gridDim.x)
            q_image_put(head + (int64_t)(r / QI_TS) * QI_TILE_U4 * 8, kp, r % QI_TS, j, 0.f);
}

// ---- MLP weight gradients, one block of the (hidden/256, sample splits) grid: see mlp_bwd_b_kernel (nadm_small_kernels.hip).
// A device function so that pass 3 can run these blocks as extra blocks of its own launch (both depend only on the MLP
// backward's outputs; the ~200 small blocks fill the under-occupied last round of pass 3 instead of a launch of their own).
constexpr int SJ = 16;          // samples per split in the weight-gradient kernel
__device__ __forceinline__ void mlp_bwd_b_block(const nadm_heads_t& hd, int b, const float* __restrict__ Zn,
                                                const float* __restrict__ H, const float* __restrict__ dL,
                                                const float* __restrict__ dHpre, const float* __restrict__ dgp,
                                                float* __restrict__ small_part, const int bx, const int by) {
    const int C = hd.C, CP = hd.CP, Hd = hd.Hd, SP = hd.SP;
    const int tid = threadIdx.x;
    const int h = bx * 256 + tid;
    const int j = by;
    const int i0 = j * SJ, i1 = min(b, i0 + SJ);
    float* out = small_part + (int64_t)j * hd.n_small;
    if (h < Hd) {
        float hv[SJ], dv[SJ];                 // this thread's column of H / dHpre for the split (coalesced loads, issued together)
#pragma unroll
        for (int ii = 0; ii < SJ; ++ii) {
            const bool ok = i0 + ii < i1;
            hv[ii] = ok ? H[(int64_t)(i0 + ii) * Hd + h] : 0.f;
            dv[ii] = ok ? dHpre[(int64_t)(i0 + ii) * Hd + h] : 0.f;
        }
        float ab = 0.f;
#pragma unroll
        for (int ii = 0; ii < SJ; ++ii) ab += dv[ii];
        out[hd.b1_off + h] = ab;
        for (int c = 0; c < C; ++c) {
            float a = 0.f;
#pragma unroll
            for (int ii = 0; ii < SJ; ++ii) a = fmaf(dv[ii], (i0 + ii < i1) ? Zn[(int64_t)(i0 + ii) * CP + c] : 0.f, a);
            out[hd.w1_off + h * C + c] = a;
        }
        for (int hh = 0; hh < hd.n_heads; ++hh) {
            const int o = hd.qoff[hh];
            for (int k = 0; k < hd.k[hh]; ++k) {
                float a = 0.f;
#pragma unroll
                for (int ii = 0; ii < SJ; ++ii) a = fmaf((i0 + ii < i1) ? dL[(int64_t)(i0 + ii) * SP + o + k] : 0.f, hv[ii], a);
                out[hd.wk_off[hh] + k * Hd + h] = a;
            }
        }
    }
    if (bx == 0) {
        for (int hh = 0; hh < hd.n_heads; ++hh)
            for (int k = tid; k < hd.k[hh]; k += 256) {
                float a = 0.f;
                for (int i = i0; i < i1; ++i) a += dL[(int64_t)i * SP + hd.qoff[hh] + k];
                out[hd.bk_off[hh] + k] = a;
            }
        for (int c = tid; c < C; c += 256) {
            float a = 0.f;
            for (int i = i0; i < i1; ++i) a += dgp[(int64_t)i * CP + c];
            out[hd.g_off + c] = a;
        }
    }
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
