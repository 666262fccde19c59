// The three passes over the 2-bit packed genotype matrix (gfx950 / MI355X).
//
//   pass 1  encode_fwd   Z  = X . V            reduce over SNPs   (neural_admixture.py:169-172)
//   pass 2  decode_bce   R  = clamp(Q.P^T), BCE(sum) loss, dP = dR^T.Q, dQ = dR.P
//                                              (neural_admixture.py:94-97, :288/:431 + autograd)
//   pass 3  encode_bwd   dV = X^T . dZ         reduce over samples (autograd of :172)
//
// The batch is gathered by row index (idx) straight out of the packed matrix; genotypes are decoded
// in registers, never materialised (the reference unpacks to uint8 [b,M] and then to fp32 [b,M]
// every step: pack2bit.cu:38-62, neural_admixture.py:404-406,169-170).
//
// All three kernels stream X tiles HBM -> LDS with 16-byte-per-lane coalesced loads of whole row
// segments (>= 128 B contiguous per gathered row), software-prefetched one tile ahead.
#include "nadm_common.h"
#include "../../include/nadm.h"
#include "nadm_host.h"
extern "C" int nadm_mlp_bwd_weight_parts(const nadm_heads_t* hd, int32_t b, const float* Zn, const float* H, const float* dL,
                                         const float* dHpre, const float* dgp, float* small_part, void* stream);
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <atomic>

namespace nadm {

// =================================================================================================
// pass 1: Z partial = X[:, chunk] . V[chunk, :]        lane <-> sample, V rows wave-uniform
// =================================================================================================
constexpr int ENC_TB = 128;            // bytes of each gathered row per tile  (512 SNPs)
constexpr int ENC_TILES = 4;           // tiles per chunk -> 2048 SNPs per chunk, the chunking of the matrix-core kernel (EM_CHUNK_SNPS)
constexpr int ENC_LDW = ENC_TB / 4 + 1;  // LDS row stride in dwords (+1: lane r reads row r -> conflict-free)

template <int CP>
__global__ void encode_fwd_kernel(const uint8_t* __restrict__ xp, int64_t ld, const int32_t* __restrict__ idx,
                                  int b, int64_t M, const float* __restrict__ V, float* __restrict__ zpart,
                                  int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [rows_per_block][ENC_LDW]
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;                  // == rows_per_block
    const int64_t chunk = blockIdx.x;
    const int row0 = blockIdx.y * rows_per_block;
    const int nrows = min(rows_per_block, b - row0);
    const int my_row = row0 + tid;
    constexpr int PPR = ENC_TB / 16;              // 16-byte pieces per row segment
    constexpr int MAXP = 8;                       // pieces per thread per tile (rows*PPR/nthr == PPR)
    static_assert(PPR == MAXP, "one row's worth of pieces per thread");

    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;

    uint4 stage[MAXP];
    auto issue = [&](int tile) {
        const int64_t byte0 = (chunk * ENC_TILES + tile) * ENC_TB;
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int piece = tid + p * nthr;       // consecutive lanes -> consecutive 16 B of one row
            const int r = piece / PPR, c16 = piece % PPR;
            uint4 v = make_uint4(0, 0, 0, 0);
            const int64_t off = byte0 + c16 * 16;
            if (r < nrows && off * 4 < M) {
                const int64_t row = idx[row0 + r];
                v = *reinterpret_cast<const uint4*>(xp + row * ld + off);
            }
            stage[p] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int piece = tid + p * nthr;
            const int r = piece / PPR, c16 = piece % PPR;
            uint32_t* d = lds + r * ENC_LDW + c16 * 4;
            d[0] = stage[p].x; d[1] = stage[p].y; d[2] = stage[p].z; d[3] = stage[p].w;
        }
    };

    issue(0);
    for (int tile = 0; tile < ENC_TILES; ++tile) {
        __syncthreads();                 // previous tile's readers are done
        commit();
        __syncthreads();
        if (tile + 1 < ENC_TILES) issue(tile + 1);
        const int64_t snp0 = (chunk * ENC_TILES + tile) * (ENC_TB * 4);
        if (snp0 < M) {
            const uint32_t* myrow = lds + tid * ENC_LDW;
            const int nd = (int)min<int64_t>(ENC_TB / 4, (M - snp0 + 15) / 16);
            for (int d = 0; d < nd; ++d) {
                const uint32_t bits = myrow[d];
                const float* vrow = V + (snp0 + d * 16) * CP;        // wave-uniform address
                const int ns = (int)min<int64_t>(16, M - (snp0 + d * 16));
                if (ns == 16) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const float x = decode_x((bits >> (2 * s)) & 3u);
#pragma unroll
                        for (int c = 0; c < CP; ++c) acc[c] = fmaf(x, vrow[s * CP + c], acc[c]);
                    }
                } else {
                    for (int s = 0; s < ns; ++s) {
                        const float x = decode_x((bits >> (2 * s)) & 3u);
#pragma unroll
                        for (int c = 0; c < CP; ++c) acc[c] = fmaf(x, vrow[s * CP + c], acc[c]);
                    }
                }
            }
        }
    }
    if (my_row < b) {
        float* o = zpart + (chunk * b + my_row) * CP;
#pragma unroll
        for (int c = 0; c < CP; c += 4)
            *reinterpret_cast<float4*>(o + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    }
}

// =================================================================================================
// pass 1, matrix-core version (CP <= 8): Z = X.V on v_mfma_f32_16x16x32_bf16.
//   X in {0,.5,1} is exact in bf16; V (fp32) is split into three bf16 pieces hi+mid+lo (24 mantissa
//   bits, so the products are exact and only the fp32 accumulation rounds, like the fp32 kernels).
//   The 16 MFMA output columns hold [V_hi(8) | V_mid(8)], a second MFMA holds [V_lo(8) | 0]; the three
//   column groups are added at the end of each 16-sample tile.
//   A operand = X tile [16 samples x 32 SNPs]: lane (i = l&15, q = l>>4) needs 8 consecutive SNPs of
//   sample i = 16 packed bits.  Each lane loads ONE 16-byte piece of its gathered row per sample
//   tile (64 genotypes = the A operands of 8 k-steps; the k-step <-> SNP mapping is chosen so those
//   bytes are contiguous) straight from HBM -- no LDS staging, no block barrier on the load path --
//   and expands nibbles (2 genotypes) to bf16 pairs through a 16-entry, conflict-free LDS table.
//   block = 8 waves, each owning a 256-SNP slice (its V operands stay in 64 VGPRs for the whole
//   block); grid.y splits the batch.  Partial Z of the 8 waves is combined through LDS per sample
//   tile (one barrier) and written as one [b,CP] slab per 2048-SNP chunk.
// =================================================================================================
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int EM_WAVES = 8;
constexpr int EM_SLICE = 256;                         // SNPs per wave
constexpr int EM_CHUNK_SNPS = EM_WAVES * EM_SLICE;    // 2048
constexpr int EM_TILES_PER_BLOCK = 52;                // upper bound of 16-sample tiles per block (grid.y splits the batch)
#ifndef NADM_EM_D
#define NADM_EM_D 4
#endif
constexpr int EM_D = NADM_EM_D;                       // X tiles in flight per lane

// byte `sel` of w holds two nibbles 00cc: -> the bf16 pair (c_lo/2, c_hi/2) in one instruction
__device__ __forceinline__ uint32_t fp4_bf16_pair(const uint32_t w, const int sel) {
    typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
    bf16x2_native r;
    switch (sel) {
        case 0: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, 1.0f, 0); break;
        case 1: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, 1.0f, 1); break;
        case 2: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, 1.0f, 2); break;
        default: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, 1.0f, 3); break;
    }
    return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ uint32_t bf16_trunc_bits(float v) { return __float_as_uint(v) & 0xFFFF0000u; }
// (a >> 16) | (b & 0xFFFF0000): the bf16 truncations of a and b as one packed pair, a in the low half
__device__ __forceinline__ uint32_t upper_halves(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }

// The sum of the MLP weight-gradient partials of the PREVIOUS step + Adam on the small parameters (what nadm_small_grads does
// as a launch of its own: 4.7 us + a launch gap at the end of every step) as side blocks of pass 1: pass 1 reads none of the
// small parameters, the MLP forward behind it reads all of them.  part == nullptr: no side blocks.
struct SmallSide {
    const float* part;
    float *out, *p, *m, *v;
    int splits, n;
    float step_size, inv_bc2, grad_scale;
};

// block id -> work item such that the blocks the dispatcher places on one XCD (id % 8) get consecutive items
__device__ __forceinline__ int64_t xcd_chunk(const unsigned bid, const unsigned nwg) {
    constexpr unsigned NX = 8;
    const unsigned q = nwg / NX, r = nwg % NX, x = bid % NX;
    return (int64_t)(x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / NX;
}

template <int CP>
__global__ __launch_bounds__(512) void encode_fwd_mfma_kernel(const uint8_t* __restrict__ xp, int64_t ld,
                                                              const int32_t* __restrict__ idx, int b, int64_t M,
                                                              const float* V, float* __restrict__ zpart, int tiles_per_block,
                                                              uint32_t missing_bf16, int n_chunks, int n_splits, SmallSide ss) {
    static_assert(CP <= 8, "two MFMA column groups hold hi|mid and lo|0");
    // 1-D grid: block = chunk + n_chunks * batch split (the dispatch order a 2-D grid would have), side blocks LAST: they run in
    // the slots the partly filled last round leaves empty instead of pushing main blocks of the first round back
    const int n_main = n_chunks * n_splits;
    if ((int)blockIdx.x >= n_main) {
        const int e = ((int)blockIdx.x - n_main) * 512 + (int)threadIdx.x;
        if (e >= ss.n) return;
        float mq = 0.f, vq = 0.f, pq = 0.f;
        if (ss.m != nullptr) { mq = ss.m[e]; vq = ss.v[e]; pq = ss.p[e]; }
        float a = 0.f;                                       // the order of additions of sum_splits() (nadm_small_kernels.hip)
        for (int j0 = 0; j0 < ss.splits; j0 += 32) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = ss.part[(int64_t)(j0 + u < ss.splits ? j0 + u : ss.splits - 1) * ss.n + e];
#pragma unroll
            for (int u = 0; u < 32; ++u) if (j0 + u < ss.splits) a += v[u];
        }
        ss.out[e] = a;
        if (ss.m != nullptr) {
            ss.p[e] = adam_element(pq, a, mq, vq, ss.step_size, ss.inv_bc2, ss.grad_scale, false);
            ss.m[e] = mq; ss.v[e] = vq;
        }
        return;
    }
    static_assert(EM_D * 128 == 64 * EM_WAVES, "one output element per thread in the cross-wave combine");
    // one LDS allocation, two lives: the prologue's image of the block's V rows (s_v), then the partial sums of the main loop
    constexpr int SV_WAVE = EM_SLICE * 8 + (EM_SLICE / 64) * 8;        // floats per wave: [256 SNPs][8] + 8 floats of skew per 64 SNPs
    constexpr int SZ_FLOATS = 2 * EM_WAVES * EM_D * 128, SOUT_FLOATS = EM_TILES_PER_BLOCK * 128;
    constexpr int S_FLOATS = EM_WAVES * SV_WAVE > SZ_FLOATS + SOUT_FLOATS ? EM_WAVES * SV_WAVE : SZ_FLOATS + SOUT_FLOATS;
    __shared__ __attribute__((aligned(16))) float s_raw[S_FLOATS];
    float (*const s_z)[EM_WAVES][EM_D][16 * 8] = reinterpret_cast<float (*)[EM_WAVES][EM_D][16 * 8]>(&s_raw[0]);
    float (*const s_out)[16 * 8] = reinterpret_cast<float (*)[16 * 8]>(&s_raw[SZ_FLOATS]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    // block -> (chunk, batch split), XCD-aware (r04): the blocks i and i + 8 -- same XCD, dispatched back to back -- take the batch
    // splits of ONE chunk, so the second one finds the chunk's V rows in that XCD's L2 instead of fetching them again
    const int64_t vwork = xcd_chunk(blockIdx.x, (unsigned)n_main);
    const int64_t chunk = vwork / n_splits;
    const int64_t slice0 = chunk * EM_CHUNK_SNPS + wave * EM_SLICE;
    const int tile_begin = (int)(vwork % n_splits) * tiles_per_block;
    const int tile_end = min((b + 15) / 16, tile_begin + tiles_per_block);
    // A missing call (code 3) is 0 in the model (neural_admixture.py:170) and 1.5 = bf16 0x3FC0 in the init-time PCA
    // projection (train.py:52), selected by the caller.  FP4 (E2M1) reads the nibble 00cc as c/2 -- 0, 0.5, 1, 1.5 -- so
    // the 1.5 case is the conversion's native result and the 0 case clears both bits of every code 3 first.
    const uint32_t kmiss = missing_bf16 == 0u ? 0x55555555u : 0u;
    // ---- B operands: V rows of this wave's slice, split hi/mid/lo, for the 8 k-steps ----
    // k-step s, lane (q, j): rows kk = 8q + e  <->  SNP slice0 + 64q + 8s + e ; column j: c = j & 7
    // The wave's [256 x CP] slice of V goes through LDS: CP full 16 B / lane loads per lane (r02 read it element by element --
    // 64 dword loads per thread whose lanes touch 4 different sectors each: 4.5 us of the kernel's 41, profiles/r03_ablations.txt).
    // Row m of the slice sits at float offset 8 m + 8 (m >> 6): the skew makes the column reads below (lanes = 4 row groups x 8
    // columns, rows 64 apart) hit 32 different banks.
    bf16x8 b1[8], b2[8];
    {
        float* const sv = &s_raw[wave * SV_WAVE];
        float4 vst[CP];
        // rows past M are read from row M - 1 (clamped, unconditional) and zeroed at the LDS store; the clamp as ONE 32-bit minimum per
        // load (a 64-bit "m < M ? m : M - 1" is a compare + two v_cndmask with the mask in VCC: 23 cycles each, nadm_common.h)
        const int row_lim = (int)(M - 1 - slice0 < (int64_t)(EM_SLICE - 1) ? M - 1 - slice0 : (int64_t)(EM_SLICE - 1));
#pragma unroll
        for (int j = 0; j < CP; ++j) {                        // unconditional, clamped; masked at the LDS store
            const int e = 4 * (lane + 64 * j);
            vst[j] = *reinterpret_cast<const float4*>(V + (slice0 + min(e / CP, row_lim)) * CP + e % CP);
        }
#pragma unroll
        for (int j = 0; j < CP; ++j) {
            const int e = 4 * (lane + 64 * j), ml = e / CP;
            const uint32_t in = lt_mask64(slice0 + ml, M);     // (a ?: here is four v_cndmask with the mask in VCC per row piece: 23 cycles each, nadm_common.h)
            *reinterpret_cast<float4*>(sv + 8 * ml + 8 * (ml >> 6) + e % CP) = make_float4(keepf(vst[j].x, in), keepf(vst[j].y, in), keepf(vst[j].z, in), keepf(vst[j].w, in));
        }
        __builtin_amdgcn_wave_barrier();                      // the slice is written and read by this wave only (LDS operations of a wave complete in order)
        const int c = i & 7;
        const uint32_t upper = lt_mask(7, i);                 // lanes i >= 8 hold the mid pieces: blends, not selects
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            uint32_t w1[4], w2[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // two values -> their hi / mid / lo pieces (truncating split: every remainder exact), packed two by two with ONE v_perm_b32
                // per piece (the upper halves of two registers side by side) instead of two shifts and an or
                float v[2], r1[2], r2[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int ml = 64 * q + 8 * s8 + 4 * (d & 1) + 2 * hh + (d >> 1);                // element order of the A operand
                    v[hh] = c < CP ? sv[8 * ml + 8 * q + c] : 0.f;                                   // (rows past M hold zeros)
                    r1[hh] = v[hh] - __uint_as_float(bf16_trunc_bits(v[hh]));
                    r2[hh] = r1[hh] - __uint_as_float(bf16_trunc_bits(r1[hh]));
                }
                const uint32_t hi = upper_halves(v[0], v[1]), mid = upper_halves(r1[0], r1[1]), lo = upper_halves(r2[0], r2[1]);
                w1[d] = blend(mid, hi, upper);
                w2[d] = lo & ~upper;
            }
            b1[s8] = __builtin_bit_cast(bf16x8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
            b2[s8] = __builtin_bit_cast(bf16x8, make_uint4(w2[0], w2[1], w2[2], w2[3]));
        }
    }
    __syncthreads();

    const int64_t byte_off = chunk * (EM_CHUNK_SNPS / 4) + wave * (EM_SLICE / 4) + 16 * q;
    const bool col_ok = byte_off * 4 < M;          // not `< ld`: on an SNP sub-range launch the bytes past M belong to the next range
    // all loads are unconditional with clamped addresses (a load inside a divergent branch makes the compiler wait for
    // every outstanding load, which would serialise the prefetch ring); invalid rows / columns are zeroed at use
    const int64_t byte_off_c = col_ok ? byte_off : 0;
    auto row_of = [&](int tile) -> int32_t {
        const int smp = tile * 16 + i;
        return idx[smp < b ? smp : b - 1];
    };
    // row index and row length as UNSIGNED 32-bit factors: one v_mad_u64_u32 per address (a signed 64 x 64 product is that + two
    // v_mul_lo_u32 + a sign extension, per tile and lane, in the chain in front of the gather's load; the launcher refuses rows of 4 GiB and more)
    const uint32_t ld32 = (uint32_t)ld;
    auto load_row = [&](int32_t row) -> uint4 { return *reinterpret_cast<const uint4*>(xp + ((uint64_t)(uint32_t)row * ld32 + (uint64_t)byte_off_c)); };
    // EM_D - 1 tiles of loads in flight ahead of the compute (a 16-sample tile is ~0.3 us of work per wave, much less
    // than the latency of the dependent idx -> row loads); the row index of the tile after those is fetched too.
    uint4 st[EM_D];
    int32_t rown[EM_D];                                       // row index ring: tiles tile+EM_D-1 .. tile+2*EM_D-2 (loads retire in order)
#pragma unroll
    for (int d = 0; d < EM_D - 1; ++d) rown[d] = row_of(tile_begin + d);
#pragma unroll
    for (int d = 0; d < EM_D - 1; ++d) {
        st[d] = load_row(rown[d]);
        rown[(d + EM_D - 1) % EM_D] = row_of(tile_begin + d + EM_D - 1);
    }

    for (int tile0 = tile_begin; tile0 < tile_end; tile0 += EM_D) {
        const int buf = ((tile0 - tile_begin) / EM_D) & 1;
#pragma unroll
        for (int u = 0; u < EM_D; ++u) {
            const int tile = tile0 + u;
            if (tile < tile_end) {                                // block-uniform
                st[(u + EM_D - 1) % EM_D] = load_row(rown[(u + EM_D - 1) % EM_D]);
                rown[(u + 2 * EM_D - 2) % EM_D] = row_of(tile + 2 * EM_D - 2);
                const bool ok = col_ok && (tile * 16 + i < b);
                const uint4 cur = st[u];
                const uint32_t raw[4] = {ok ? cur.x : 0u, ok ? cur.y : 0u, ok ? cur.z : 0u, ok ? cur.w : 0u};    // (wave-level mask in an SGPR pair: cheaper here than building a lane mask per tile, measured)
                f32x4_t d1 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, d2 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // 16 genotypes of the word -> two A operands.  Nibble-spread the 2-bit codes (even SNPs, odd SNPs), then
                    // v_cvt_scalef32_pk_bf16_fp4 turns one byte = two nibbles into a bf16 pair: 8 + 16 VALU instructions
                    // per 16 genotypes, no LDS table.  Element order within a k-step: SNPs 0 2 4 6 1 3 5 7 (B matches).
                    uint32_t r = raw[w];
                    const uint32_t m3 = r & (r >> 1) & kmiss;                 // low bit of every code that is 3
                    r ^= m3 | (m3 << 1);
                    const uint32_t ev = r & 0x33333333u, od = (r >> 2) & 0x33333333u;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int s8 = 2 * w + hf;
                        const bf16x8 av = __builtin_bit_cast(bf16x8, make_uint4(fp4_bf16_pair(ev, 2 * hf), fp4_bf16_pair(ev, 2 * hf + 1),
                                                                                fp4_bf16_pair(od, 2 * hf), fp4_bf16_pair(od, 2 * hf + 1)));
                        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b1[s8], d1, 0, 0, 0);
                        d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b2[s8], d2, 0, 0, 0);
                    }
                }
                // D rows = samples 4q + r, column = i: fold [hi | mid] + [lo | 0] -> columns 0..7
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float u2 = d1[r] + d2[r];
                    u2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u2), 0x128 /*row_ror:8*/, 0xf, 0xf, false));
                    if (i < 8) s_z[buf][wave][u][(4 * q + r) * 8 + i] = u2;
                }
            }
        }
        // one barrier per EM_D tiles: thread -> (tile u = tid >> 7, element tid & 127) sums the 8 waves' partials; the
        // result stays in LDS and is stored after the loop (a global store in this loop would make later waits drain
        // every load in flight)
        __syncthreads();
        {
            const int u = tid >> 7, e = tid & 127;
            if (tile0 + u < tile_end) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < EM_WAVES; ++w) sum += s_z[buf][w][u][e];
                s_out[tile0 + u - tile_begin][e] = sum;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < (tile_end - tile_begin) * 128; e += 512) {
        const int smp = tile_begin * 16 + (e >> 3), c = e & 7;
        if (smp < b && c < CP) zpart[(chunk * b + smp) * CP + c] = (&s_out[0][0])[e];
    }
}

// =================================================================================================
// X tile loader shared by passes 2 and 3: [TS rows] x [RB bytes] into LDS, 256 threads
// =================================================================================================
constexpr int TS = 32;   // samples per tile

template <int RB>
struct TileLoader {
    static constexpr int PPR = RB / 16;                 // 16 B pieces per row
    static constexpr int NP = (TS * PPR + 255) / 256;   // pieces per thread
    uint4 stage[NP];
    __device__ __forceinline__ void issue(const uint8_t* __restrict__ xp, int64_t ld, const int32_t* __restrict__ idx,
                                          int i0, int b, int64_t byte0, int tid, int64_t M) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int piece = tid + p * 256;
            const int r = piece / PPR, c16 = piece % PPR;
            uint4 v = make_uint4(0, 0, 0, 0);
            const int64_t off = byte0 + c16 * 16;
            if (piece < TS * PPR && i0 + r < b && off * 4 < M) {
                const int64_t row = idx[i0 + r];
                v = *reinterpret_cast<const uint4*>(xp + row * ld + off);
            }
            stage[p] = v;
        }
    }
    __device__ __forceinline__ void commit(uint8_t* buf, int tid) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int piece = tid + p * 256;
            if (piece < TS * PPR) *reinterpret_cast<uint4*>(buf + piece * 16) = stage[p];   // row-major [TS][RB]
        }
    }
};

// =================================================================================================
// pass 2: decoder + BCE forward/backward for one head.   lane <-> SPL consecutive SNPs
//   block = 256 threads = 4 waves, each wave owns 64*SPL SNPs and walks all b samples;
//   dP stays in registers for the whole block (written once), dQ partials are reduced over the
//   wave (DPP) per sample, over the 4 waves through LDS per 32-sample tile, and written as one
//   [b, KP] slab per chunk (row stride KP; deterministic; reduced over chunks by mlp_bwd).
// =================================================================================================
template <int KP, int SPL, bool LOSS>
__global__ __launch_bounds__(256) void decode_bce_kernel(
    const uint8_t* __restrict__ xp, int64_t ld, const int32_t* __restrict__ idx, int b, int64_t M,
    const float* __restrict__ P, const float* __restrict__ Q, int SP,
    float* __restrict__ dP, float* __restrict__ dqpart, float* __restrict__ losspart) {
    constexpr int RB = 256 * SPL / 4;                    // bytes of each row per tile
    __shared__ __attribute__((aligned(16))) uint8_t s_x[2][TS * RB];
    __shared__ __attribute__((aligned(16))) float s_q[2][TS * KP];
    __shared__ __attribute__((aligned(16))) float s_t[4][TS * KP];
    __shared__ float s_loss[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t chunk = blockIdx.x;
    const int64_t byte0 = chunk * RB;
    const int64_t m0 = (chunk * 256 + tid) * SPL;

    float p[SPL][KP], dp[SPL][KP];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const bool ok = (m0 + j) < M;
#pragma unroll
        for (int k = 0; k < KP; k += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(P + (m0 + j) * KP + k);
            p[j][k] = v.x; p[j][k + 1] = v.y; p[j][k + 2] = v.z; p[j][k + 3] = v.w;
            dp[j][k] = dp[j][k + 1] = dp[j][k + 2] = dp[j][k + 3] = 0.f;
        }
    }
    float lossacc = 0.f;

    TileLoader<RB> loader;
    auto load_q = [&](int i0, int buf) {       // Q tile [TS][KP] -> LDS (coalesced)
        for (int e = tid; e < TS * KP; e += 256) {
            const int r = e / KP, k = e % KP;
            s_q[buf][e] = (i0 + r < b) ? Q[(int64_t)(i0 + r) * SP + k] : 0.f;
        }
    };

    const int ntiles = (b + TS - 1) / TS;
    loader.issue(xp, ld, idx, 0, b, byte0, tid, M);
    loader.commit(s_x[0], tid);
    load_q(0, 0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const int i0 = t * TS;
        const int nt = min(TS, b - i0);
        if (t + 1 < ntiles) loader.issue(xp, ld, idx, i0 + TS, b, byte0, tid, M);

        for (int ii = 0; ii < nt; ++ii) {
            uint32_t bits;
            if constexpr (SPL == 8) bits = *reinterpret_cast<const uint16_t*>(&s_x[cur][ii * RB + tid * 2]);
            else bits = (uint32_t)s_x[cur][ii * RB + (tid * SPL) / 4] >> (((tid * SPL) & 3) * 2);
            float q[KP], tq[KP];
#pragma unroll
            for (int k = 0; k < KP; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(&s_q[cur][ii * KP + k]);   // LDS broadcast
                q[k] = v.x; q[k + 1] = v.y; q[k + 2] = v.z; q[k + 3] = v.w;
                tq[k] = tq[k + 1] = tq[k + 2] = tq[k + 3] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const float x = decode_x((bits >> (2 * j)) & 3u);
                float rr = 0.f;
#pragma unroll
                for (int k = 0; k < KP; ++k) rr = fmaf(q[k], p[j][k], rr);
                float r;
                const float dR = bce_grad(rr, x, r);
                if constexpr (LOSS) lossacc += bce_loss(r, x);
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    dp[j][k] = fmaf(dR, q[k], dp[j][k]);
                    tq[k] = fmaf(dR, p[j][k], tq[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float s = wave_sum_lane63(tq[k]);
                if (lane == 63) s_t[wave][ii * KP + k] = s;
            }
        }
        __syncthreads();                       // s_t complete; everyone done reading s_x[cur], s_q[cur]
        for (int e = tid; e < nt * KP; e += 256) {
            const float s = (s_t[0][e] + s_t[1][e]) + (s_t[2][e] + s_t[3][e]);
            const int r = e / KP, k = e % KP;
            dqpart[(chunk * b + i0 + r) * KP + k] = s;
        }
        if (t + 1 < ntiles) {
            loader.commit(s_x[cur ^ 1], tid);
            load_q(i0 + TS, cur ^ 1);
        }
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        if ((m0 + j) < M) {
#pragma unroll
            for (int k = 0; k < KP; k += 4)
                *reinterpret_cast<float4*>(dP + (m0 + j) * KP + k) =
                    make_float4(dp[j][k], dp[j][k + 1], dp[j][k + 2], dp[j][k + 3]);
        }
    }
    if constexpr (LOSS) {
        const float s = wave_sum_lane63(lossacc);
        if (lane == 63) s_loss[wave] = s;
        __syncthreads();
        if (tid == 0) losspart[chunk] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
    }
}

// =================================================================================================
// pass 3: dV = X^T . dZ     lane <-> 4 consecutive SNPs (one packed byte), dZ rows via LDS broadcast
// =================================================================================================
template <int CP>
__global__ __launch_bounds__(256) void encode_bwd_kernel(
    const uint8_t* __restrict__ xp, int64_t ld, const int32_t* __restrict__ idx, int b, int64_t M,
    const float* __restrict__ dZ, float* __restrict__ dV) {
    constexpr int SPL = 4;
    constexpr int RB = 256;
    __shared__ __attribute__((aligned(16))) uint8_t s_x[2][TS * RB];
    __shared__ __attribute__((aligned(16))) float s_z[2][TS * CP];
    const int tid = threadIdx.x;
    const int64_t chunk = blockIdx.x;
    const int64_t byte0 = chunk * RB;
    const int64_t m0 = (chunk * 256 + tid) * SPL;

    float acc[SPL][CP];
#pragma unroll
    for (int j = 0; j < SPL; ++j)
#pragma unroll
        for (int c = 0; c < CP; ++c) acc[j][c] = 0.f;

    TileLoader<RB> loader;
    auto load_z = [&](int i0, int buf) {
        for (int e = tid; e < TS * CP; e += 256) {
            const int r = e / CP;
            s_z[buf][e] = (i0 + r < b) ? dZ[(int64_t)i0 * CP + e] : 0.f;
        }
    };
    const int ntiles = (b + TS - 1) / TS;
    loader.issue(xp, ld, idx, 0, b, byte0, tid, M);
    loader.commit(s_x[0], tid);
    load_z(0, 0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const int i0 = t * TS;
        const int nt = min(TS, b - i0);
        if (t + 1 < ntiles) loader.issue(xp, ld, idx, i0 + TS, b, byte0, tid, M);
        for (int ii = 0; ii < nt; ++ii) {
            const uint32_t bits = s_x[cur][ii * RB + tid];
            float z[CP];
#pragma unroll
            for (int c = 0; c < CP; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(&s_z[cur][ii * CP + c]);
                z[c] = v.x; z[c + 1] = v.y; z[c + 2] = v.z; z[c + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const float x = decode_x((bits >> (2 * j)) & 3u);
#pragma unroll
                for (int c = 0; c < CP; ++c) acc[j][c] = fmaf(x, z[c], acc[j][c]);
            }
        }
        if (t + 1 < ntiles) {
            loader.commit(s_x[cur ^ 1], tid);
            load_z(i0 + TS, cur ^ 1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        if ((m0 + j) < M) {
#pragma unroll
            for (int c = 0; c < CP; c += 4)
                *reinterpret_cast<float4*>(dV + (m0 + j) * CP + c) =
                    make_float4(acc[j][c], acc[j][c + 1], acc[j][c + 2], acc[j][c + 3]);
        }
    }
}


typedef float f32x4 __attribute__((ext_vector_type(4)));

// Shape of the bf16 matrix-core pass-2 kernel (KP <= 8), tuned on MI355X with bench.py (b=800, M=500k, K=8; decode time
// with / without the loss):  8 waves x 4 tiles, occupancy 2: 435 / 373 us;  8 x 2, occ. 4: 401 / 331;  4 x 2, occ. 4:
// 400 / 334;  4 waves x 4 tiles at <= 168 VGPRs (3 blocks = 3 waves per SIMD, 42 KB LDS each): 388 / 310  <- default.
#ifndef NADM_BF_WAVES
#define NADM_BF_WAVES 4      // waves per block; chunk = WAVES * NTW * 16 SNPs
#endif
#ifndef NADM_BF_NTW
#define NADM_BF_NTW 4        // 16-SNP tiles per wave (2 or 4)
#endif
#ifndef NADM_BF_TS
#define NADM_BF_TS 64        // samples per LDS tile
#endif
#ifndef NADM_BF_WPE
#define NADM_BF_WPE 3        // waves per SIMD the register allocator must leave room for (K <= 8)
#endif
#ifndef NADM_BF_WPE_W
#define NADM_BF_WPE_W 2      // ... for K 9..16: 7 MFMAs per tile and three more resident operands want 218 VGPRs; at 168 (3 waves) the
#endif                       // kernel ran 721 us (K = 16, M = 1M), at 2 waves per SIMD 652, and 629 with the pair-product loss (r03)
constexpr int bf_wpe(int kp) { return kp > 8 ? NADM_BF_WPE_W : NADM_BF_WPE; }
constexpr int mf_waves(int kp) { return NADM_BF_WAVES; }   // waves per block
constexpr int MF_RS_PAD = 16;       // LDS row stride of the X tile = row bytes + 16 (16 B aligned, de-phased banks)
constexpr int mf_ntw(int kp) { return NADM_BF_NTW; }            // 16-SNP tiles per wave
constexpr int mf_chunk_snps(int kp) { return mf_waves(kp) * 16 * mf_ntw(kp); }

// =================================================================================================
// pass 2, bf16 matrix-core version (KP <= 8).  f32-input MFMA runs on the vector ALUs and does not overlap
// with VALU work (tools/ubench_mfma_valu*.hip), so the three small GEMMs are moved to the real matrix pipe
// (v_mfma_f32_16x16x32_bf16, 16x the f32 rate) with fp32 operands split into bf16 pieces:
//   R^T = P.Q^T      P, Q split 3-way (hi+mid+lo = 24 mantissa bits); the 6 product terms >= 2^-16 fill the
//                    K = 32 dimension of two MFMAs: [Ph Ph Pm Pm].[Qh Qm Qh Qm] and [Ph Pl 0 0].[Ql Qh 0 0]
//   dQ^T = P^T.dR^T  dR split 2-way with round-to-nearest (v_cvt_pk_bf16_f32; 16-17 bits, unbiased), P with its hi
//                    and mid pieces in the 16 output rows [Ph|Pm] (a third piece would be below dR's own 2^-18
//                    truncation); reduction = 32 SNPs = 2 tiles, the lane's own 8 dR values ARE the B operand
//   dP = dR^T.Q      same dR pieces, transposed through a 2 KB per-wave LDS buffer and read back with
//                    ds_read_b64_tr_b16; Q hi|mid in the 16 output columns; reduction = 32 samples = 2 tiles
// The VALU only does the per-genotype BCE algebra and the bf16 split.  Work unit of a wave: 2 SNP tiles x
// 2 sample tiles.  Everything else (tile staging, chunking, dQ partial slabs) is as in the f32 MFMA kernel.
// =================================================================================================
typedef short s16x4_t __attribute__((ext_vector_type(4)));
// (pk_bf16 / split3: nadm_common.h, shared with the producer of the Q operand images)
// the same for TWO values at once: one v_cvt_pk_bf16_f32 per piece serves both, and the three results are already the
// packed (v0 | v1 << 16) dwords the MFMA operands are built from -- half the instructions of two split3 calls
__device__ __forceinline__ void split3_pair(float v0, float v1, uint32_t& H, uint32_t& Md, uint32_t& Lo) {
    H = pk_bf16(v0, v1);
    const f32x2_t r1 = (f32x2_t){v0, v1} - (f32x2_t){__uint_as_float(H << 16), __uint_as_float(H & 0xFFFF0000u)};
    Md = pk_bf16(r1.x, r1.y);
    const f32x2_t r2 = r1 - (f32x2_t){__uint_as_float(Md << 16), __uint_as_float(Md & 0xFFFF0000u)};
    Lo = pk_bf16(r2.x, r2.y);
}
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// Two genotypes at a time with packed f32 math.  Issue cost on gfx950 with a saturated SIMD (tools/ubench_valu_asm.hip, cycles per
// wave64 instruction): v_add / v_sub / v_mul / v_fma / v_and ~2.9; v_max / v_min / v_med3 / v_max3 / shifts / v_perm / conversions
// ~4.4; v_pk_{add,mul,fma}_f32 ~4.8 (two results); v_rcp / v_log ~8.3; a 16x16x32 bf16 MFMA ~17 and NOT hidden behind other
// waves' VALU work (tools/ubench_issue.hip).  max, rcp, log and the conversions have no packed form.
// Returns dR (gradient w.r.t. the pre-clamp reconstruction) and accumulates the BCE loss terms; x = genotype/2 with missing
// already mapped to 0 (fp4_pair below).
//   gradient   den = d - d^2 (one fma) from the UNCLAMPED d: inside [0, 1] that is r(1 - r) to an ulp (r == d there), 0 exactly at
//              d = 0 / 1, negative outside.  The reference divides by max(den, 1e-12) and its clamp_ backward zeroes the gradient
//              where the PRE-clamp value lies outside [0, 1] (bounds inclusive): inv = min(1/den, 1e12) for den >= 0, 0 for
//              den < 0.  The tile loop forms inv * 1e-12 = sat(1e-12 * rcp(den)) -- a v_mul_f32 carrying the clamp bit
//              (saturation to [0, 1]: +inf -> 1, negative and -inf -> 0; profiles/r03_ubench_clamp_mfma16.txt) -- and the factor
//              KMAX = rcp(1e-12) goes onto the dQ / dP sums once per block (scale_back below): dR' = (d - x) * inv * 1e-12 is
//              what the matrix pipe sees (bf16 / fp32 have the exponent range to spare: |dR'| >= 1e-19 for any d - x an fp32
//              subtraction can produce).  rcp is monotonic, so the saturation sets in exactly at den <= 1e-12.  The numerator
//              uses d itself: no clamp instruction on the gradient path at all.  (r02 formed max3(den, 1e-12, den * -inf) in
//              front of the reciprocal, r03 first med3(rcp(den), 0, KMAX) per value: -10 us; then this form: a v_med3 becomes
//              a v_mul of the cheap class, and 1 - d is no longer needed for den.)
//   loss       x*max(log r, -100) + (1-x)*max(log(1-r), -100), x in {0, .5, 1}.
//     exact    (bce_loss_exact2; the steps before the first restrict_P and the fallback below) needs the two clamps only for
//              r == 0 / r == 1 (log = -inf; an fp32 r is never in (0, e^-100)).  log2(v * 2^20 + 2^(20 - 100/ln2)) - 20 equals
//              log2 v for every normal v (the addend is below half an ulp of v * 2^20) and -100/ln2 for v == 0: one packed fma
//              per pair replaces two max, and since the weights x and 1-x add up to 1 the "- 20" is a constant per genotype that
//              the caller subtracts once per tile pair.  r for the logs: d is never negative (P >= 0, Q >= 0 and the dropped
//              split terms are 2^-24 relative); 1 - r needs the clamp from above only, and 2*max(v, 0) = v + |v| is one add
//              of the fast class (the factor 2 goes into the scale) instead of a v_max.  log r itself may use d as long as
//              P lies in [0, 1] (d <= 1 + a rounding error, log2 of that is < 2e-7): true from the first restrict_P on.
//              UNIT_P = false clamps d from above for the step(s) before that -- the reference's supervised run starts
//              from per-class means of the raw codes, which reach 2 (train.py:82).
//     fast     (r03, bce_loss_prod2; UNIT_P only) ONE logarithm per PAIR of genotypes instead of four.  With c = 2x in {0, 1, 2}
//              twice the term of a genotype is c*log d + (2-c)*log(1-d) = log f, f = (1-d)^2, d(1-d), d^2.  With o = sat(1 - d)
//              (a v_sub with the clamp bit) and q = o - x:  q = 1-d | 1/2 - d | -d  for c = 0 | 1 | 2, so q^2 IS f for the two
//              homozygous calls, and for the heterozygous one d(1-d) = 1/4 - (1/2 - d)^2 = 1/4 - q^2.  Since x(1-x) = 0 | 1/4 | 0,
//                  f = | q^2 - x(1-x) | = | fma(q, q, fma(x, x, -x)) |
//              for all three -- two packed fmas per pair behind q, no selector, nothing taken from the gradient's chain -- and
//              the two f of a pair are multiplied (the |.| are source modifiers of that v_mul) before the one v_log.  (r03-r06:
//              f = qq + h * (den - qq) with den = d - d^2 from the gradient and a selector h = [c == 1] converted from a masked
//              code word: one packed instruction + one conversion + 3/4 mask instruction more per pair and a dependence on the
//              gradient's den; the form above took pass 2 from 247-249 to 236-237 us, profiles/r06_abl_lossabs.txt.)
//              o is clamped: with d > 1 by a rounding error a homozygous-reference genotype gives f = 0 like the reference's
//              log(1 - clamp(d)), the heterozygous one q = -1/2 and f = 0 as well, the homozygous-alternative one f = 1 (log 1);
//              d = 0 or 1 exactly gives f = 0 under a call whose reference term is a clamped logarithm.  The product is 0 or
//              underflows exactly when a clamp of the reference could be active or d is tiny -- then, and only then, the
//              logarithm is -inf and the wave recomputes the loss of that tile pair with the exact form (cold branch in the loop,
//              decode_bce_bf16_kernel; a NaN among the inputs takes the same branch).
//              Accuracy: the factor under a non-zero genotype carries the ABSOLUTE rounding error of 1 - d (3e-8 for d < 1/2),
//              i.e. 3e-8 / d relative (heterozygous: the same 3e-8 on d(1-d), where den had an ulp) -- against the fp32
//              accumulation of ~4e8 terms this is invisible in the sum (tests: 2e-6 against the exact form, 5e-6 against the
//              oracle), but a single term under d = 1e-6 is only good to 3 %; the exact form has no such error.  (First r03
//              version: u = o + s1 * (d - o) with two selectors and the product u * v: three instructions more per pair and
//              twice that error.  Multiplying the products of the lane's two pairs of a tile as well -- one logarithm per four
//              genotypes -- measured 250 us against 246 in r03 and 233-237 against 236-237 on the form above: nothing for the
//              narrower underflow margin of a four-factor product.)
constexpr float LOSS_LOG_SHIFT = 20.f;                    // log2 of the scale of the exact form
__device__ __forceinline__ f32x2_t two_max0(const f32x2_t v) {                 // 2 * max(v, 0), exact.  As asm: the compiler would pair
    f32x2_t r;                                                                 // the two adds into a v_pk_add_f32, which has no |abs|
    asm("v_add_f32_e64 %0, %1, |%1|" : "=v"(r.x) : "v"(v.x));                  // modifier, behind two extra v_and
    asm("v_add_f32_e64 %0, %1, |%1|" : "=v"(r.y) : "v"(v.y));
    return r;
}
// gradient w.r.t. the pre-clamp reconstruction for two genotypes, TIMES 1e-12 (eps; the caller multiplies the sums by rcp(1e-12))
// (r05, profiles/r05_abl_p2scalar.txt: the same algebra with scalar v_*_f32 instead of v_pk_*_f32 -- MI355X_MICROARCH.md prices packed f32
// beside MFMAs at +22..26 cycles per instruction against two scalar ones -- is SLOWER here, 260 vs 239 us: twice the issue slots cost more
// than the penalty saves at three waves per SIMD.  The arm is gone from the source.)
__device__ __forceinline__ f32x2_t bce_grad2(const f32x2_t d, const f32x2_t x, const float eps) {
    const f32x2_t den = __builtin_elementwise_fma(-d, d, d);                                                    // d - d^2 in one rounding; < 0 exactly when d is outside [0, 1]
    // sat(1e-12 / den): 1 at den <= 1e-12 (and +0), 0 for den < 0.  Written per element so that the clamp folds into the multiply
    // (v_mul_f32_e64 ... clamp); as inline asm (v_pk_mul_f32 ... clamp) it would sit right behind the v_rcp without the wait state
    // the hardware needs between a transcendental and its consumer -- the hazard recognizer does not look into asm statements
    const f32x2_t inv = {__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(den.x) * eps, 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(den.y) * eps, 0.f, 1.f)};
    return (d - x) * inv;
}
// exact form: adds x*log2(r') + (1-x)*log2((1-r)') + 20 per genotype to lossacc (packed halves)
template <bool UNIT_P>
__device__ __forceinline__ void bce_loss_exact2(const f32x2_t d, const f32x2_t omd, const f32x2_t x, f32x2_t& lossacc) {
    constexpr float kScale = 1048576.f;                                    // 2^20
    constexpr float kFloor = 3.900782386632024e-38f;                       // 2^20 * e^-100 (a normal number): log2 = 20 - 100/ln2
    const f32x2_t omr2 = two_max0(omd);
    f32x2_t dl = d;
    if constexpr (!UNIT_P) dl = (f32x2_t){__builtin_fminf(d.x, 1.f), __builtin_fminf(d.y, 1.f)};
    const f32x2_t a1 = __builtin_elementwise_fma(dl, (f32x2_t){kScale, kScale}, (f32x2_t){kFloor, kFloor});
    const f32x2_t a0 = __builtin_elementwise_fma(omr2, (f32x2_t){0.5f * kScale, 0.5f * kScale}, (f32x2_t){kFloor, kFloor});
    const f32x2_t l1 = {__builtin_amdgcn_logf(a1.x), __builtin_amdgcn_logf(a1.y)};
    const f32x2_t l0 = {__builtin_amdgcn_logf(a0.x), __builtin_amdgcn_logf(a0.y)};
    lossacc = __builtin_elementwise_fma(x, l1, lossacc);
    lossacc = __builtin_elementwise_fma((f32x2_t){1.f, 1.f} - x, l0, lossacc);
    asm volatile("" : "+v"(lossacc));     // pin the accumulation here: otherwise LLVM sinks all the logs of a tile pair
                                          // to the end of the loop body and keeps their 32 inputs alive (+60 VGPRs)
}
// fast form: adds log2(f0 * f1) = 2 * (the two genotypes' terms in log2 units) to acc; -inf (or NaN) flags the pair
__device__ __forceinline__ void bce_loss_prod2(const f32x2_t d, const f32x2_t x, float& acc) {
    const f32x2_t o = {__builtin_amdgcn_fmed3f(1.f - d.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - d.y, 0.f, 1.f)};   // 1 - r: v_sub_f32 ... clamp
    const f32x2_t q = o - x;                                                             // 1-r | 1/2 - r | -d  for c = 0 | 1 | 2
    const f32x2_t mh = __builtin_elementwise_fma(x, x, -x);                              // x^2 - x = -[c == 1] / 4
    const f32x2_t f = __builtin_elementwise_fma(q, q, mh);                               // (1-r)^2 | -r(1-r) | d^2
    acc += __builtin_amdgcn_logf(__builtin_fabsf(f.x) * __builtin_fabsf(f.y));
    asm volatile("" : "+v"(acc));             // pin the accumulation here (see bce_loss_exact2)
}

// missing calls (code 3) -> 0 in all 16 codes of a word: the model's input (neural_admixture.py:170)
__device__ __forceinline__ uint32_t clean_codes(uint32_t w) { return w & ~((w & (w >> 1) & 0x55555555u) * 3u); }
// the batch copy pass 2 leaves for pass 3 (nadm_decode_bce_gather): byte `col` of batch row `row` lives in tile col / 128 of
// [b rows][128 bytes]
constexpr int XG_TILE_COLS = 128;
__device__ __forceinline__ uint8_t* xg_piece(uint8_t* xg, int b, int row, int64_t col) {
    return xg + (col / XG_TILE_COLS) * ((int64_t)b * XG_TILE_COLS) + (int64_t)row * XG_TILE_COLS + (col % XG_TILE_COLS);
}

// Two genotypes -> two floats in ONE instruction: a nibble 00cc read as FP4 (E2M1) is exactly cc/2 (0, .5, 1), so
// v_cvt_scalef32_pk_f32_fp4 on byte `sel` of a word whose nibbles hold one 2-bit code each yields x for both.
__device__ __forceinline__ f32x2_t fp4_pair(const uint32_t w, const int sel) {
    switch (sel) {
        case 0: return __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 0);
        case 1: return __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 1);
        case 2: return __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 2);
        default: return __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 3);
    }
}

constexpr int BF_WAVES = NADM_BF_WAVES;
// floats one (sample slice, SNP chunk) block of pass 2 parks for the block that adds the slices up: the chunk's [SNPs x KP] partial of dP +
// its loss partial (+ 3 pad)
constexpr int p2_slab_floats(int kp) { return NADM_BF_WAVES * 16 * NADM_BF_NTW * kp + 4; }
constexpr int BF_NTW = NADM_BF_NTW;     // 16-SNP tiles per wave
constexpr int BF_TS = NADM_BF_TS;       // samples per LDS tile

// QIMG: the Q operands come ready-made from `qimg` (this head's tile images written by the MLP forward, nadm_common.h) instead
// of being split from the fp32 Q by every block: same bf16 pieces, same results.
// SLICED: the sample-slice form (gridDim.y slices, see the head of the body); false compiles the S = 1 kernel exactly as it stood before
// the slices existed (the headline shape's launch: on an A/B box the merged form cost it 1.2 %, profiles/r05_ablations.txt item 11)
// PROBE: the measurement build of the S = 1 form (nadm_clock_probe): launched only while a probe pointer is set, so that the kernel every
// other launch runs is instruction for instruction the one without it (as an always-present run-time branch the probe cost 0.8 %: an extra
// scalar load + wait in the prologue)
template <int KP, bool LOSS, bool UNIT_P = true, bool QIMG = false, bool SLICED = false, bool PROBE = false>
__global__ __launch_bounds__(64 * BF_WAVES) __attribute__((amdgpu_waves_per_eu(bf_wpe(KP), bf_wpe(KP)))) void decode_bce_bf16_kernel(
    const uint8_t* __restrict__ xp, int64_t ld, const int32_t* __restrict__ idx, int b, int64_t M,
    float* P, const float* __restrict__ Q, int SP,
    float* __restrict__ dP, float* __restrict__ dqpart, float* __restrict__ losspart, uint8_t* __restrict__ xg, AdamFused ad,
    const uint4* __restrict__ qimg, float* slab, int* slice_cnt, unsigned long long* clk) {
    static_assert(KP <= 16, "one or two 8-wide k slots");
    // measurement only (nadm_clock_probe, bench.py "box"): the block in the middle of the grid brackets itself with the shader-cycle counter
    // (s_memtime) and the constant-rate one (s_memrealtime): cycles / ticks x rate = the clock THIS kernel ran at in THIS run -- a denser
    // stream than any calibration kernel, it clocks lower on the same box, and by how much is the box's business (S = 1 form only)
    static_assert(!(PROBE && SLICED), "the probe brackets the S = 1 form");
    const bool probe = PROBE && clk != nullptr && blockIdx.x == (gridDim.x >> 1);
    unsigned long long pc0 = 0, pr0 = 0;
    if (probe) { pc0 = __builtin_readcyclecounter(); pr0 = __builtin_amdgcn_s_memrealtime(); }
    // gridDim.y = S sample SLICES (r05): with few SNP chunks (M below ~130k) a launch lasts as long as one block's serial chain over all the
    // sample tiles, not as long as the chip needs -- the batch's sample tiles are dealt to S blocks per chunk.  Everything a block writes is per sample (dQ slab rows, the batch copy) except dP and
    // the loss value: every slice parks its partial [chunk SNPs x KP] sum (+ its loss partial) in `slab`, is counted, and the block that
    // is counted LAST adds the S partials in slice order and runs the epilogue (Adam or the gradient store) -- the hand-off idiom of
    // the MLP backward's dZ image (nadm_small_kernels.hip: write-through stores, vmcnt(0), device-scope counter, device-scope loads;
    // DESIGN 4.3), no block ever waits for another.  The sum's order is fixed, so a step is reproducible bit for bit; it is NOT the
    // S = 1 kernel's order (one accumulator over all tiles), which is why the slicing is decided inside the library from (b, M) alone
    // (nadm_decode_slices) and every path to this kernel takes the same decision.
    // W (KP 9..16): k spans two 8-wide MFMA slots.  The pieces can no longer share an MFMA's 16 rows / columns, so
    //   R^T  = [Ph Ph' Ph Ph'].[Qh Qh' Qm Qm'] + [Pm Pm' Pm Pm'].[Qh Qh' Qm Qm'] + [Ph Ph' Pl Pl'].[Ql Ql' Qh Qh']   (X' = k 8..15)
    //   dQ^T = Ph.(dRh + dRl) + Pm.dRh    rows = the 16 k, three MFMAs (per two tiles) into one accumulator, no fold
    //   dP   = (dRh + dRl).Qh + dRh.Qm    columns = the 16 k, three MFMAs, no fold
    // = 6 MFMAs per 16 x 16 tile instead of 4; everything else is shared with the K <= 8 path.  (r05: the mid x lo products -- 2^-16 of
    // the product, at the level of dR's own 16-17 bits -- cost a seventh MFMA here, where the pieces cannot share an instruction's rows;
    // against float64 at b = 800, M = 500k, K = 16 the gradients err by 7.6e-7 of their maximum with them and 8.8e-7 without, for
    // 3 % of the kernel: profiles/r05_ablations.txt item 7.  K <= 8 carries them for free.)
    constexpr bool W = KP > 8;
    // ONE (KP <= 4, r06): the six product terms of R^T are 6 x 4 = 24 of an MFMA's 32 K values -- every 8-wide slot carries TWO pieces of
    // four k each and ONE instruction forms R^T:  [Ph|Ph  Pm|Pm  Ph|Pl  0].[Qh|Qm  Qh|Qm  Ql|Qh  0]   (3 MFMAs per tile instead of 4)
    constexpr bool ONE = KP <= 4;
    constexpr int KW = W ? 16 : 8;                       // k columns of the operand images
    static_assert(BF_NTW == 4 || BF_NTW == 2, "tile bits are read as one 32- or 16-bit word");
    constexpr int NTW = BF_NTW, MF_WAVES = BF_WAVES, MF_TS = BF_TS;
    constexpr int RB = MF_WAVES * 4 * NTW;               // 128 packed bytes per row per block
    constexpr int RS = RB + MF_RS_PAD;
    constexpr int PPR = RB / 16;
    constexpr int NTHR = 64 * MF_WAVES;
    __shared__ __attribute__((aligned(16))) uint8_t s_x[MF_TS * RS];
    // one contiguous image, laid out like a tile image of nadm_common.h: B operands of R^T per 16-sample tile, then of dP per 32-sample pair
    constexpr int IMG_U4 = (MF_TS / 16) * 2 * 64 + (MF_TS / 32) * (W ? 2 : 1) * 64;
    static_assert(!QIMG || (MF_TS == QI_TS && IMG_U4 == qi_tile_u4(KP)), "tile images of the MLP forward are 64 samples deep");
    __shared__ __attribute__((aligned(16))) uint4 s_qimg[IMG_U4];
    uint4 (*const s_qr)[2][64] = reinterpret_cast<uint4 (*)[2][64]>(&s_qimg[0]);
    uint4 (*const s_qd)[W ? 2 : 1][64] = reinterpret_cast<uint4 (*)[W ? 2 : 1][64]>(&s_qimg[(MF_TS / 16) * 2 * 64]);
    __shared__ __attribute__((aligned(16))) float s_dq[MF_WAVES][W ? 1 : 2][MF_TS * KP];   // K <= 8: [hi part | mid part] of the P operand, added up below
    // the transposition buffer: dense 32-byte rows (16 SNPs as bf16) with the four 8-byte chunks of row r rotated by (r >> 2) & 3 -- the
    // writes of a half-wave (16 rows x 2 chunks) then fall on every bank exactly twice (their minimum) AND the transposing reads (4
    // consecutive rows x 4 chunks per 16 lanes) on every bank once, which no padded row stride gives (reads want stride = 8 mod 32 dwords,
    // writes 2 x odd; r01's unswizzled dense rows: 4-way conflicts, 56 % of the LDS-busy cycles; padded 40-byte rows: the r02-r04 form)
    constexpr int TWS = 16, TWP = 32 * TWS;
    __shared__ __attribute__((aligned(16))) uint16_t s_t[MF_WAVES][2][2][TWP];  // per wave: [SNP tile of the pair][hi / lo][32 samples][16 SNPs (+ pad)]
    __shared__ float s_loss[MF_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = lane >> 4, n = lane & 15;
    // block -> chunk, XCD-aware: a block owns 64 bytes of every batch row, and the L2 fetches 128: with the dispatcher's round robin
    // (block i on XCD i % 8) the two halves of a line are fetched by two different L2s -- 2 x the payload on the fabric, 2.74 x with
    // unaligned rows (profiles/r04_pmc_req.json).  Here every XCD gets a contiguous range of chunks, so the blocks i and i + 8 -- same
    // XCD, dispatched back to back, walking the rows in the same order -- own the two halves of the same lines and the second one
    // hits.  A bijection for any grid size; placement is a matter of speed only (chunk is the index of everything the block writes).
    const int64_t chunk = xcd_chunk(blockIdx.x, gridDim.x);
    const int64_t byte0 = chunk * RB;
    const int64_t snp_wave0 = chunk * (MF_WAVES * 16 * NTW) + wave * (16 * NTW);
    // MFMA row 4*ab + r of tile t holds SNP code sigma(r) = {0,2,1,3}[r] of the lane group's byte t: registers (0,1) of an
    // accumulator are then the codes in the low/high nibble of (byte & 0x33), registers (2,3) those of ((byte >> 2) & 0x33)
    // -- the pairing fp4_pair() converts with one instruction.
    auto snp_of = [&](int t, int ab, int r) -> int64_t { return snp_wave0 + 4 * NTW * ab + 4 * t + (((r & 1) << 1) | (r >> 1)); };

    // zero the operand slots that are never written (k-slots 2,3 of the second R MFMA; columns 8..15 of the second dP MFMA)
    if constexpr (!QIMG)                     // (a ready-made image carries its zeros)
        for (int e = tid; e < IMG_U4; e += NTHR) s_qimg[e] = make_uint4(0, 0, 0, 0);

    // ---- the block's P rows [chunk SNPs x KP] -> LDS, once, with full 16 B / lane lines (the transposition buffers are free until
    // the first tile); the operands below are built from that image.  Read element by element from global memory -- 48 dword
    // loads per thread whose lanes each touch another sector -- the prologue took 16-27 k cycles of a block's ~215 k (s_memtime
    // probe), most of it memory latency.
    static_assert(sizeof(s_t) >= (size_t)MF_WAVES * 16 * NTW * KP * sizeof(float), "the P image fits the transposition buffers");
    float* const s_p = reinterpret_cast<float*>(&s_t[0][0][0][0]);
    {
        constexpr int ROW4P = KP / 4;
        const int64_t blk0 = chunk * (MF_WAVES * 16 * NTW);
        for (int e = tid; e < MF_WAVES * 16 * NTW * ROW4P; e += NTHR) {
            const int64_t m = blk0 + e / ROW4P;
            float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M) {
                const int64_t o = m * KP + 4 * (e % ROW4P);
                p4 = *reinterpret_cast<const float4*>(P + o);
            }
            reinterpret_cast<float4*>(s_p)[e] = p4;
        }
        __syncthreads();
    }
    auto p_at = [&](int64_t m, int k) -> float { return s_p[(int)(m - chunk * (MF_WAVES * 16 * NTW)) * KP + k]; };   // rows past M hold zeros
    // ---- resident A operands built from P ----
    uint4 pa_r1[NTW], pa_r2[NTW], pa_r3[W ? NTW : 1];     // R^T: lane (row = SNP n, slot a)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int64_t m = snp_of(t, n >> 2, n & 3);
        const int k0 = W ? 8 * (a & 1) : 0;               // W: slots alternate k 0..7 / 8..15
        uint32_t h[4], md[4], lo[4];
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const float v0 = (k0 + k < KP) ? p_at(m, k0 + k) : 0.f;
            const float v1 = (k0 + k + 1 < KP) ? p_at(m, k0 + k + 1) : 0.f;
            split3_pair(v0, v1, h[k >> 1], md[k >> 1], lo[k >> 1]);
        }
        const uint4 H = make_uint4(h[0], h[1], h[2], h[3]);
        const uint4 Md = make_uint4(md[0], md[1], md[2], md[3]);
        const uint4 Lo = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        // lane-group dependent choice of the piece, written as mask blends: a ?: on whole uint4 values is turned into a
        // table in scratch memory indexed by the lane group (64 B of scratch stores + loads per tile and thread)
        const uint32_t m01 = lt_mask(a, 2), m0 = lt_mask(a, 1), m1 = m01 & ~m0;          // a < 2, a == 0, a == 1 (nadm_common.h: no selects)
        if constexpr (W) {
            pa_r1[t] = H;                                                                        // slots [Ph Ph' Ph Ph']
            pa_r2[t] = Md;                                                                       // slots [Pm Pm' Pm Pm']
            pa_r3[t] = make_uint4((H.x & m01) | (Lo.x & ~m01), (H.y & m01) | (Lo.y & ~m01), (H.z & m01) | (Lo.z & ~m01),
                                  (H.w & m01) | (Lo.w & ~m01));                              // slots [Ph Ph' Pl Pl']
        } else if constexpr (ONE) {
            const uint32_t m2 = lt_mask(a, 3) & ~m01;                                            // a == 2
            pa_r1[t] = make_uint4((H.x & (m0 | m2)) | (Md.x & m1), (H.y & (m0 | m2)) | (Md.y & m1),
                                  (H.x & m0) | (Md.x & m1) | (Lo.x & m2), (H.y & m0) | (Md.y & m1) | (Lo.y & m2));   // slots [Ph|Ph Pm|Pm Ph|Pl 0]
            pa_r2[t] = make_uint4(0, 0, 0, 0);
        } else {
            pa_r1[t] = make_uint4((H.x & m01) | (Md.x & ~m01), (H.y & m01) | (Md.y & ~m01), (H.z & m01) | (Md.z & ~m01),
                                  (H.w & m01) | (Md.w & ~m01));                              // slots [Ph Ph Pm Pm]
            pa_r2[t] = make_uint4((H.x & m0) | (Lo.x & m1), (H.y & m0) | (Lo.y & m1), (H.z & m0) | (Lo.z & m1),
                                  (H.w & m0) | (Lo.w & m1));                                 // slots [Ph Pl 0 0]
        }
    }
    // dQ^T: lane (row n, slot a = 8 SNPs of the tile pair).  P enters dQ (and Q enters dP) with its hi and mid pieces only:
    // dR itself is carried as hi + lo = 16-17 bits, so a 24-bit partner would add MFMAs without adding accuracy.
    // K <= 8: rows 0-7 = k (hi), rows 8-15 = k (mid) in ONE operand; W: rows = the 16 k, one operand per piece
    uint4 pa_q1[NTW / 2], pa_q2[W ? NTW / 2 : 1];
#pragma unroll
    for (int tp = 0; tp < NTW / 2; ++tp) {
        uint32_t w1[4], w2[4];
        const int kq = W ? n : (n & 7);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const int64_t m0 = snp_of(2 * tp + (e >> 2), a, e & 3), m1 = snp_of(2 * tp + (e >> 2), a, (e & 3) + 1);
            const float v0 = (kq < KP) ? p_at(m0, kq) : 0.f;
            const float v1 = (kq < KP) ? p_at(m1, kq) : 0.f;
            uint32_t h, md, lo;
            split3_pair(v0, v1, h, md, lo);
            if constexpr (W) { w1[e >> 1] = h; w2[e >> 1] = md; }
            else { w1[e >> 1] = blend(h, md, lt_mask(n, 8)); w2[e >> 1] = 0u; }
        }
        pa_q1[tp] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        if constexpr (W) pa_q2[tp] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
    }
    f32x4 dpacc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) dpacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x2_t lossacc = {0.f, 0.f};

    // ---- X / Q staging (same scheme as the f32 MFMA kernel: unconditional clamped loads, index one tile ahead) ----
    constexpr int NPIECE = MF_TS * PPR;
    constexpr int QPT = (MF_TS * KW + NTHR - 1) / NTHR;    // Q elements staged per thread
    static_assert(NPIECE <= NTHR && (MF_TS * KW) % QPT == 0 && NTHR % KW == 0, "at most one X piece per thread");
    const bool has_piece = tid < NPIECE, has_q = tid < MF_TS * KW / QPT;
    const int pr = has_piece ? tid / PPR : 0, pc16 = tid % PPR;
    const int64_t poff = byte0 + pc16 * 16;
    const bool pcol_ok = poff * 4 < M;             // not `< ld`: on an SNP sub-range launch the bytes past M belong to the next range
    const uint32_t pcol_okm = lt_mask64(poff * 4, M);
    const int64_t poff_c = pcol_ok ? poff : 0;
    auto row_index = [&](int i0) -> int32_t { const int smp = i0 + pr; return idx[smp < b ? smp : b - 1]; };
    int32_t row_pref = row_index(0);
    uint4 stage;
    float qstage[QIMG ? 1 : QPT];
    static_assert(!QIMG || (IMG_U4 > 2 * NTHR && IMG_U4 <= 3 * NTHR), "three image pieces per thread, the third partly");
    uint4 qi0, qi1, qi2;                                    // (named scalars: an array indexed inside the two lambdas stays in scratch memory)
    const int qr0 = has_q ? tid / KW : 0, qk = tid % KW;   // this thread's Q elements: rows qr0 + j*NTHR/KW of the tile, column qk
    auto issue = [&](int i0) {
        if constexpr (QIMG) {
            const uint4* src = qimg + (int64_t)(i0 / MF_TS) * QI_TILE_U4;
            qi0 = src[tid];
            qi1 = src[tid + NTHR];
            qi2 = src[tid + 2 * NTHR < IMG_U4 ? tid + 2 * NTHR : IMG_U4 - 1];
            __builtin_amdgcn_sched_barrier(0);              // or the scheduler sinks the three loads to their use at the END of the tile
                                                            // (shorter live ranges) and every tile waits for an L2 round trip: +13 %
        } else {
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const int qr = qr0 + j * (MF_TS / QPT);
                const int smp = i0 + qr < b ? i0 + qr : b - 1;
                qstage[j] = Q[(int64_t)smp * SP + (qk < KP ? qk : 0)];
            }
        }
        // (unsigned 32-bit factors: one v_mad_u64_u32; the signed 64 x 64 product -- two v_mul_lo_u32 and a sign extension more per tile and thread, in the chain
        // in front of the tile's gather load -- cost the launch 1.2 %, more than their 4.6 issue cycles each: profiles/r06_abl_p2addr.txt; the launcher refuses rows of 4 GiB and more)
        stage = *reinterpret_cast<const uint4*>(xp + ((uint64_t)(uint32_t)row_pref * (uint32_t)ld + (uint64_t)poff_c));
        row_pref = row_index(i0 + MF_TS);
    };
    auto commit = [&](int i0) {
        const bool ok = pcol_ok && (i0 + pr < b);
        const uint32_t okm = pcol_okm & lt_mask(i0 + pr, b);           // (a select here is four 23-cycle v_cndmask per tile, nadm_common.h)
        // missing calls -> 0 HERE, once per loaded word (r03; until then every lane cleaned the word it took out of the tile): the tile
        // and the copy for pass 3 both hold the model's input
        const uint4 cl = make_uint4(clean_codes(stage.x & okm), clean_codes(stage.y & okm), clean_codes(stage.z & okm), clean_codes(stage.w & okm));
        if (has_piece) *reinterpret_cast<uint4*>(&s_x[pr * RS + pc16 * 16]) = cl;
        // by-product for pass 3: the batch as a copy of its own, missing calls already 0, TILED the way pass 3 walks it -- tile t = the
        // 128 byte columns of pass 3's chunk t, [b rows][128 bytes] back to back (xg_piece).  A block of pass 3 then streams ONE
        // contiguous b x 128 byte region instead of gathering 128-byte pieces 125 KB apart, which arrive at 2.6 TB/s (and, out of a
        // 12.5 GB resident matrix, miss the per-CU translation cache on 13 % of the requests); here the store is one instruction
        // per tile and thread.
        if (xg != nullptr && has_piece && ok) *reinterpret_cast<uint4*>(xg_piece(xg, b, i0 + pr, poff)) = cl;
        if constexpr (QIMG) {
            s_qimg[tid] = qi0;
            s_qimg[tid + NTHR] = qi1;
            if (tid + 2 * NTHR < IMG_U4) s_qimg[tid + 2 * NTHR] = qi2;
            return;
        }
        if (!has_q) return;
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            // Q element -> bf16 pieces scattered into the MFMA operand images
            const int qr = qr0 + j * (MF_TS / QPT);
            const float v = (i0 + qr < b && qk < KP) ? qstage[j] : 0.f;
            uint32_t h, md, lo;
            split3(v, h, md, lo);
            const int st = qr >> 4, i = qr & 15;
            const int pair = qr >> 5, within = qr & 31, q8 = within >> 3, e = within & 7;
            if constexpr (W) {
                const int sl = qk >> 3, kk = qk & 7;                                         // k slot and position inside it
                uint16_t* r1 = reinterpret_cast<uint16_t*>(&s_qr[st][0][0]) + kk;          // + lane*8 (uint16 units)
                uint16_t* r2 = reinterpret_cast<uint16_t*>(&s_qr[st][1][0]) + kk;
                r1[(i + 16 * sl) * 8] = (uint16_t)h;   r1[(i + 16 * (2 + sl)) * 8] = (uint16_t)md;   // [Qh Qh' Qm Qm']
                r2[(i + 16 * sl) * 8] = (uint16_t)lo;  r2[(i + 16 * (2 + sl)) * 8] = (uint16_t)h;    // [Ql Ql' Qh Qh']
                uint16_t* d1 = reinterpret_cast<uint16_t*>(&s_qd[pair][0][0]) + e;
                uint16_t* d2 = reinterpret_cast<uint16_t*>(&s_qd[pair][1][0]) + e;
                d1[(q8 * 16 + qk) * 8] = (uint16_t)h;                                        // column qk of Qh / Qm
                d2[(q8 * 16 + qk) * 8] = (uint16_t)md;
            } else {
                uint16_t* r1 = reinterpret_cast<uint16_t*>(&s_qr[st][0][0]) + qk;       // + lane*8 (uint16 units)
                uint16_t* r2 = reinterpret_cast<uint16_t*>(&s_qr[st][1][0]) + qk;
                if constexpr (ONE) {
                    if (qk < 4) {
                        r1[(i) * 8] = (uint16_t)h;        r1[(i) * 8 + 4] = (uint16_t)md;        // slot 0: Qh | Qm
                        r1[(i + 16) * 8] = (uint16_t)h;   r1[(i + 16) * 8 + 4] = (uint16_t)md;   // slot 1: Qh | Qm
                        r1[(i + 32) * 8] = (uint16_t)lo;  r1[(i + 32) * 8 + 4] = (uint16_t)h;    // slot 2: Ql | Qh
                    }
                } else {
                r1[(i) * 8] = (uint16_t)h;        r1[(i + 32) * 8] = (uint16_t)h;            // slots 0,2: Qh
                r1[(i + 16) * 8] = (uint16_t)md;  r1[(i + 48) * 8] = (uint16_t)md;           // slots 1,3: Qm
                r2[(i) * 8] = (uint16_t)lo;       r2[(i + 16) * 8] = (uint16_t)h;            // slots 0,1: Ql, Qh
                }
                uint16_t* d1 = reinterpret_cast<uint16_t*>(&s_qd[pair][0][0]) + e;
                d1[(q8 * 16 + qk) * 8] = (uint16_t)h;                                        // columns 0..7: Qh
                d1[(q8 * 16 + qk + 8) * 8] = (uint16_t)md;                                   // columns 8..15: Qm
            }
        }
    };

    const int n_slices = SLICED ? (int)gridDim.y : 1, slice = SLICED ? (int)blockIdx.y : 0;
    const int ntiles = (b + MF_TS - 1) / MF_TS;
    const int tps = (ntiles + n_slices - 1) / n_slices;      // (the host picks S so that no slice is empty)
    const int tl0 = SLICED ? slice * tps : 0, tl1 = SLICED ? min(ntiles, tl0 + tps) : ntiles;
    if constexpr (SLICED) row_pref = row_index(tl0 * MF_TS);
    __syncthreads();                                        // zero fill visible before the first commit
    issue(tl0 * MF_TS);
    commit(tl0 * MF_TS);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    float kmax_v = 1e-12f;                                 // 1 / 1e-12 as v_rcp_f32 computes it (opaque to the constant folder)
    asm volatile("v_rcp_f32 %0, %0" : "+v"(kmax_v));
    const float scale_back = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kmax_v)));   // bce_grad2: dR is carried times 1e-12
    float eps_v = 1e-12f;
    asm volatile("" : "+s"(eps_v));                        // (opaque: the compiler would otherwise pair the two multiplies and clamp separately)
    const float eps = eps_v;
    constexpr bool FAST_LOSS = LOSS && UNIT_P;              // one logarithm per pair of genotypes, exact form as the cold fallback
    // the lane's codes of one 16-sample tile: nibble = one 2-bit code, codes 0,2 / 1,3 of each byte (byte t = 4 SNPs of tile t)
    auto load_codes = [&](int st, uint32_t& ev, uint32_t& od) {
        uint32_t w;
        if constexpr (NTW == 4) w = *reinterpret_cast<const uint32_t*>(&s_x[(16 * st + n) * RS + wave * 16 + 4 * a]);
        else w = *reinterpret_cast<const uint16_t*>(&s_x[(16 * st + n) * RS + wave * 8 + 2 * a]);
        ev = w & 0x33333333u;                              // (missing calls are 0 already: commit)
        od = (w >> 2) & 0x33333333u;
    };
    // R^T tile t (16 SNPs x 16 samples) from the resident P operands and one sample tile's Q operands
    auto recon = [&](int t, const uint4& q1, const uint4& q2) -> f32x4 {
        f32x4 D = (f32x4){0.f, 0.f, 0.f, 0.f};
        D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa_r1[t]), as_bf16x8(q1), D, 0, 0, 0);
        if constexpr (W) {
            D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa_r2[t]), as_bf16x8(q1), D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa_r3[t]), as_bf16x8(q2), D, 0, 0, 0);
        } else if constexpr (!ONE) {
            D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa_r2[t]), as_bf16x8(q2), D, 0, 0, 0);
        }
        return D;
    };
    uint16_t* const tw = &s_t[wave][0][0][0];
    const int wchunk = (a + (n >> 2)) & 3;                 // where this lane's 8-byte chunk of row 16*s2 + n goes (rotated, see s_t)
    for (int tl = tl0; tl < tl1; ++tl) {
        const int i0 = tl * MF_TS;
        const int nt = min(MF_TS, b - i0);
        if (tl + 1 < tl1) issue(i0 + MF_TS);

#pragma unroll 1
        for (int p = 0; p < MF_TS / 32; ++p) {
            if (i0 + 32 * p < b) {                                     // block-uniform
                uint32_t even[2], odd[2];
                uint4 qb1[2], qb2[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int st = 2 * p + s2;
                    load_codes(st, even[s2], odd[s2]);
                    qb1[s2] = s_qr[st][0][lane];
                    qb2[s2] = s_qr[st][1][lane];
                }
                float it_acc = 0.f;                                    // fast loss of this tile pair: sum of log2(f0 * f1) over the lane's pairs
                const uint4 qd1 = s_qd[p][0][lane];
                uint4 qd2 = make_uint4(0, 0, 0, 0);
                if constexpr (W) qd2 = s_qd[p][1][lane];
                f32x4 dq[2];
                dq[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dq[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tp = 0; tp < NTW / 2; ++tp) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        uint32_t hi[2][2], lo[2][2];                   // [t2][pair of r] of this sample tile
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            const int t = 2 * tp + t2;
                            const f32x4 D = recon(t, qb1[s2], qb2[s2]);
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const uint32_t cw = h2 ? odd[s2] : even[s2];
                                const f32x2_t d = {D[2 * h2], D[2 * h2 + 1]}, x = fp4_pair(cw, t);
                                const f32x2_t dR = bce_grad2(d, x, eps);
                                if constexpr (FAST_LOSS)
                                    bce_loss_prod2(d, x, it_acc);
                                else if constexpr (LOSS)
                                    bce_loss_exact2<UNIT_P>(d, (f32x2_t){1.f, 1.f} - d, x, lossacc);
                                const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(dR, bf16x2_t));
                                hi[t2][h2] = hp;
                                const f32x2_t rem = dR - (f32x2_t){__uint_as_float(hp << 16), __uint_as_float(hp & 0xFFFF0000u)};
                                lo[t2][h2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, bf16x2_t));
                            }
                            // transposition buffer of this SNP tile: T[t2][hl][sample 16*s2 + n][SNP 4a .. 4a+3]  (row = 16 bf16 = 32 B)
                            *reinterpret_cast<uint2*>(tw + (2 * t2 + 0) * TWP + (16 * s2 + n) * TWS + 4 * wchunk) = make_uint2(hi[t2][0], hi[t2][1]);
                            *reinterpret_cast<uint2*>(tw + (2 * t2 + 1) * TWP + (16 * s2 + n) * TWS + 4 * wchunk) = make_uint2(lo[t2][0], lo[t2][1]);
                        }
                        // dQ^T of this sample tile: the lane's 8 dR values (2 tiles x 4 SNPs) are the B operand
                        const bf16x8 bh = as_bf16x8(make_uint4(hi[0][0], hi[0][1], hi[1][0], hi[1][1]));
                        const bf16x8 bl = as_bf16x8(make_uint4(lo[0][0], lo[0][1], lo[1][0], lo[1][1]));
                        dq[s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa_q1[tp]), bh, dq[s2], 0, 0, 0);
                        dq[s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa_q1[tp]), bl, dq[s2], 0, 0, 0);
                        if constexpr (W) {
                            dq[s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa_q2[tp]), bh, dq[s2], 0, 0, 0);
                        }
                    }
                    // dP: per SNP tile, read dR (hi, lo) of the 32 samples back transposed, then 2 (W: 4) MFMAs
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        // A operand: lane (row = SNP n, slot a = samples 8a..8a+7): two transposing reads of 4 samples each
                        typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_p;
                        const int trow = 8 * a + (n >> 2);
                        const int tcol = 4 * (((n & 3) + 2 * a) & 3);                               // rows trow: (trow >> 2) & 3 = 2a
                        const int tcol4 = 4 * (((n & 3) + 2 * a + 1) & 3);                          // rows trow + 4: 2a + 1
                        const uint16_t* th = tw + (2 * t2 + 0) * TWP, *tlw = tw + (2 * t2 + 1) * TWP;
                        const s16x4_t h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(th + trow * TWS + tcol));
                        const s16x4_t h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(th + (trow + 4) * TWS + tcol4));
                        const s16x4_t l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(tlw + trow * TWS + tcol));
                        const s16x4_t l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(tlw + (trow + 4) * TWS + tcol4));
                        const bf16x8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                        const bf16x8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                        const int t = 2 * tp + t2;
                        dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf16x8(qd1), dpacc[t], 0, 0, 0);
                        dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, as_bf16x8(qd1), dpacc[t], 0, 0, 0);
                        if constexpr (W) {
                            dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf16x8(qd2), dpacc[t], 0, 0, 0);
                        }
                    }
                }
                if constexpr (FAST_LOSS) {
                    // it_acc = sum over the lane's 4 * NTW pairs of 2 * (their terms in log2 units).  -inf (a zero or underflowed
                    // product: a clamp of the reference may be active) or NaN anywhere in the wave: recompute the tile pair's
                    // loss in the exact form -- R^T again from the operands still in LDS / registers, loss algebra only
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(it_acc) < __builtin_inff())) != 0, 0)) {
                        f32x2_t ex = {0.f, 0.f};
#pragma unroll 1
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const int st = 2 * p + s2;
                            uint32_t ev, od;
                            load_codes(st, ev, od);
                            const uint4 q1 = s_qr[st][0][lane], q2 = s_qr[st][1][lane];
#pragma unroll
                            for (int t = 0; t < NTW; ++t) {
                                const f32x4 D = recon(t, q1, q2);
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    const f32x2_t d = {D[2 * h2], D[2 * h2 + 1]};
                                    bce_loss_exact2<true>(d, (f32x2_t){1.f, 1.f} - d, fp4_pair(h2 ? od : ev, t), ex);
                                }
                            }
                        }
                        it_acc = 2.f * ((ex.x + ex.y) - LOSS_LOG_SHIFT * NTW * 8);
                    }
                    lossacc.x += 0.5f * it_acc;
                } else if constexpr (LOSS) {   // the "- 20" of the shifted logs: 2 * NTW * 4 genotypes per lane and tile pair, half of them per packed half
                    lossacc -= (f32x2_t){LOSS_LOG_SHIFT * NTW * 4, LOSS_LOG_SHIFT * NTW * 4};
                }
                // K <= 8: dQ^T rows k (from the hi piece of P, lanes a < 2) and k + 8 (from the mid piece, lanes a >= 2) are two
                // partial sums of the same dQ element: both are parked in LDS and meet in the cross-wave sum below (a register fold
                // with v_permlane32_swap cost 16 VALU instructions per tile pair).  W: rows 4a + r ARE k.
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int ah = W ? a : (a & 1), part = W ? 0 : (a >> 1);
                    if (4 * ah < KP)
                        *reinterpret_cast<float4*>(&s_dq[wave][part][(16 * (2 * p + s2) + n) * KP + 4 * ah]) =
                            make_float4(dq[s2][0], dq[s2][1], dq[s2][2], dq[s2][3]);
                }
            }
        }
        __syncthreads();
        for (int e4 = tid; e4 < nt * KP / 4; e4 += NTHR) {          // 16 B per lane: the tile's slab rows are contiguous
            float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < MF_WAVES; ++w)
#pragma unroll
                for (int part = 0; part < (W ? 1 : 2); ++part) {
                    const float4 v = *reinterpret_cast<const float4*>(&s_dq[w][part][4 * e4]);
                    sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
                }
            *reinterpret_cast<float4*>(dqpart + (chunk * b + i0) * KP + 4 * e4) =
                make_float4(sm.x * scale_back, sm.y * scale_back, sm.z * scale_back, sm.w * scale_back);
        }
        if (tl + 1 < tl1) commit(i0 + MF_TS);
        __syncthreads();
    }

    // ---- dP: columns 0..7 (hi part) and 8..15 (mid part) fold with a rotate by 8 inside the 16-lane row ----
    // The block's dP rows are one contiguous [chunk SNPs x KP] slab: staged through LDS (the transposition buffers are free
    // now) and written as full 16 B / lane rows instead of 32 B pieces scattered over rows.
    static_assert(sizeof(s_t) >= (size_t)MF_WAVES * 16 * NTW * KP * sizeof(float), "dP staging fits the transposition buffers");
    float* const s_dp = reinterpret_cast<float*>(&s_t[0][0][0][0]);
    const int64_t snp_blk0 = chunk * (MF_WAVES * 16 * NTW);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dpacc[t][r];
            if constexpr (!W) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /*row_ror:8*/, 0xf, 0xf, false));
            if (n < KP) s_dp[(int)(snp_of(t, a, r) - snp_blk0) * KP + n] = v * scale_back;
        }
    }
    constexpr int ROW4 = KP / 4;                                      // float4 per SNP row
    constexpr int CH_F4 = MF_WAVES * 16 * NTW * ROW4;                 // float4 of a chunk's [SNPs x KP] slab
    auto finish = [&](int e, const float4 g4) {                       // row piece e of the chunk, gradient complete
        const int64_t m = snp_blk0 + e / ROW4;
        if (m < M) {
            const int64_t o = m * KP + 4 * (e % ROW4);
            // single-GPU step: the gradient of these rows is final here and nothing else in the step reads P again, so
            // Adam + clamp is applied on the spot (no dP round trip through HBM, no separate launch for the P matrices)
            if (ad.m != nullptr) adam_float4(P + o, g4, ad.m + o, ad.v + o, ad.step_size, ad.inv_bc2, ad.grad_scale, true);
            else *reinterpret_cast<float4*>(dP + o) = g4;
        }
    };
    if constexpr (!SLICED) {
        __syncthreads();
        for (int e = tid; e < CH_F4; e += NTHR) finish(e, *reinterpret_cast<const float4*>(s_dp + 4 * e));
        if constexpr (LOSS) {
            const float sl = wave_sum_lane63(-0.69314718055994530942f * (lossacc.x + lossacc.y));
            if (lane == 63) s_loss[wave] = sl;
            __syncthreads();
            if (tid == 0) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < MF_WAVES; ++w) tot += s_loss[w];
                losspart[chunk] = tot;
            }
        }
        if (probe && tid == 0) {
            clk[0] = __builtin_readcyclecounter() - pc0;
            clk[1] = __builtin_amdgcn_s_memrealtime() - pr0;
        }
        return;
    }
    float my_loss = 0.f;
    if constexpr (LOSS) {
        const float sl = wave_sum_lane63(-0.69314718055994530942f * (lossacc.x + lossacc.y));
        if (lane == 63) s_loss[wave] = sl;
    }
    __syncthreads();
    if constexpr (LOSS) {
        if (tid == 0) {
#pragma unroll
            for (int w = 0; w < MF_WAVES; ++w) my_loss += s_loss[w];
        }
    }
    // ---- S > 1: park the partial, be counted, the last one adds them up (see the head of the kernel)
    constexpr int SLAB_F = p2_slab_floats(KP);                        // floats per (slice, chunk): the slab + 4 (the loss partial + pad)
    float* const mine = slab + ((int64_t)slice * gridDim.x + chunk) * SLAB_F;
    for (int e = tid; e < CH_F4; e += NTHR) {
        const float4 g4 = *reinterpret_cast<const float4*>(s_dp + 4 * e);
        __hip_atomic_store(mine + 4 * e + 0, g4.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // written THROUGH to memory: the block
        __hip_atomic_store(mine + 4 * e + 1, g4.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // that adds the partials may sit on
        __hip_atomic_store(mine + 4 * e + 2, g4.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // another XCD, behind another L2
        __hip_atomic_store(mine + 4 * e + 3, g4.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store(mine + 4 * CH_F4, my_loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this wave's stores have been acknowledged ...
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(&slice_cnt[chunk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the block is counted
        s_last = old == n_slices - 1;
        if (s_last) __hip_atomic_store(&slice_cnt[chunk], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;                                              // block-uniform
    for (int e = tid; e < CH_F4; e += NTHR) {
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sl = 0; sl < n_slices; ++sl) {                       // slice order, whoever is last
            const float* part = slab + ((int64_t)sl * gridDim.x + chunk) * SLAB_F + 4 * e;
            g4.x += __hip_atomic_load(part + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g4.y += __hip_atomic_load(part + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g4.z += __hip_atomic_load(part + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g4.w += __hip_atomic_load(part + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        finish(e, g4);
    }
    if constexpr (LOSS) {
        if (tid == 0) {
            float tot = 0.f;
            for (int sl = 0; sl < n_slices; ++sl)
                tot += __hip_atomic_load(slab + ((int64_t)sl * gridDim.x + chunk) * SLAB_F + 4 * CH_F4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            losspart[chunk] = tot;
        }
    }
}

// side work of pass 3's launch: the MLP weight-gradient partials as extra blocks (mlp_bwd_b_block)
struct MlpSide {               // small_part == nullptr: no side work
    nadm_heads_t hd;
    int b, gx;
    const float *Zn, *H, *dL, *dHpre, *dgp;
    float* small_part;
};
constexpr int EB_G = 2;                   // groups of 16 byte columns (64 SNPs) per wave
constexpr int EB_COLS = 4 * EB_G * 16;    // packed byte columns per block (128)
constexpr int EB_CHUNK_SNPS = EB_COLS * 4;

// =================================================================================================
// pass 3 on the FP4 x FP6 matrix instruction (CP <= 8, r03): dV = X^T . dZ on v_mfma_scale_f32_16x16x128_f8f6f4.
//   A nibble 00cc IS the FP4 (E2M1) number c/2, so the genotypes enter the matrix pipe WITHOUT a conversion: the B operand of
//   one instruction is 32 samples x 1 SNP per lane (4 registers of nibbles), K = 128 samples per instruction.  What is left on
//   the VALU is the bit transposition (the packed byte holds 4 SNPs of one sample, the operand wants 8 samples of one SNP per
//   register): 40 instructions per 16 byte columns x 128 samples, where the bf16 kernel of rounds 1-2 (v_mfma_f32_16x16x32_bf16, dZ
//   split hi + mid + lo, X converted with v_cvt_scalef32_pk_bf16_fp4) spent 4 x 40 on shifts and masks and 64 conversions.  And
//   K = 128 costs 22-25 issue cycles where four K = 32 bf16 instructions cost 71 (profiles/r03_ubench_fp4_mfma.txt).
//   dZ is the A operand, as FP6 (E2M3) pieces: the block of 32 samples x one column a lane holds is cut into EIGHT pieces of
//   four bits -- the hexadecimal digits of |dZ| in fixed point below 16 x the block's largest magnitude, each with the sign
//   of the value; digit h is the FP6 number h/8 exactly -- and the instruction's per-lane E8M0 scale carries the digit's
//   weight 2^(E0 - 4p + 3).  Rows of the instruction = (piece parity, column), four instructions (row groups) per X operand
//   accumulate all eight pieces into ONE accumulator; rows c and c + 8 are folded in the epilogue.  32 bits below the
//   block maximum: an element within 2^-8 of it is carried exactly, a smaller one to an absolute error < 2^-31 of the block
//   maximum (the bf16 kernel carried 24 bits below every element's own magnitude; against a float64 product this pass is 3-4 x
//   closer than an fp32 matmul, profiles/r03_p3_accuracy.json).  The image is the same for every block of the launch: it is built
//   ONCE per step -- by the MLP backward (nadm_mlp_bwd_image), or by dz_image_kernel below -- instead of once per block and tile
//   (7 KB per 128 samples; layout and arithmetic: nadm_common.h, fp6_piece / dzi_build_piece).
//   The rows come as a stream when pass 2 has left its tiled copy of the batch (CLEAN_SRC, xg_piece): gathered out of a row-major
//   matrix the 128-byte pieces arrive at 2.6 TB/s and the kernel takes 50 us instead of 40.
//   Element order inside a lane (dzi_sample): what the bit transposition produces, see the tile loop.
// =================================================================================================
typedef int i32x8_t __attribute__((ext_vector_type(8)));

// the image on its own (callers whose dZ does not come out of nadm_mlp_bwd_image): grid = tiles of 128 samples, 256 threads:
// thread (piece p = tid & 7, column c = (tid >> 3) & 7, K-block q = tid >> 6)
__global__ __launch_bounds__(256) void dz_image_kernel(const float* __restrict__ dZ, int b, int CP, uint4* __restrict__ img) {
    const int tid = threadIdx.x;
    dzi_build_piece<false>(dZ, b, CP, img, blockIdx.x, tid >> 6, (tid >> 3) & 7, tid & 7);
}

constexpr int EB4_XS = 132;               // bytes per byte column of the LDS tile: 128 samples + 4 (dword stride 33: the loader's 4-byte
                                          // stores of 32 consecutive column quads and the 4-byte column reads spread over the banks)
// CLEAN_SRC: xp is pass 2's copy of the batch (xg_piece: tiled by this kernel's chunks, rows in batch order, missing calls already 0)
// floats one (sample slice, SNP chunk) block of pass 3 parks for the block that adds the slices up: the chunk's [512 SNPs x CP] partial of dV
constexpr int p3_slab_floats(int cp) { return EB_CHUNK_SNPS * cp; }

// SLICED (r06): S sample slices.  The grid is one block per 512-SNP chunk with the batch's 128-sample tiles as a loop inside:
// a rank of the SNP-sharded mode (6400 rows x 62.5k SNPs at configs[3] on 8 GPUs) launches 122 blocks on 256 CUs and runs 75 us where the
// single-GPU shape (800 x 500k, the same genotypes) runs 41.  The tiles are dealt to S blocks per chunk; every slice parks its partial
// [512 x CP] sum in `slab`, is counted, and the block counted LAST adds the S partials in slice order and runs the epilogue (Adam on the
// chunk's V rows, or the gradient store) -- pass 2's idiom (decode_bce_bf16_kernel), the dZ image's hand-off contract: nobody
// waits, the sum's order is fixed, S is a function of (b, M) alone (nadm_encode_slices).  false compiles the kernel as it stood.
template <int CP, bool CLEAN_SRC, bool SLICED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void encode_bwd_fp4_kernel(const uint8_t* __restrict__ xp, int64_t ld,
                                                              const int32_t* __restrict__ idx, int b, int64_t M,
                                                              const uint4* __restrict__ dzimg, float* __restrict__ dV,
                                                              uint32_t missing_bf16, float* __restrict__ Vrw, AdamFused ad, MlpSide side,
                                                              float* slab, int* slice_cnt, int n_slices_arg) {
    static_assert(CP <= 8, "rows of the instruction: piece parity x 8 columns");
    const int64_t nchunks = (M + EB_CHUNK_SNPS - 1) / EB_CHUNK_SNPS;
    const int n_slices = SLICED ? n_slices_arg : 1;
    // grid.x = chunks x slices (slice-major: all chunks of slice 0, then slice 1, ...) + the side blocks LAST: as a second grid dimension
    // the slices were dispatched behind slice 0's ~1600 side blocks (b = 6400) and every further slice made the launch slower
    {   // blocks past the SNP chunks: the MLP weight-gradient partials (independent of pass 3; see mlp_bwd_b_block)
        if ((int64_t)blockIdx.x >= nchunks * n_slices) {
            const int e = (int)(blockIdx.x - nchunks * n_slices);
            mlp_bwd_b_block(side.hd, side.b, side.Zn, side.H, side.dL, side.dHpre, side.dgp, side.small_part, e % side.gx, e / side.gx);
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) uint8_t s_xt[2][EB_COLS * EB4_XS];      // [byte column][128 samples]; the block's dV rows at the end
    static_assert(sizeof(s_xt) >= (size_t)EB_CHUNK_SNPS * CP * sizeof(float), "the dV image fits the tile buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mcol = lane & 15, q = lane >> 4;
    const int64_t chunk = SLICED ? (int64_t)(blockIdx.x % nchunks) : (int64_t)blockIdx.x;
    const int64_t byte0 = chunk * EB_COLS;
    // a missing call (code 3) is 0 in the model and 1.5 in the init-time products: 1.5 is the FP4 value of the nibble 0011, for the
    // model the loader clears both bits of every code 3 before the tile goes to LDS
    const uint32_t kmiss = missing_bf16 == 0u ? 0x55555555u : 0u;
    // ---- loader: thread -> 4 rows (row quad rq of the 128-sample tile) x 16 byte columns: four 16-byte loads per tile.  (With the bf16
    // kernel's 4-byte loads -- 16 per thread and tile -- the X loads were not hidden behind the tile's arithmetic at all: 43.5 us
    // against 25.5 us without them, profiles/r03_p3_fp4.txt.)  The 16 x 4 bytes go through four 4x4 byte transposes into the
    // column-major LDS tile.  Column c sits in slot (c >> 4) * 8 + (c & 7) + 8 * (c & 8): with the dword stride 33 the 4-byte stores
    // of a wave (8 column groups x 8 row quads) then fall on every bank twice -- in natural order 16 columns apart means 16 banks
    // apart and they would pile up four deep.
    const int cgrp = tid & 7, rq = tid >> 3;
    const int64_t loff = byte0 + 16 * cgrp;
    const bool lcol_ok = loff * 4 < M;             // not `< ld`: sub-range launches, see pass 2
    const int64_t loff_c = lcol_ok ? loff : 0;
    uint32_t lmask[4];                                        // 16 bytes = 64 SNPs, M need not be a multiple: per dword
#pragma unroll
    for (int t = 0; t < 4; ++t) lmask[t] = lt_mask64((loff + 4 * t) * 4, M);
    auto row_idx = [&](int i0, int k) -> int32_t {
        const int smp = i0 + 4 * rq + k, smc = smp < b ? smp : b - 1;
        if constexpr (CLEAN_SRC) return smc;                  // the copy holds the batch in order
        else return idx[smc];
    };
    int32_t rows[4];
    uint4 xw[4];
    const uint32_t ld32 = (uint32_t)ld;                       // (a row is < 4 GB)
    static_assert(XG_TILE_COLS == EB_COLS, "the batch copy is tiled by this kernel's chunks");
    const uint8_t* const xtile = xp + chunk * ((int64_t)b * XG_TILE_COLS) + 16 * cgrp;
    const bool col_edge = (byte0 + EB_COLS) * 4 > M;          // block-uniform: only the last chunk has byte columns past M
    auto fetch_rows = [&](int i0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rows[k] = row_idx(i0, k);
    };
    auto issue = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (CLEAN_SRC) xw[k] = *reinterpret_cast<const uint4*>(xtile + rows[k] * XG_TILE_COLS);       // one contiguous b x 128 B region per block
            else xw[k] = *reinterpret_cast<const uint4*>(xp + ((uint64_t)(uint32_t)rows[k] * ld32 + (uint64_t)loff_c));   // one v_mad_u64_u32
        }
    };
    auto col_slot = [](int c) -> int { return ((c >> 4) << 3) | (c & 7) | ((c & 8) << 3); };
    // dword t of the four rows -> byte columns 16 cgrp + 4t .. + 3, samples 4 rq .. 4 rq + 3 of tile buffer `buf`.  EDGE: the tile
    // reaches past the batch or the chunk past M (tile- / block-uniform): only then are rows / columns masked
    auto commit = [&](int buf, int i0, int t, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        uint32_t d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t raw = t == 0 ? xw[k].x : (t == 1 ? xw[k].y : (t == 2 ? xw[k].z : xw[k].w));
            uint32_t r = raw;
            if constexpr (EDGE) r = raw & lmask[t] & lt_mask(i0 + 4 * rq + k, b);  // (no selects: nadm_common.h)
            if constexpr (CLEAN_SRC) d[k] = r;
            else { const uint32_t m3 = r & (r >> 1) & kmiss; d[k] = r ^ (m3 | (m3 << 1)); }
        }
        // 4x4 byte transpose: e[c] = byte c of rows 0..3
        const uint32_t t01l = __builtin_amdgcn_perm(d[1], d[0], 0x05010400u);   // d0.b0 d1.b0 d0.b1 d1.b1
        const uint32_t t01h = __builtin_amdgcn_perm(d[1], d[0], 0x07030602u);   // d0.b2 d1.b2 d0.b3 d1.b3
        const uint32_t t23l = __builtin_amdgcn_perm(d[3], d[2], 0x05010400u);
        const uint32_t t23h = __builtin_amdgcn_perm(d[3], d[2], 0x07030602u);
        const uint32_t e0 = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);     // t01l.b0 t01l.b1 t23l.b0 t23l.b1
        const uint32_t e1 = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
        const uint32_t e2 = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
        const uint32_t e3 = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
        uint8_t* base = &s_xt[buf][col_slot(16 * cgrp + 4 * t) * EB4_XS + 4 * rq];     // (slots of columns 4t .. 4t + 3 are consecutive)
        *reinterpret_cast<uint32_t*>(base) = e0;
        *reinterpret_cast<uint32_t*>(base + EB4_XS) = e1;
        *reinterpret_cast<uint32_t*>(base + 2 * EB4_XS) = e2;
        *reinterpret_cast<uint32_t*>(base + 3 * EB4_XS) = e3;
    };

    f32x4 acc[EB_G][4];
#pragma unroll
    for (int g = 0; g < EB_G; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[g][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: tile 0 committed, tile 1 in flight, the row indices of tile 2 fetched.  A tile's loads are issued one tile phase
    // (~1.5 k cycles of work per wave) before they are committed, its row indices one phase before that.
    const int ntiles_all = (b + DZI_TS - 1) / DZI_TS;
    const int slice = SLICED ? (int)(blockIdx.x / nchunks) : 0;
    const int tps = (ntiles_all + n_slices - 1) / n_slices;       // (the host picks S so that no slice is empty)
    const int tl0 = SLICED ? slice * tps : 0, ntiles = SLICED ? min(ntiles_all, tl0 + tps) : ntiles_all;      // this block's tiles: [tl0, ntiles)
    fetch_rows(tl0 * DZI_TS);
    issue();
    fetch_rows((tl0 + 1) * DZI_TS);
    auto commit_tile = [&](int buf, int i0) {
        if (col_edge || i0 + DZI_TS > b) {
#pragma unroll
            for (int t = 0; t < 4; ++t) commit(buf, i0, t, std::true_type{});
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) commit(buf, i0, t, std::false_type{});
        }
    };
    commit_tile(0, tl0 * DZI_TS);
    issue();
    fetch_rows((tl0 + 2) * DZI_TS);
    __syncthreads();

    for (int T = tl0; T < ntiles; ++T) {
        const int cur = (T - tl0) & 1;
        // the tile's A operands: 7 x 16 B per lane of the image every block of the launch reads (L2)
        const uint4* zi = dzimg + (int64_t)T * DZI_TILE_U4 + lane;
        uint4 z[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) z[k] = zi[64 * k];
        const uint32_t zd[24] = {z[0].x, z[0].y, z[0].z, z[0].w, z[1].x, z[1].y, z[1].z, z[1].w, z[2].x, z[2].y, z[2].z, z[2].w,
                                 z[3].x, z[3].y, z[3].z, z[3].w, z[4].x, z[4].y, z[4].z, z[4].w, z[5].x, z[5].y, z[5].z, z[5].w};
        const int zscale = (int)z[6].x;
#pragma unroll
        for (int g = 0; g < EB_G; ++g) {
            // the NEXT tile goes to the other buffer first thing, and the loads of the tile after that take its registers: they have
            // the whole phase to arrive.  (TWO tiles in flight -- 128 registers, 88 bytes of scratch -- ran 67 us against 50.)
            if (g == 0) {
                if (T + 1 < ntiles) commit_tile(cur ^ 1, (T + 1) * DZI_TS);
                issue();                                                    // (clamped indices: issuing past the batch is harmless)
                fetch_rows((T + 3) * DZI_TS);
            }
            // the lane's byte column, samples 32q .. 32q + 31: 8 words.  Bit transposition: words (2i, 2i + 1) hold samples 8i..8i+3 and
            // 8i+4..8i+7, four 2-bit fields (SNPs) per byte.  u takes fields 0, 1 of both words into the low / high nibble of every
            // byte, v fields 2, 3; masking with 0x33 then leaves one field per nibble: register i of field j's operand = nibbles
            // (sample 8i + t, sample 8i + 4 + t), t = 0..3 -- the order dzi_sample() names
            const uint32_t* colp = reinterpret_cast<const uint32_t*>(&s_xt[cur][col_slot(wave * (16 * EB_G) + g * 16 + mcol) * EB4_XS + 32 * q]);
            i32x8_t bx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bx[j] = (i32x8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t w0 = colp[2 * i], w1 = colp[2 * i + 1];
                const uint32_t uu = (w0 & 0x0F0F0F0Fu) | ((w1 << 4) & 0xF0F0F0F0u);
                const uint32_t vv = ((w0 >> 4) & 0x0F0F0F0Fu) | (w1 & 0xF0F0F0F0u);
                bx[0][i] = (int)(uu & 0x33333333u);
                bx[1][i] = (int)((uu >> 2) & 0x33333333u);
                bx[2][i] = (int)(vv & 0x33333333u);
                bx[3][i] = (int)((vv >> 2) & 0x33333333u);
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const i32x8_t az = {(int)zd[6 * rg], (int)zd[6 * rg + 1], (int)zd[6 * rg + 2], (int)zd[6 * rg + 3], (int)zd[6 * rg + 4], (int)zd[6 * rg + 5], 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    switch (rg) {          // (the byte of the scale word is an immediate of the instruction)
                        case 0: acc[g][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(az, bx[j], acc[g][j], 2, 4, 0, zscale, 0, 127); break;
                        case 1: acc[g][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(az, bx[j], acc[g][j], 2, 4, 1, zscale, 0, 127); break;
                        case 2: acc[g][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(az, bx[j], acc[g][j], 2, 4, 2, zscale, 0, 127); break;
                        default: acc[g][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(az, bx[j], acc[g][j], 2, 4, 3, zscale, 0, 127); break;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- fold the piece parities (rows c and c + 8 sit 32 lanes apart) into an LDS image of the block's dV rows [512 SNPs][CP], then
    // every thread handles whole float4s of that contiguous region: full 16 B/lane lines for the gradient store, or -- single-GPU
    // step -- for Adam on these V rows (3 reads + 3 writes per element: it has to be coalesced) ----
    float* s_dv = reinterpret_cast<float*>(&s_xt[0][0]);       // (the barrier that ends the last tile has passed)
#pragma unroll
    for (int g = 0; g < EB_G; ++g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = acc[g][j][r];
                o[r] = u + __shfl_xor(u, 32, 64);
            }
            const int ml = (wave * (16 * EB_G) + g * 16 + mcol) * 4 + j;          // SNP within the block's 512
            if (q < 2 && 4 * q < CP) *reinterpret_cast<float4*>(s_dv + ml * CP + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    constexpr int ROW4 = CP / 4;
    constexpr int CH_F4 = EB_CHUNK_SNPS * ROW4;
    const int64_t m0 = chunk * EB_CHUNK_SNPS;
    auto finish = [&](int e, const float4 g4) {                       // row piece e of the chunk, gradient complete
        const int64_t m = m0 + e / ROW4;
        if (m < M) {
            const int64_t o = m * CP + 4 * (e % ROW4);
            if (ad.m != nullptr) adam_float4(Vrw + o, g4, ad.m + o, ad.v + o, ad.step_size, ad.inv_bc2, ad.grad_scale, false);
            else *reinterpret_cast<float4*>(dV + o) = g4;
        }
    };
    if constexpr (!SLICED) {
        for (int e = tid; e < CH_F4; e += 256) finish(e, *reinterpret_cast<const float4*>(s_dv + 4 * e));
        return;
    }
    // ---- S > 1: park the partial, be counted, the last one adds them up (pass 2's hand-off: write-through stores, vmcnt(0), a device-scope
    // counter, device-scope loads)
    constexpr int SLAB_F = p3_slab_floats(CP);
    float* const mine = slab + ((int64_t)slice * nchunks + chunk) * SLAB_F;
    for (int e = tid; e < CH_F4; e += 256) {
        const float4 g4 = *reinterpret_cast<const float4*>(s_dv + 4 * e);
        __hip_atomic_store(mine + 4 * e + 0, g4.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 4 * e + 1, g4.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 4 * e + 2, g4.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 4 * e + 3, g4.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this wave's stores have been acknowledged ...
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(&slice_cnt[chunk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the block is counted
        s_last = old == n_slices - 1;
        if (s_last) __hip_atomic_store(&slice_cnt[chunk], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;                                              // block-uniform
    for (int e = tid; e < CH_F4; e += 256) {
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sl = 0; sl < n_slices; ++sl) {                       // slice order, whoever is last
            const float* part = slab + ((int64_t)sl * nchunks + chunk) * SLAB_F + 4 * e;
            g4.x += __hip_atomic_load(part + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g4.y += __hip_atomic_load(part + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g4.z += __hip_atomic_load(part + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g4.w += __hip_atomic_load(part + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        finish(e, g4);
    }
}

// ---- SNPs per lane in pass 2 as a function of the padded head width (register budget ~128) ----
constexpr int dec_spl(int kp) { return kp <= 8 ? 8 : (kp <= 16 ? 4 : (kp <= 32 ? 2 : 1)); }

// Adam on n floats for the variants of passes 2 and 3 that have no fused epilogue (same element function)
__global__ __launch_bounds__(256) void adam_range_kernel(float* __restrict__ p, const float* __restrict__ g, AdamFused ad, int64_t n, int clamp01) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        float mq = ad.m[e], vq = ad.v[e];
        p[e] = adam_element(p[e], g[e], mq, vq, ad.step_size, ad.inv_bc2, ad.grad_scale, clamp01 != 0);
        ad.m[e] = mq; ad.v[e] = vq;
    }
}
static int launch_adam_range(float* p, const float* g, AdamFused ad, int64_t n, hipStream_t st, int clamp01) {
    int64_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(adam_range_kernel, dim3((unsigned)gx), dim3(256), 0, st, p, g, ad, n, clamp01);
    return check_launch("adam_range");
}

// rows idx[0..b) of xp, the byte columns that hold SNPs [0, M) -> rows 0..b-1 of the tiled copy xg (xg_piece), missing calls set to 0: what
// the bf16 pass-2 kernel writes as a by-product, as a kernel of its own for the K > 16 / A-B-reference variants of pass 2
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* __restrict__ xp, int64_t ld, const int32_t* __restrict__ idx,
                                                          int b, int64_t pieces, uint8_t* __restrict__ xg) {
    const int64_t row = idx[blockIdx.y];
    for (int64_t p = blockIdx.x * 256 + threadIdx.x; p < pieces; p += (int64_t)gridDim.x * 256)
    {
        const uint4 w = *reinterpret_cast<const uint4*>(xp + row * ld + p * 16);
        *reinterpret_cast<uint4*>(xg_piece(xg, b, (int)blockIdx.y, p * 16)) = make_uint4(clean_codes(w.x), clean_codes(w.y), clean_codes(w.z), clean_codes(w.w));
    }
}

static int launch_gather_rows(const uint8_t* xp, int64_t ld, const int32_t* idx, int b, int64_t M, uint8_t* xg, hipStream_t st) {
    const int64_t pieces = (M + 63) / 64;                          // 16-byte pieces that hold SNPs below M (ld is a multiple of 16)
    int64_t gx = (pieces + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)gx, (unsigned)b), dim3(256), 0, st, xp, ld, idx, b, pieces, xg);
    return check_launch("gather_rows");
}

// measurement only: where pass 2's probe block leaves its two counter differences (NULL: no probe).  Process-wide, like the launches' device.
static std::atomic<unsigned long long*> g_clock_probe{nullptr};
extern "C" void nadm_clock_probe(uint64_t* out2_dev) { g_clock_probe.store(reinterpret_cast<unsigned long long*>(out2_dev)); }

template <int KP>
static int launch_decode_mfma(const uint8_t* xp, int64_t ld, const int32_t* idx, int b, int64_t M, float* P,
                              const float* Q, int SP, float* dP, float* dqpart, float* losspart, int with_loss,
                              hipStream_t st, uint8_t* xg, AdamFused ad, const uint4* qimg, int n_slices, float* slab, int* slice_cnt) {
    static_assert(KP <= 16 && mf_chunk_snps(KP) == BF_WAVES * 16 * BF_NTW, "chunking published by nadm_decode_chunk_snps");
    const int64_t chunks = (M + mf_chunk_snps(KP) - 1) / mf_chunk_snps(KP);
    dim3 grid((unsigned)chunks, (unsigned)n_slices), block(64 * BF_WAVES);
    constexpr bool CAN_IMG = BF_TS == QI_TS;                 // (variant builds with another tile depth: decode_bce_impl refuses qimg)
    unsigned long long* const clk_probe = g_clock_probe.load();
#define NADM_P2_LAUNCH(...)                                                                                                                     \
    do {                                                                                                                                       \
        if (n_slices > 1)                                                                                                                      \
            hipLaunchKernelGGL((decode_bce_bf16_kernel<KP, __VA_ARGS__, true>), grid, block, 0, st, xp, ld, idx, b, M, P, Q, SP, dP, dqpart,   \
                               losspart, xg, ad, qimg, slab, slice_cnt, (unsigned long long*)nullptr);                                        \
        else if (clk_probe != nullptr)                                                                                                         \
            hipLaunchKernelGGL((decode_bce_bf16_kernel<KP, __VA_ARGS__, false, true>), grid, block, 0, st, xp, ld, idx, b, M, P, Q, SP, dP,    \
                               dqpart, losspart, xg, ad, qimg, slab, slice_cnt, clk_probe);                                                    \
        else                                                                                                                                   \
            hipLaunchKernelGGL((decode_bce_bf16_kernel<KP, __VA_ARGS__, false>), grid, block, 0, st, xp, ld, idx, b, M, P, Q, SP, dP, dqpart,  \
                               losspart, xg, ad, qimg, slab, slice_cnt, (unsigned long long*)nullptr);                                        \
    } while (0)
    if (with_loss & 2) {        // loss value with P possibly outside [0, 1] (before the first restrict_P)
        if (qimg) NADM_P2_LAUNCH(true, false, CAN_IMG); else NADM_P2_LAUNCH(true, false, false);
    } else if (with_loss) {
        if (qimg) NADM_P2_LAUNCH(true, true, CAN_IMG); else NADM_P2_LAUNCH(true, true, false);
    } else {
        if (qimg) NADM_P2_LAUNCH(false, true, CAN_IMG); else NADM_P2_LAUNCH(false, true, false);
    }
#undef NADM_P2_LAUNCH
    return check_launch("decode_bce_bf16");
}

template <int KP>
static int launch_decode(const uint8_t* xp, int64_t ld, const int32_t* idx, int b, int64_t M, float* P,
                         const float* Q, int SP, float* dP, float* dqpart, float* losspart, int with_loss,
                         hipStream_t st, uint8_t* xg, AdamFused ad) {
    constexpr int SPL = dec_spl(KP);
    const int64_t chunks = (M + 256 * SPL - 1) / (256 * SPL);
    dim3 grid((unsigned)chunks), block(256);
    if (with_loss)
        hipLaunchKernelGGL((decode_bce_kernel<KP, SPL, true>), grid, block, 0, st, xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart);
    else
        hipLaunchKernelGGL((decode_bce_kernel<KP, SPL, false>), grid, block, 0, st, xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart);
    if (check_launch("decode_bce")) return 1;
    if (xg && launch_gather_rows(xp, ld, idx, b, M, xg, st)) return 1;
    return ad.m ? launch_adam_range(P, dP, ad, M * KP, st, 1) : 0;
}

}  // namespace nadm

using namespace nadm;

// The matrix-core kernels of passes 1 and 3 cover CP <= 8; wider encoders (n_components > 8) run the VALU kernels.
// The chunk count must not depend on CP (callers size zpart before they pass CP): both pass-1 kernels write 2048-SNP chunks.
static_assert(ENC_TB * 4 * ENC_TILES == EM_CHUNK_SNPS, "one chunking for both pass-1 kernels");
extern "C" int64_t nadm_encode_chunks(int64_t M) { return (M + EM_CHUNK_SNPS - 1) / EM_CHUNK_SNPS; }

extern "C" int32_t nadm_decode_chunk_snps(int kp) {
    if (kp <= 16) return mf_chunk_snps(kp);
    return 256 * dec_spl(kp);
}

extern "C" int64_t nadm_decode_chunks(int64_t M, int kp) {
    const int64_t c = nadm_decode_chunk_snps(kp);
    return (M + c - 1) / c;
}

// compute units of the current device (cached per device ordinal; 256 if the runtime cannot say)
static int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

static int enc_rows_per_block(int b) {
    const int max_rows = 832;                           // 13 waves; LDS = rows * 132 B <= 110 KB
    const int gy = (b + max_rows - 1) / max_rows;
    const int per = (b + gy - 1) / gy;
    return ((per + 63) / 64) * 64;
}

static int encode_fwd_impl(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                           const float* V, int32_t CP, float* zpart, void* stream, uint32_t missing_bf16,
                           SmallSide ss = SmallSide{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.f, 0.f, 0.f}, int64_t total_chunks = 0) {
    if (!xp || !idx || !V || !zpart) return fail("nadm_encode_fwd: null pointer");
    if (b <= 0 || M <= 0) return fail("nadm_encode_fwd: empty batch or M");
    if (ld % 16 != 0 || ld * 4 < M) return fail("nadm_encode_fwd: ld must be a multiple of 16 and >= ceil(M/4)");
    if (ld >> 32) return fail("nadm_encode_fwd: rows of 4 GiB and more (ld >= 2^32) are not supported");      // the matrix pass forms row addresses from 32-bit factors
    const int rpb = enc_rows_per_block(b);
    dim3 grid((unsigned)nadm_encode_chunks(M), (unsigned)((b + rpb - 1) / rpb)), block(rpb);
    const size_t lds = (size_t)rpb * ENC_LDW * 4;
    hipStream_t st = (hipStream_t)stream;
    if (CP <= 8) {
        // Every block first splits its 2048 x CP slice of V into bf16 operands (about as much work as 11 sample tiles), so a block
        // should see many tiles; but the 512 block slots of the chip (2 per CU) want to be filled, and a partly filled last round
        // costs most of a full one.  grid.y = the number of batch splits (never fewer than 4 tiles per block) that minimises
        //   rounds_eff(chunks * gy / slots) * (11.3 + tiles / gy),   slots = 2 blocks per CU of the device the launch goes to (512 on an
        //   MI355X in SPX mode; a partition or another SKU reports its own CU count)
        // with rounds_eff() read off measured launches (profiles/r03_p1_batch_splits.txt: b = 800, M = 500k .. 1M, grid.y = 1..4; the
        // model reproduces all twenty within 2 us except one): a block alone on its CU runs 1.5 x faster, a round that is a
        // little over-full costs 1.4 rounds.  M = 500k: 2 splits (43.6 us; 1 / 3 / 4: 49.1 / 47.7 / 49.0), 600k: 3 (56.1; 60.0 with
        // the 2 that r02's rule "as few splits as give 480 blocks" picked), 800k: 1 (70.9; 73.3 with 2), 1M: 1 (73.6).
        const int64_t chunks = nadm_encode_chunks(M);
        const int ntiles = (b + 15) / 16;
        auto rounds_eff = [](double x) {
            if (x <= 0.5) return 0.67;
            if (x <= 1.0) return 0.92 + 0.16 * (x - 0.5);
            if (x <= 1.5) return 1.40;
            if (x <= 1.95) return 1.70;
            if (x <= 2.0) return 2.00;
            return 0.45 + 0.72 * x;
        };
        const double slots = 2.0 * (double)device_cu_count();
        // (a launch that is one PART of a pass shares the device with the other parts: the split is the whole pass's, nadm_encode_fwd_part)
        const int64_t fill_chunks = total_chunks > chunks ? total_chunks : chunks;
        int64_t gy = 1;
        double best = 1e30;
        for (int64_t g = 1; g <= 64 && g <= (ntiles + 3) / 4; ++g) {
            const double cost = rounds_eff((double)(fill_chunks * g) / slots) * (11.3 + (double)((ntiles + g - 1) / g));
            if (cost < best) { best = cost; gy = g; }
        }
        int tpb = (int)((ntiles + gy - 1) / gy);
        if (tpb > EM_TILES_PER_BLOCK) tpb = EM_TILES_PER_BLOCK;
        gy = (ntiles + tpb - 1) / tpb;
        const int side_blocks = ss.part ? (ss.n + 511) / 512 : 0;
        dim3 g2((unsigned)(chunks * gy + side_blocks)), b2(512);
        if (CP == 4) hipLaunchKernelGGL((encode_fwd_mfma_kernel<4>), g2, b2, 0, st, xp, ld, idx, b, M, V, zpart, tpb, missing_bf16, (int)chunks, (int)gy, ss);
        else hipLaunchKernelGGL((encode_fwd_mfma_kernel<8>), g2, b2, 0, st, xp, ld, idx, b, M, V, zpart, tpb, missing_bf16, (int)chunks, (int)gy, ss);
        return check_launch("encode_fwd_mfma");
    }
    if (ss.part) return fail("nadm_encode_fwd_small: only the matrix-core pass (CP <= 8) hosts the side blocks");
#define ENC_LAUNCH(cp)                                                                                                 \
    {                                                                                                                  \
        if (lds > 48 * 1024) {                                                                                         \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_fwd_kernel<cp>),            \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            if (e != hipSuccess) {                                                                                     \
                snprintf(err_buf(), 512, "nadm_encode_fwd: cannot raise dynamic LDS limit to %zu: %s", lds,            \
                         hipGetErrorString(e));                                                                        \
                return 1;                                                                                              \
            }                                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((encode_fwd_kernel<cp>), grid, block, lds, st, xp, ld, idx, b, M, V, zpart, rpb);           \
    }
#define ENC_CASE(cp)                                                                                                   \
    case cp:                                                                                                           \
        ENC_LAUNCH(cp)                                                                                                 \
        break;
    switch (CP) {
        ENC_CASE(12) ENC_CASE(16) ENC_CASE(24) ENC_CASE(32)          // (CP 4, 8: the matrix-core kernel above)
        default: return fail("nadm_encode_fwd: unsupported CP (4,8,12,16,24,32)");
    }
#undef ENC_LAUNCH
#undef ENC_CASE
    return check_launch("encode_fwd");
}

static int adam_fused_args(const nadm_adam_t* adam, const char* who, AdamFused* out) {
    *out = AdamFused{nullptr, nullptr, 0.f, 0.f, 0.f};
    if (!adam) return 0;
    if (!adam->m || !adam->v) return fail(who);
    if (adam->step < 1) return fail("nadm_*_step: Adam step is 1-based");
    if (((uintptr_t)adam->m | (uintptr_t)adam->v) & 15) return fail("nadm_*_step: Adam state must be 16-byte aligned");
    out->m = adam->m; out->v = adam->v; out->grad_scale = adam->grad_scale;
    if (adam->reserved != 0) return fail("nadm_*_step: nadm_adam_t.reserved must be 0");
    adam_scalars(adam->lr, adam->step, &out->step_size, &out->inv_bc2);
    return 0;
}

extern "C" int nadm_encode_fwd(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                               const float* V, int32_t CP, float* zpart, void* stream) {
    return encode_fwd_impl(xp, ld, idx, b, M, V, CP, zpart, stream, 0u);
}

extern "C" int nadm_encode_fwd_part(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                    const float* V, int32_t CP, float* zpart, int64_t total_chunks, void* stream) {
    return encode_fwd_impl(xp, ld, idx, b, M, V, CP, zpart, stream, 0u, SmallSide{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.f, 0.f, 0.f}, total_chunks);
}

extern "C" int nadm_encode_fwd_small(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                     const float* V, int32_t CP, float* zpart, const float* small_part, int32_t splits, int32_t n_small,
                                     float* grad_small, float* small, const nadm_adam_t* adam, void* stream) {
    if (!small_part || !grad_small) return fail("nadm_encode_fwd_small: null pointer");
    if (splits <= 0 || n_small <= 0) return fail("nadm_encode_fwd_small: empty");
    SmallSide ss{small_part, grad_small, small, nullptr, nullptr, splits, n_small, 0.f, 0.f, 0.f};
    if (adam) {
        if (!adam->m || !adam->v || !small) return fail("nadm_encode_fwd_small: Adam state / parameters are NULL");
        if (adam->step < 1) return fail("nadm_encode_fwd_small: Adam step is 1-based");
        ss.m = adam->m; ss.v = adam->v; ss.grad_scale = adam->grad_scale;
        adam_scalars(adam->lr, adam->step, &ss.step_size, &ss.inv_bc2);
    }
    return encode_fwd_impl(xp, ld, idx, b, M, V, CP, zpart, stream, 0u, ss);
}

extern "C" int nadm_pca_project(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                const float* V, int32_t CP, float* zpart, void* stream) {
    if (CP > 8) return fail("nadm_pca_project: only the matrix-core pass (CP <= 8) has the missing = 1.5 variant");
    return encode_fwd_impl(xp, ld, idx, b, M, V, CP, zpart, stream, 0x3FC0u);
}

static int decode_bce_impl(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                           float* P, int32_t kp, const float* Q, int32_t SP, float* dP, float* dqpart,
                           float* losspart, int32_t with_loss, void* stream, uint8_t* xg, AdamFused ad, const uint4* qimg = nullptr,
                           int32_t n_slices = 1, float* slab = nullptr, int32_t* slice_cnt = nullptr) {
    if (!xp || !idx || !P || !Q || !dP || !dqpart) return fail("nadm_decode_bce: null pointer");
    if (ld >> 32) return fail("nadm_decode_bce: rows of 4 GiB and more (ld >= 2^32) are not supported");      // the matrix pass forms row addresses from 32-bit factors
    if (n_slices < 1 || n_slices > NADM_MAX_P2_SLICES) return fail("nadm_decode_bce_sliced: n_slices must be in 1..8");
    if (n_slices > 1) {
        if (kp > 16) return fail("nadm_decode_bce_sliced: sample slices exist for padded K <= 16 only");
        if (!slab || !slice_cnt || ((uintptr_t)slab & 15)) return fail("nadm_decode_bce_sliced: n_slices > 1 needs the slab (16-byte aligned) and the counters");
        const int tiles = (b + NADM_BF_TS - 1) / NADM_BF_TS, tps = (tiles + n_slices - 1) / n_slices;
        if ((int64_t)(n_slices - 1) * tps >= tiles) return fail("nadm_decode_bce_sliced: a slice would be empty (take n_slices from nadm_decode_slices)");
    }
    if (qimg && (kp > 16 || NADM_BF_TS != nadm::QI_TS)) return fail("nadm_decode_bce_images: Q images exist for padded K <= 16 only");
    if ((uintptr_t)qimg & 15) return fail("nadm_decode_bce_images: qimg must be 16-byte aligned");
    if (with_loss < 0 || with_loss > 3) return fail("nadm_decode_bce: with_loss is a bit set (1: loss value, 2: P may lie outside [0,1])");
    if (!(with_loss & 1)) with_loss = 0;
    if (with_loss && !losspart) return fail("nadm_decode_bce: with_loss needs losspart");
    if (b <= 0 || M <= 0) return fail("nadm_decode_bce: empty batch or M");
    if (ld % 16 != 0 || ld * 4 < M) return fail("nadm_decode_bce: ld must be a multiple of 16 and >= ceil(M/4)");
    hipStream_t st = (hipStream_t)stream;
    if (kp <= 16) {
        switch (kp) {
            case 4: return launch_decode_mfma<4>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad, qimg, n_slices, slab, slice_cnt);
            case 8: return launch_decode_mfma<8>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad, qimg, n_slices, slab, slice_cnt);
            case 12: return launch_decode_mfma<12>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad, qimg, n_slices, slab, slice_cnt);
            case 16: return launch_decode_mfma<16>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad, qimg, n_slices, slab, slice_cnt);
            default: return fail("nadm_decode_bce: unsupported padded K (use nadm_pad_k)");
        }
    }
    switch (kp) {
        case 24: return launch_decode<24>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad);
        case 32: return launch_decode<32>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad);
        case 48: return launch_decode<48>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad);
        case 64: return launch_decode<64>(xp, ld, idx, b, M, P, Q, SP, dP, dqpart, losspart, with_loss, st, xg, ad);
        default: return fail("nadm_decode_bce: unsupported padded K (use nadm_pad_k)");
    }
}

extern "C" int nadm_decode_bce(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                               const float* P, int32_t kp, const float* Q, int32_t SP, float* dP, float* dqpart,
                               float* losspart, int32_t with_loss, void* stream) {
    return decode_bce_impl(xp, ld, idx, b, M, const_cast<float*>(P), kp, Q, SP, dP, dqpart, losspart, with_loss, stream, nullptr, AdamFused{nullptr, nullptr, 0.f, 0.f, 0.f});
}

extern "C" int nadm_decode_bce_gather(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                      const float* P, int32_t kp, const float* Q, int32_t SP, float* dP, float* dqpart,
                                      float* losspart, int32_t with_loss, uint8_t* xg, void* stream) {
    if (!xg) return fail("nadm_decode_bce_gather: null pointer");
    return decode_bce_impl(xp, ld, idx, b, M, const_cast<float*>(P), kp, Q, SP, dP, dqpart, losspart, with_loss, stream, xg, AdamFused{nullptr, nullptr, 0.f, 0.f, 0.f});
}

extern "C" int nadm_decode_bce_step(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                    float* P, int32_t kp, const float* Q, int32_t SP, float* dP, float* dqpart,
                                    float* losspart, int32_t with_loss, uint8_t* xg, const nadm_adam_t* adam, void* stream) {
    AdamFused ad;
    if (adam_fused_args(adam, "nadm_decode_bce_step: Adam state is NULL", &ad)) return 1;
    if ((uintptr_t)P & 15) return fail("nadm_decode_bce_step: P must be 16-byte aligned");
    return decode_bce_impl(xp, ld, idx, b, M, P, kp, Q, SP, dP, dqpart, losspart, with_loss, stream, xg, ad);
}

extern "C" int64_t nadm_q_image_bytes(int32_t b) { return (int64_t)((b + nadm::QI_TS - 1) / nadm::QI_TS) * nadm::QI_TILE_U4 * 16; }

extern "C" int nadm_decode_bce_images(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                      float* P, int32_t kp, const float* Q, int32_t SP, float* dP, float* dqpart,
                                      float* losspart, int32_t with_loss, uint8_t* xg, const nadm_adam_t* adam, const void* qimg,
                                      void* stream) {
    AdamFused ad;
    if (adam_fused_args(adam, "nadm_decode_bce_images: Adam state is NULL", &ad)) return 1;
    if (ad.m && ((uintptr_t)P & 15)) return fail("nadm_decode_bce_images: P must be 16-byte aligned");
    if (!qimg) return fail("nadm_decode_bce_images: qimg is NULL (use nadm_decode_bce_step)");
    return decode_bce_impl(xp, ld, idx, b, M, P, kp, Q, SP, dP, dqpart, losspart, with_loss, stream, xg, ad, static_cast<const uint4*>(qimg));
}

// ---- pass 2 with the batch's sample tiles dealt to n_slices blocks per SNP chunk (see decode_bce_bf16_kernel)
#ifdef NADM_TEST_HOOKS          // the test build only (csrc/build.sh -> libnadm_testhooks.so): the shipping library has no way to override the rule below
static std::atomic<int> g_force_slices{0};
extern "C" void nadm_test_force_slices(int32_t n) { g_force_slices.store(n < 0 ? 0 : (n > NADM_MAX_P2_SLICES ? NADM_MAX_P2_SLICES : n)); }
#endif

extern "C" int32_t nadm_decode_slices(int32_t b, int64_t M, int32_t kp) {
    if (kp > 16 || b <= 0 || M <= 0) return 1;
    const int64_t chunks = (M + mf_chunk_snps(8) - 1) / mf_chunk_snps(8);
    const int tiles = (b + NADM_BF_TS - 1) / NADM_BF_TS;
    int s = 0;
#ifdef NADM_TEST_HOOKS
    s = g_force_slices.load();
#endif
    if (s == 0) {
        // Measured (profiles/r05_ablations.txt item 11): below ~130k SNPs pass 2 is bound by the serial chain of ONE block -- its prologue
        // (P rows into operands, ~1.5 tile-times) + 13 tiles at b = 800 = 49 us whether 98 or 196 blocks run -- not by the chip; slices of
        // ~4-5 tiles shorten the chain (M = 25k: 49 -> 32 us, 50k: 50 -> 43, 100k: 73.5 -> 66.5; 6400 rows x 62.5k SNPs, the SNP-sharded
        // rank of configs[3]: 326 -> 244).  A slice costs a prologue and the launch ~5 us for the hand-off: short batches (< 8 tiles: b = 400
        // at M = 100k 44.4 -> 45.3, b = 200 at 50k 20.9 -> 23.6) and launches of more than ~1.7 rounds of blocks do not pay; from 150k SNPs
        // on the chunks fill the chip and nothing changes.
        if (chunks >= 512 || tiles < 8) return 1;
        s = (int)((2 * tiles + 4) / 9);                             // round(tiles / 4.5)
        const int64_t lim = 1280 / chunks;                           // blocks <= 1280 = 1.7 rounds of 768
        if (s > lim) s = (int)lim;
        if (s > NADM_MAX_P2_SLICES) s = NADM_MAX_P2_SLICES;
    }
    if (s > tiles) s = tiles;
    if (s < 2) return 1;
    const int tps = (tiles + s - 1) / s;
    return (tiles + tps - 1) / tps;                                   // no empty slice
}

extern "C" int32_t nadm_decode_slices_max(int32_t bmax, int64_t M, int32_t kp) {       // the most slices any batch of up to bmax rows is cut into
    int32_t mx = 1;
    for (int32_t b = bmax; b > 0; b -= NADM_BF_TS) { const int32_t s = nadm_decode_slices(b, M, kp); if (s > mx) mx = s; }
    return mx;
}

extern "C" int64_t nadm_decode_slab_floats(int64_t M, int32_t kp, int32_t n_slices) {
    if (kp > 16 || n_slices < 2 || M <= 0) return 0;
    return (int64_t)n_slices * ((M + mf_chunk_snps(8) - 1) / mf_chunk_snps(8)) * p2_slab_floats(kp);
}

extern "C" int nadm_decode_bce_sliced(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                      float* P, int32_t kp, const float* Q, int32_t SP, float* dP, float* dqpart,
                                      float* losspart, int32_t with_loss, uint8_t* xg, const nadm_adam_t* adam, const void* qimg,
                                      int32_t n_slices, float* slab, int32_t* counters, void* stream) {
    AdamFused ad;
    if (adam_fused_args(adam, "nadm_decode_bce_sliced: Adam state is NULL", &ad)) return 1;
    if (ad.m && ((uintptr_t)P & 15)) return fail("nadm_decode_bce_sliced: P must be 16-byte aligned");
    return decode_bce_impl(xp, ld, idx, b, M, P, kp, Q, SP, dP, dqpart, losspart, with_loss, stream, xg, ad, static_cast<const uint4*>(qimg),
                           n_slices, slab, counters);
}

extern "C" int64_t nadm_batch_copy_bytes(int32_t b, int64_t M) { return ((M + 4 * nadm::XG_TILE_COLS - 1) / (4 * nadm::XG_TILE_COLS)) * (int64_t)b * nadm::XG_TILE_COLS; }
extern "C" int64_t nadm_dz_image_tile_bytes(void) { return (int64_t)nadm::DZI_TILE_U4 * 16; }
extern "C" int64_t nadm_dz_image_bytes(int32_t b) { return (int64_t)((b + nadm::DZI_TS - 1) / nadm::DZI_TS) * nadm::DZI_TILE_U4 * 16; }

extern "C" int nadm_dz_image(const float* dZ, int32_t b, int32_t CP, void* dzimg, void* stream) {
    if (!dZ || !dzimg) return fail("nadm_dz_image: null pointer");
    if (b <= 0 || CP <= 0 || CP > 8) return fail("nadm_dz_image: b > 0 and 0 < CP <= 8 (the matrix-core pass 3)");
    if ((uintptr_t)dzimg & 15) return fail("nadm_dz_image: the image must be 16-byte aligned");
    hipLaunchKernelGGL(dz_image_kernel, dim3((unsigned)((b + DZI_TS - 1) / DZI_TS)), dim3(256), 0, (hipStream_t)stream, dZ, b, CP, (uint4*)dzimg);
    return check_launch("dz_image");
}

static int encode_bwd_impl(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                           const float* dZ, const void* dzimg, int32_t CP, float* dV, void* stream, uint32_t missing_bf16, int32_t flags = 0,
                           float* Vrw = nullptr, AdamFused ad = AdamFused{nullptr, nullptr, 0.f, 0.f, 0.f},
                           const nadm_mlp_weights_t* mw = nullptr, int32_t n_slices = 1, float* slab = nullptr, int32_t* counters = nullptr) {
    if (!xp || !idx || !dZ || !dV) return fail("nadm_encode_bwd: null pointer");
    if (CP > 8 && (flags & NADM_X_CLEAN)) return fail("nadm_encode_bwd: the batch copy (NADM_X_CLEAN) is tiled for the matrix-core pass, C <= 8");
    if (CP <= 8 && (!dzimg || ((uintptr_t)dzimg & 15)))
        return fail("nadm_encode_bwd: C <= 8 runs on the FP4 x FP6 matrix instruction and needs the operand image of dZ (nadm_dz_image), 16-byte aligned");
    if (b <= 0 || M <= 0) return fail("nadm_encode_bwd: empty batch or M");
    if (ld % 16 != 0 || ld * 4 < M) return fail("nadm_encode_bwd: ld must be a multiple of 16 and >= ceil(M/4)");
    dim3 grid((unsigned)((M + 1023) / 1024)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (CP <= 8) {
        MlpSide side;
        memset(&side, 0, sizeof(side));
        int64_t extra = 0;
        if (mw) {
            side.hd = *mw->hd; side.b = b; side.gx = (mw->hd->Hd + 255) / 256;
            side.Zn = mw->Zn; side.H = mw->H; side.dL = mw->dL; side.dHpre = mw->dHpre; side.dgp = mw->dgp; side.small_part = mw->small_part;
            extra = (int64_t)side.gx * nadm_sample_splits(b);
        }
        if (n_slices < 1 || n_slices > NADM_MAX_P2_SLICES) return fail("nadm_encode_bwd_sliced: n_slices must be in 1..8");
        const int ntiles = (b + DZI_TS - 1) / DZI_TS;
        if (n_slices > 1 && ((n_slices - 1) * ((ntiles + n_slices - 1) / n_slices) >= ntiles)) return fail("nadm_encode_bwd_sliced: an empty sample slice");
        if (n_slices > 1 && (!slab || !counters || ((uintptr_t)slab & 15))) return fail("nadm_encode_bwd_sliced: the slab (16-byte aligned) and the counters are NULL");
        dim3 g2((unsigned)(((M + EB_CHUNK_SNPS - 1) / EB_CHUNK_SNPS) * n_slices + extra));
#define NADM_P3_LAUNCH(CPV, CL, SL) hipLaunchKernelGGL((encode_bwd_fp4_kernel<CPV, CL, SL>), g2, block, 0, st, xp, ld, idx, b, M, (const uint4*)dzimg, dV, missing_bf16, Vrw, ad, side, slab, counters, (int)n_slices)
        const bool clean = (flags & NADM_X_CLEAN) != 0 && missing_bf16 == 0u;
        if (n_slices > 1) {
            if (CP == 4) { if (clean) NADM_P3_LAUNCH(4, true, true); else NADM_P3_LAUNCH(4, false, true); }
            else { if (clean) NADM_P3_LAUNCH(8, true, true); else NADM_P3_LAUNCH(8, false, true); }
        } else if (CP == 4) { if (clean) NADM_P3_LAUNCH(4, true, false); else NADM_P3_LAUNCH(4, false, false); }
        else { if (clean) NADM_P3_LAUNCH(8, true, false); else NADM_P3_LAUNCH(8, false, false); }
#undef NADM_P3_LAUNCH
        return check_launch("encode_bwd_fp4");
    }
    switch (CP) {
        case 12: hipLaunchKernelGGL((encode_bwd_kernel<12>), grid, block, 0, st, xp, ld, idx, b, M, dZ, dV); break;
        case 16: hipLaunchKernelGGL((encode_bwd_kernel<16>), grid, block, 0, st, xp, ld, idx, b, M, dZ, dV); break;
        case 24: hipLaunchKernelGGL((encode_bwd_kernel<24>), grid, block, 0, st, xp, ld, idx, b, M, dZ, dV); break;
        case 32: hipLaunchKernelGGL((encode_bwd_kernel<32>), grid, block, 0, st, xp, ld, idx, b, M, dZ, dV); break;
        default: return fail("nadm_encode_bwd: unsupported CP (4,8,12,16,24,32)");
    }
    if (check_launch("encode_bwd")) return 1;
    if (ad.m && launch_adam_range(Vrw, dV, ad, M * CP, st, 0)) return 1;  // fp32 variants of pass 3: stand-alone update
    return mw ? nadm_mlp_bwd_weight_parts(mw->hd, b, mw->Zn, mw->H, mw->dL, mw->dHpre, mw->dgp, mw->small_part, stream) : 0;
}

extern "C" int nadm_encode_bwd_step(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                    const float* dZ, const void* dzimg, int32_t CP, float* V, float* dV, const nadm_adam_t* adam,
                                    const nadm_mlp_weights_t* weights, int32_t flags, void* stream) {
    AdamFused ad;
    if (adam_fused_args(adam, "nadm_encode_bwd_step: Adam state is NULL", &ad)) return 1;
    if (ad.m && (!V || ((uintptr_t)V & 15))) return fail("nadm_encode_bwd_step: V must be non-NULL and 16-byte aligned");
    if (weights && (!weights->hd || !weights->Zn || !weights->H || !weights->dL || !weights->dHpre || !weights->dgp || !weights->small_part))
        return fail("nadm_encode_bwd_step: null pointer in the MLP weight-gradient arguments");
    return encode_bwd_impl(xp, ld, idx, b, M, dZ, dzimg, CP, dV, stream, 0u, flags, V, ad, weights);
}

// ---- pass 3 with the batch's 128-sample tiles dealt to n_slices blocks per SNP chunk (see encode_bwd_fp4_kernel)
#ifdef NADM_TEST_HOOKS
static std::atomic<int> g_force_p3_slices{0};
extern "C" void nadm_test_force_p3_slices(int32_t n) { g_force_p3_slices.store(n < 0 ? 0 : (n > NADM_MAX_P2_SLICES ? NADM_MAX_P2_SLICES : n)); }
#endif

extern "C" int32_t nadm_encode_slices(int32_t b, int64_t M, int32_t cp) {
    if (cp > 8 || b <= 0 || M <= 0) return 1;
    const int64_t chunks = (M + EB_CHUNK_SNPS - 1) / EB_CHUNK_SNPS;
    const int tiles = (b + DZI_TS - 1) / DZI_TS;
    int s = 0;
#ifdef NADM_TEST_HOOKS
    s = g_force_p3_slices.load();
#endif
    if (s == 0) {
        // Measured (profiles/r06_p3_slices.txt): a launch of ~120 chunk blocks that each walk 50 tiles (6400 rows x 62.5k SNPs, the SNP-sharded
        // rank of configs[3] on 8 GPUs) lasts as long as one block's chain, 77 us where the same genotypes as 977 blocks take 41; two slices
        // 57 us, three 62, four 64, eight 76 -- every slice pays a prologue and the hand-off (park 16 KB through to memory, be counted, the
        // last one reads them all back and runs Adam: ~8 us of tail).  At 3200 x 125k (245 chunks x 25 tiles, 46 us) any cut loses: 53 / 65.
        // So: two slices, only for few chunks AND long chains; every single-GPU BASELINE shape stays whole.
        if (chunks >= 192 || tiles < 32) return 1;
        s = 2;
    }
    if (s > tiles) s = tiles;
    if (s < 2) return 1;
    const int tps = (tiles + s - 1) / s;
    return (tiles + tps - 1) / tps;                                   // no empty slice
}

extern "C" int32_t nadm_encode_slices_max(int32_t bmax, int64_t M, int32_t cp) {
    int32_t mx = 1;
    for (int32_t b = bmax; b > 0; b -= DZI_TS) { const int32_t s = nadm_encode_slices(b, M, cp); if (s > mx) mx = s; }
    return mx;
}

extern "C" int64_t nadm_encode_slab_floats(int64_t M, int32_t cp, int32_t n_slices) {
    if (cp > 8 || n_slices < 2 || M <= 0) return 0;
    return (int64_t)n_slices * ((M + EB_CHUNK_SNPS - 1) / EB_CHUNK_SNPS) * p3_slab_floats(cp);
}

extern "C" int64_t nadm_encode_bwd_chunks(int64_t M) { return (M + EB_CHUNK_SNPS - 1) / EB_CHUNK_SNPS; }

extern "C" int nadm_encode_bwd_sliced(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                      const float* dZ, const void* dzimg, int32_t CP, float* V, float* dV, const nadm_adam_t* adam,
                                      const nadm_mlp_weights_t* weights, int32_t flags, int32_t n_slices, float* slab, int32_t* counters, void* stream) {
    AdamFused ad;
    if (adam_fused_args(adam, "nadm_encode_bwd_sliced: Adam state is NULL", &ad)) return 1;
    if (ad.m && (!V || ((uintptr_t)V & 15))) return fail("nadm_encode_bwd_sliced: V must be non-NULL and 16-byte aligned");
    if (CP > 8) return fail("nadm_encode_bwd_sliced: the matrix-core pass only (C <= 8)");
    if (weights && (!weights->hd || !weights->Zn || !weights->H || !weights->dL || !weights->dHpre || !weights->dgp || !weights->small_part))
        return fail("nadm_encode_bwd_sliced: null pointer in the MLP weight-gradient arguments");
    return encode_bwd_impl(xp, ld, idx, b, M, dZ, dzimg, CP, dV, stream, 0u, flags, V, ad, weights, n_slices, slab, counters);
}

extern "C" int nadm_encode_bwd(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                               const float* dZ, const void* dzimg, int32_t CP, float* dV, int32_t flags, void* stream) {
    return encode_bwd_impl(xp, ld, idx, b, M, dZ, dzimg, CP, dV, stream, 0u, flags);
}

extern "C" int nadm_pca_project_t(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                                  const float* Y, const void* yimg, int32_t CP, float* out, void* stream) {
    if (CP > 8) return fail("nadm_pca_project_t: only the matrix-core pass (CP <= 8) has the missing = 1.5 variant");
    return encode_bwd_impl(xp, ld, idx, b, M, Y, yimg, CP, out, stream, 0x3FC0u);
}
