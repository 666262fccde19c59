// Small-tensor kernels: partial reductions, RMSNorm + MLP + softmax forward/backward, Adam, packing,
// synthetic genotype generator.  None of these touches the genotype matrix except pack/unpack/synth;
// they are latency-bound (a few microseconds each) and written for simplicity and determinism
// (fixed reduction orders, no atomics).
#include "nadm_common.h"
#include "../../include/nadm.h"
#include "nadm_host.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <array>
#include <thread>
#include <vector>

namespace nadm {

constexpr int MLP_SB = 4;       // samples per block in mlp_fwd / mlp_bwd_a (2 and 1 were measured: slower)
struct DqChunks {                       // per head: rows of the dQ partial slab to add up (n) and rows the slab occupies (full)
    int64_t n[NADM_MAX_HEADS];
    int64_t full[NADM_MAX_HEADS];
};

// -------------------------------------------------------------------------------------------------
// block-wide sum of `n` (<= 64) per-thread values each; result valid for all threads afterwards.
// -------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* s_red /*[NT/64]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float s = wave_sum_lane63(v);
    __syncthreads();
    if (lane == 63) s_red[wave] = s;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += s_red[w];
    return t;
}

// -------------------------------------------------------------------------------------------------
// Sums of NV per-thread values over the 256 threads of a block through LDS: every thread parks its NV partials ([value][thread],
// row stride 264 floats), thread (value v = t >> 3, part p = t & 7) adds the 32 entries 8 i + p of row v in ascending i, and the
// caller adds the 8 parts of a value in a fixed tree.  NV = 32 here (4 samples x 8 columns).  The DPP form this replaces -- 32
// wave_sum_lane63 chains of six dependent cross-lane adds + a 4-wave combine -- was the longest single phase of the MLP forward:
// 4774 of the 24 k cycles of a block (s_memtime probe, profiles/r03_mlp_probe.txt).  Fixed order, no atomics.
// Two barriers inside; s_part must be free on entry (the caller's barrier).
// -------------------------------------------------------------------------------------------------
constexpr int BR_STRIDE = 264;
template <int NV>
__device__ __forceinline__ void block_reduce_lds(const float (&vals)[NV], float* __restrict__ s_part /*[NV * BR_STRIDE]*/, float* __restrict__ s_parts8 /*[NV * 8]*/) {
    static_assert(NV * 8 == 256, "one (value, part) pair per thread");
    const int tid = threadIdx.x;
#pragma unroll
    for (int v = 0; v < NV; ++v) s_part[v * BR_STRIDE + tid] = vals[v];
    __syncthreads();
    {
        const int v = tid >> 3, p = tid & 7;
        const float* row = s_part + v * BR_STRIDE + p;
        float a[4] = {0.f, 0.f, 0.f, 0.f};                  // four interleaved chains (fixed): i = 4 q + u
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += row[8 * (4 * q + u)];
        s_parts8[v * 8 + p] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    __syncthreads();
}
__device__ __forceinline__ float block_reduce_result(const float* __restrict__ s_parts8, int v) {
    const float* r = s_parts8 + v * 8;
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// =================================================================================================
// mlp_fwd: one block (256 threads) per SB consecutive samples (SB = 4: the partial-slab rows of
// 4 samples are one 128 B line at CP = 8; weights are fetched once per block and reused SB times).
//   Z = sum_chunks zpart ; Zn = Z * rsqrt(mean(Z^2)+1e-8) * g   (torch.nn.RMSNorm, neural_admixture.py:135,173)
//   H = relu(Zn W1^T + b1) (:138-140,174) ; per head: softmax(H Wk^T + bk) (:29,48,176)
// All reductions are fixed-order (bit-reproducible).
// =================================================================================================
template <int SB>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(nadm_heads_t hd, const float* __restrict__ small,
                                                      const float* __restrict__ zpart, int64_t n_chunks, int b,
                                                      float* __restrict__ Z, float* __restrict__ rinv,
                                                      float* __restrict__ Zn, float* __restrict__ H,
                                                      float* __restrict__ Q, uint4* __restrict__ qimg, int64_t qimg_head_u4) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = hd.C, CP = hd.CP, Hd = hd.Hd, SP = hd.SP;
    float* s_grp = sm;                        // [256] float4
    float* s_zn = s_grp + 1024;               // [SB][CP]
    float* s_h = s_zn + SB * CP;              // [SB][Hd]
    float* s_logit = s_h + SB * Hd;           // [SB][SP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * SB;
    const int ns = min(SB, b - i0);
    const int row = SB * CP;                  // floats of the SB samples in one chunk (contiguous)

    // ---- Z = sum over chunks: thread (group g, float4 e4) sums chunks g, g+G, ... (independent 16 B loads,
    //      unrolled so they are in flight together); then a fixed-order combine over the G groups ----
    {
        const int row4 = row / 4;             // float4 per chunk row of the SB samples (<= 32)
        const int G = 256 / row4;
        const int e4 = tid % row4, g = tid / row4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G && e4 * 4 < ns * CP) {
            const float4* src = reinterpret_cast<const float4*>(zpart + (int64_t)i0 * CP) + e4;
            const int64_t stride4 = (int64_t)b * CP / 4;
#pragma unroll 8
            for (int64_t ch = g; ch < n_chunks; ch += G) {
                const float4 v = src[ch * stride4];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        reinterpret_cast<float4*>(s_grp)[tid] = a;
        __syncthreads();
        if (tid < row) {
            float t = 0.f;
            for (int gg = 0; gg < G; ++gg) t += s_grp[(gg * row4 + tid / 4) * 4 + (tid & 3)];
            s_zn[tid] = t;                    // raw Z
        }
        __syncthreads();
    }
    if (tid < ns) {
        float* z = s_zn + tid * CP;
        float ms = 0.f;
        for (int c = 0; c < C; ++c) ms = fmaf(z[c], z[c], ms);
        const float ri = 1.0f / sqrtf(ms / (float)C + 1e-8f);
        const int64_t i = i0 + tid;
        rinv[i] = ri;
        for (int c = 0; c < CP; ++c) {
            const float zz = z[c];
            Z[i * CP + c] = zz;
            const float zn = (c < C) ? zz * ri * small[hd.g_off + c] : 0.f;
            Zn[i * CP + c] = zn;
            z[c] = zn;
        }
    }
    __syncthreads();
    // ---- hidden layer: W1 row fetched once, used for the SB samples ----
    const float* W1 = small + hd.w1_off;
    const float* b1 = small + hd.b1_off;
    for (int h = tid; h < Hd; h += 256) {
        float w[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) w[c] = (c < C) ? W1[h * C + c] : 0.f;
        const float bb = b1[h];
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            float a = bb;
#pragma unroll
            for (int c = 0; c < 32; ++c)
                if (c < C) a = fmaf(s_zn[s * CP + c], w[c], a);
            a = fmaxf(a, 0.f);
            s_h[s * Hd + h] = a;
            if (s < ns) H[(int64_t)(i0 + s) * Hd + h] = a;
        }
    }
    __syncthreads();
    // ---- head logits: each wave takes columns wave, wave+4, ...; a Wk row is read once per SB samples ----
    for (int hh = 0; hh < hd.n_heads; ++hh) {
        const float* Wk = small + hd.wk_off[hh];
        const float* bk = small + hd.bk_off[hh];
        for (int k = wave; k < hd.k[hh]; k += 4) {
            float a[SB];
#pragma unroll
            for (int s = 0; s < SB; ++s) a[s] = 0.f;
#pragma unroll 4
            for (int h = lane; h < Hd; h += 64) {
                const float w = Wk[k * Hd + h];
#pragma unroll
                for (int s = 0; s < SB; ++s) a[s] = fmaf(s_h[s * Hd + h], w, a[s]);
            }
#pragma unroll
            for (int s = 0; s < SB; ++s) {
                const float t = wave_sum_lane63(a[s]);
                if (lane == 63) s_logit[s * SP + hd.qoff[hh] + k] = t + bk[k];
            }
        }
    }
    __syncthreads();
    // ---- softmax per (sample, head) ----
    if (tid < ns * hd.n_heads) {
        const int s = tid / hd.n_heads, hh = tid % hd.n_heads;
        const int k = hd.k[hh], kp = hd.kp[hh], o = hd.qoff[hh];
        float* lg = s_logit + s * SP + o;
        float mx = -INFINITY;
        for (int j = 0; j < k; ++j) mx = fmaxf(mx, lg[j]);
        float sum = 0.f;
        for (int j = 0; j < k; ++j) { const float e = expf(lg[j] - mx); lg[j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < kp; ++j) {
            const float qv = (j < k) ? lg[j] * inv : 0.f;
            Q[(int64_t)(i0 + s) * SP + o + j] = qv;
            q_image_store(qimg, qimg_head_u4, hh, kp, i0 + s, j, qv, b, s == 0);
        }
    }
}

// =================================================================================================
// mlp_bwd_a: one block per SB samples.  dQ = sum_chunks dqpart ; softmax backward ; dH ; relu mask ;
// dZn ; RMSNorm backward -> dZ.   (autograd of neural_admixture.py:173-176)
//   dlogit = Q * (dQ - sum_k dQ*Q) ; dZ = rinv*t - Z*rinv^3*mean_c(t*Z), t = dZn*g ; dg_i = dZn*Z*rinv
// Block 0 also folds the step's loss partials (double) into loss_acc.
// =================================================================================================
template <int SB>
__global__ __launch_bounds__(256) void mlp_bwd_a_kernel(nadm_heads_t hd, const float* __restrict__ small,
                                                        const float* __restrict__ dqpart, DqChunks dq_chunks, int b,
                                                        const float* __restrict__ Z, const float* __restrict__ rinv,
                                                        const float* __restrict__ H, const float* __restrict__ Q,
                                                        float* __restrict__ dL, float* __restrict__ dHpre,
                                                        float* __restrict__ dgp, float* __restrict__ dZ,
                                                        const float* __restrict__ losspart, int64_t n_loss,
                                                        double* __restrict__ loss_acc) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = hd.C, CP = hd.CP, Hd = hd.Hd, SP = hd.SP;
    float* s_grp = sm;                        // [256] float4
    float* s_dl = s_grp + 1024;               // [SB][SP]
    float* s_dzn = s_dl + SB * SP;            // [SB][CP]
    float* s_dh = s_dzn + SB * CP;            // [SB][Hd]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * SB;
    const int ns = min(SB, b - i0);

    // ---- dQ = sum over chunks, per head (slab [chunks_h, b, kp_h]; the SB samples' rows are contiguous):
    //      thread (group g, float4 e4), independent 16 B loads kept in flight by unrolling ----
    {
        int64_t base = 0;
        for (int hh = 0; hh < hd.n_heads; ++hh) {
            const int kp = hd.kp[hh];
            const int row = SB * kp, row4 = row / 4;           // row4 <= 64
            const int64_t nch = dq_chunks.n[hh];
            const int G = 256 / row4;
            const int e4 = tid % row4, g = tid / row4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < G && e4 * 4 < ns * kp) {
                const float4* src = reinterpret_cast<const float4*>(dqpart + base + (int64_t)i0 * kp) + e4;
                const int64_t stride4 = (int64_t)b * kp / 4;
#pragma unroll 8
                for (int64_t ch = g; ch < nch; ch += G) {
                    const float4 v = src[ch * stride4];
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
            }
            reinterpret_cast<float4*>(s_grp)[tid] = a;
            __syncthreads();
            if (tid < row) {
                float t = 0.f;
                for (int gg = 0; gg < G; ++gg) t += s_grp[(gg * row4 + tid / 4) * 4 + (tid & 3)];
                s_dl[(tid / kp) * SP + hd.qoff[hh] + (tid % kp)] = t;      // raw dQ
            }
            __syncthreads();
            base += dq_chunks.full[hh] * b * kp;
        }
    }
    // ---- softmax backward per (sample, head) ----
    if (tid < ns * hd.n_heads) {
        const int s = tid / hd.n_heads, hh = tid % hd.n_heads;
        const int k = hd.k[hh], kp = hd.kp[hh], o = hd.qoff[hh];
        const float* q = Q + (int64_t)(i0 + s) * SP + o;
        float* dl = s_dl + s * SP + o;
        float dot = 0.f;
        for (int j = 0; j < k; ++j) dot = fmaf(dl[j], q[j], dot);
        for (int j = 0; j < kp; ++j) {
            const float v = (j < k) ? q[j] * (dl[j] - dot) : 0.f;
            dl[j] = v;
            dL[(int64_t)(i0 + s) * SP + o + j] = v;
        }
    }
    __syncthreads();
    // ---- dH with relu mask: a Wk element is read once per SB samples ----
    for (int h = tid; h < Hd; h += 256) {
        float a[SB];
#pragma unroll
        for (int s = 0; s < SB; ++s) a[s] = 0.f;
        for (int hh = 0; hh < hd.n_heads; ++hh) {
            const float* Wk = small + hd.wk_off[hh];
            const int o = hd.qoff[hh];
            for (int k = 0; k < hd.k[hh]; ++k) {
                const float w = Wk[k * Hd + h];
#pragma unroll
                for (int s = 0; s < SB; ++s) a[s] = fmaf(s_dl[s * SP + o + k], w, a[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            float v = 0.f;
            if (s < ns) {
                v = (H[(int64_t)(i0 + s) * Hd + h] > 0.f) ? a[s] : 0.f;
                dHpre[(int64_t)(i0 + s) * Hd + h] = v;
            }
            s_dh[s * Hd + h] = v;
        }
    }
    __syncthreads();
    // ---- dZn[s][c] = sum_h dHpre[s][h] W1[h][c]: wave w takes pairs w, w+4, ... ----
    {
        const float* W1 = small + hd.w1_off;
        for (int pr = wave; pr < SB * C; pr += 4) {
            const int s = pr / C, c = pr % C;
            float a = 0.f;
#pragma unroll 4
            for (int h = lane; h < Hd; h += 64) a = fmaf(s_dh[s * Hd + h], W1[h * C + c], a);
            a = wave_sum_lane63(a);
            if (lane == 63) s_dzn[s * CP + c] = a;
        }
    }
    __syncthreads();
    if (tid < ns) {
        const int64_t i = i0 + tid;
        const float ri = rinv[i];
        const float* g = small + hd.g_off;
        const float* dzn = s_dzn + tid * CP;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(dzn[c] * g[c], Z[i * CP + c], dot);
        const float mean_tz = dot / (float)C;
        const float ri3 = ri * ri * ri;
        for (int c = 0; c < CP; ++c) {
            float dz = 0.f, dgv = 0.f;
            if (c < C) {
                const float z = Z[i * CP + c];
                dz = ri * (dzn[c] * g[c]) - z * ri3 * mean_tz;
                dgv = dzn[c] * z * ri;
            }
            dZ[i * CP + c] = dz;
            dgp[i * CP + c] = dgv;
        }
    }
    // ---- loss (block 0 only) ----
    if (blockIdx.x == 0 && n_loss > 0) {
        double a = 0.0;
        for (int64_t e = tid; e < n_loss; e += 256) a += (double)losspart[e];
        a = wave_sum_all_f64(a);
        __shared__ double s_l[4];
        __syncthreads();
        if ((tid & 63) == 0) s_l[tid >> 6] = a;
        __syncthreads();
        if (tid == 0) {
            const double tot = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
            loss_acc[0] += tot;          // running (epoch) sum
            loss_acc[1] = tot;           // last step
        }
    }
}

// =================================================================================================
// Fast variants for Hd <= 256*MLP_JMAX, C <= 8 (the defaults: Hd = 1024, C = 8).  Same math and the same
// fixed reduction orders per output, but organised around latency: the kernels above walk the weight
// matrices in short dependent batches of loads (a dozen L2 round trips each); here thread t owns hidden
// units t, t+256, ... -- its W1 rows, Wk columns and activations stay in registers, every weight load of
// a phase is issued before the first use, and sums over the hidden dimension are thread-partials ->
// DPP wave sums -> 4-wave combine.
// =================================================================================================
constexpr int MLP_JMAX = 8;     // hidden units per thread: JH = 4 (Hd <= 1024) or 8 (Hd <= 2048)
constexpr int MLP_KT = 8;       // head columns per pass of the hidden-dimension reductions

template <int SB, int JH, bool C8>
__global__ __launch_bounds__(256) void mlp_fwd_fast_kernel(nadm_heads_t hd, const float* __restrict__ small,
                                                           const float* __restrict__ zpart, int64_t n_chunks, int b,
                                                           float* __restrict__ Z, float* __restrict__ rinv,
                                                           float* __restrict__ Zn, float* __restrict__ H,
                                                           float* __restrict__ Q, uint4* __restrict__ qimg, int64_t qimg_head_u4) {
    __shared__ __attribute__((aligned(16))) float s_grp[1024];
    __shared__ float s_zn[SB * 8];
    static_assert(SB * MLP_KT == 32, "block_reduce_lds<32>");
    __shared__ float s_part[32 * BR_STRIDE];
    __shared__ float s_parts8[32 * 8];
    extern __shared__ __attribute__((aligned(16))) float s_logit[];      // [SB][SP]
    const int C = C8 ? 8 : hd.C, CP = C8 ? 8 : hd.CP, Hd = hd.Hd, SP = hd.SP;     // (C8: C == CP == 8 at compile time -- no run-time divisions by the row length)
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * SB;
    const int ns = min(SB, b - i0);
    const int row = SB * CP;

    // ---- head biases into LDS now (one parallel fetch; read per head pass by a few threads below) ----
    float* const s_bk = s_logit + SB * SP;                           // [SP], behind s_logit in the dynamic allocation
    for (int hh = 0; hh < hd.n_heads; ++hh)
        for (int k = tid; k < hd.k[hh]; k += 256) s_bk[hd.qoff[hh] + k] = small[hd.bk_off[hh] + k];
    // ---- weights of this thread's hidden units: issued first, consumed after the Z reduction ----
    const float* W1 = small + hd.w1_off;
    const float* b1 = small + hd.b1_off;
    float w1[JH][8], bb[JH];
    float gw[8];                                          // RMSNorm weight: loaded here with everything else, not one dependent load
#pragma unroll                                            // per component inside the single-thread normalisation below
    for (int c = 0; c < 8; ++c) gw[c] = (c < C) ? small[hd.g_off + c] : 0.f;
    // W1 is [Hd, C] row-major: with C == 8 a thread's row is 32 contiguous bytes = two 16-byte loads (8 load instructions per
    // thread); read element by element it is 32 instructions whose 64 lanes each touch a different 32-byte sector -- those
    // loads alone kept the texture path busy for ~8 us per block (s_memtime probe, warm caches make no difference)
    // Every load of a phase is UNCONDITIONAL with a clamped index, in straight-line code, and masked afterwards: a load under a
    // lane-dependent condition -- or inside a run-time branch -- becomes its own block with a wait behind it, and the 24-56
    // loads of a phase then pay one L2 round trip EACH instead of one together (s_memtime probe: 5.5 us for the H + W1 loads
    // of mlp_bwd_a).  C8 (C == 8, 16-byte aligned W1: the default model) is a template parameter for the same reason: W1 is
    // [Hd, C] row-major, a thread's row is then two 16-byte loads instead of 8 loads whose lanes each touch another sector.
#pragma unroll
    for (int j = 0; j < JH; ++j) {
        const int h = tid + 256 * j, hc = h < Hd ? h : Hd - 1;
        bb[j] = b1[hc];
        if constexpr (C8) {
            const float4 lo4 = reinterpret_cast<const float4*>(W1)[hc * 2];
            const float4 hi4 = reinterpret_cast<const float4*>(W1)[hc * 2 + 1];
            w1[j][0] = lo4.x; w1[j][1] = lo4.y; w1[j][2] = lo4.z; w1[j][3] = lo4.w;
            w1[j][4] = hi4.x; w1[j][5] = hi4.y; w1[j][6] = hi4.z; w1[j][7] = hi4.w;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) w1[j][c] = W1[hc * C + (c < C ? c : 0)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t cmask[8];                                               // column c < C (bit masks instead of selects: nadm_common.h lt_mask)
#pragma unroll
    for (int c = 0; c < 8; ++c) cmask[c] = lt_mask(c, C);
#pragma unroll
    for (int j = 0; j < JH; ++j) {                                   // masks, after every load has been issued
        const uint32_t in = lt_mask(tid + 256 * j, Hd);
        bb[j] = keepf(bb[j], in);
#pragma unroll
        for (int c = 0; c < 8; ++c) w1[j][c] = keepf(w1[j][c], in & cmask[c]);
    }
    // ---- Z = sum over chunks (same scheme as mlp_fwd_kernel) ----
    {
        const int row4 = row / 4;
        const int G = 256 / row4;
        const int e4 = tid % row4, g = tid / row4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G && e4 * 4 < ns * CP) {
            const float4* src = reinterpret_cast<const float4*>(zpart + (int64_t)i0 * CP) + e4;
            const int64_t stride4 = (int64_t)b * CP / 4;
            // eight rows per trip, loaded unconditionally (clamped row) and added under a mask: with the trip count unknown the
            // unrolled loop tested and waited per row -- 4 of the kernel's 14 us for 8 loads
            for (int64_t ch0 = g; ch0 < n_chunks; ch0 += 8 * G) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int64_t ch = ch0 + (int64_t)u * G; v[u] = src[(ch < n_chunks ? ch : n_chunks - 1) * stride4]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t ok = lt_mask64(ch0 + (int64_t)u * G, n_chunks);
                    a.x += keepf(v[u].x, ok); a.y += keepf(v[u].y, ok); a.z += keepf(v[u].z, ok); a.w += keepf(v[u].w, ok);
                }
            }
        }
        reinterpret_cast<float4*>(s_grp)[tid] = a;
        __syncthreads();
        if (tid < row) {
            float t = 0.f;
            for (int gg = 0; gg < G; ++gg) t += s_grp[(gg * row4 + tid / 4) * 4 + (tid & 3)];
            s_zn[tid] = t;
        }
        __syncthreads();
    }
    if (tid < ns) {
        float* z = s_zn + tid * CP;
        float ms = 0.f;
        for (int c = 0; c < C; ++c) ms = fmaf(z[c], z[c], ms);
        const float ri = 1.0f / sqrtf(ms / (float)C + 1e-8f);
        const int64_t i = i0 + tid;
        rinv[i] = ri;
#pragma unroll
        for (int c = 0; c < 8; ++c) {                     // this kernel: C <= CP <= 8
            if (c < CP) {
                const float zz = z[c];
                Z[i * CP + c] = zz;
                const float zn = (c < C) ? zz * ri * gw[c] : 0.f;
                Zn[i * CP + c] = zn;
                z[c] = zn;
            }
        }
    }
    __syncthreads();
    // ---- hidden layer, kept in registers ----
    float hv[JH][SB];
#pragma unroll
    for (int j = 0; j < JH; ++j) {
        const int h = tid + 256 * j;
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            float a = bb[j];
#pragma unroll
            for (int c = 0; c < 8; ++c) a = fmaf(s_zn[s * CP + (c < CP ? c : 0)], w1[j][c], a);     // w1 = 0 beyond C
            a = fmaxf(a, 0.f);
            hv[j][s] = keepf(a, lt_mask(h, Hd));
            if (h < Hd && s < ns) H[(int64_t)(i0 + s) * Hd + h] = a;
        }
    }
    // ---- head logits, MLP_KT columns per pass ----
    for (int hh = 0; hh < hd.n_heads; ++hh) {
        const float* Wk = small + hd.wk_off[hh];
        const float* bk = s_bk + hd.qoff[hh];
        const int K = hd.k[hh];
        for (int k0 = 0; k0 < K; k0 += MLP_KT) {
            float wk[JH][MLP_KT];
#pragma unroll
            for (int kk = 0; kk < MLP_KT; ++kk) {                              // unconditional, all in flight together (see the W1 loads)
                const float* row = Wk + (k0 + kk < K ? k0 + kk : K - 1) * Hd;   // wave-uniform row, clamped
#pragma unroll
                for (int j = 0; j < JH; ++j) { const int h = tid + 256 * j; wk[j][kk] = row[h < Hd ? h : Hd - 1]; }
            }
            __builtin_amdgcn_sched_barrier(0);                                 // keep the loads above their first use
#pragma unroll
            for (int kk = 0; kk < MLP_KT; ++kk) {
                const uint32_t kin = lt_mask(k0 + kk, K);
#pragma unroll
                for (int j = 0; j < JH; ++j) wk[j][kk] = keepf(wk[j][kk], kin & lt_mask(tid + 256 * j, Hd));
            }
            float acc[SB][MLP_KT];
#pragma unroll
            for (int s = 0; s < SB; ++s)
#pragma unroll
                for (int kk = 0; kk < MLP_KT; ++kk) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < JH; ++j) a = fmaf(hv[j][s], wk[j][kk], a);
                    acc[s][kk] = a;
                }
            __syncthreads();                                  // s_part / s_parts8 free (previous pass consumed)
            block_reduce_lds<32>(reinterpret_cast<const float (&)[32]>(acc), s_part, s_parts8);      // value v = s * MLP_KT + kk
            if (tid < SB * MLP_KT) {
                const int s = tid / MLP_KT, kk = tid % MLP_KT;
                if (k0 + kk < K) s_logit[s * SP + hd.qoff[hh] + k0 + kk] = block_reduce_result(s_parts8, tid) + bk[k0 + kk];
            }
        }
    }
    __syncthreads();
    // ---- softmax: one thread per (sample, column).  Phase A: the thread's own exponential (every thread of a head finds the head's
    // maximum itself) parked in LDS; phase B: every thread of a head adds the head's k exponentials in order, so all of them arrive
    // at the same sum bit for bit.  (r02 had every thread recompute all k exponentials of its head: k - 1 expf too many per
    // thread in the kernel's last dependent phase.) ----
    float* const s_ex = s_part;                                      // [SB][SP], the reduction scratch is free now
    for (int e = tid; e < ns * SP; e += 256) {
        const int s = e / SP, c = e % SP;
        int hh = 0;
        while (hh + 1 < hd.n_heads && c >= hd.qoff[hh + 1]) ++hh;
        const int k = hd.k[hh], o = hd.qoff[hh], j = c - o;
        const float* lg = s_logit + s * SP + o;
        float mx = -INFINITY;
        for (int jj = 0; jj < k; ++jj) mx = fmaxf(mx, lg[jj]);
        s_ex[e] = (j < k) ? expf(lg[j] - mx) : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < ns * SP; e += 256) {
        const int s = e / SP, c = e % SP;
        int hh = 0;
        while (hh + 1 < hd.n_heads && c >= hd.qoff[hh + 1]) ++hh;
        const int k = hd.k[hh], o = hd.qoff[hh], j = c - o;
        const float* ex = s_ex + s * SP + o;
        float sum = 0.f;
        for (int jj = 0; jj < k; ++jj) sum += ex[jj];
        const float inv = 1.0f / sum;
        const float qv = (j < k) ? ex[j] * inv : 0.f;
        Q[(int64_t)(i0 + s) * SP + c] = qv;
        q_image_store(qimg, qimg_head_u4, hh, hd.kp[hh], i0 + s, j, qv, b, s == 0);
    }
}

template <int SB, int JH, bool C8>
__global__ __launch_bounds__(256) void mlp_bwd_a_fast_kernel(nadm_heads_t hd, const float* __restrict__ small,
                                                             const float* __restrict__ dqpart, DqChunks dq_chunks, int b,
                                                             const float* __restrict__ Z, const float* __restrict__ rinv,
                                                             const float* __restrict__ H, const float* __restrict__ Q,
                                                             float* __restrict__ dL, float* __restrict__ dHpre,
                                                             float* __restrict__ dgp, float* dZ,
                                                             const float* __restrict__ losspart, int64_t n_loss,
                                                             double* __restrict__ loss_acc, uint4* dzimg, int* dzcnt) {
    __shared__ __attribute__((aligned(16))) float s_grp[1024];
    __shared__ float s_dzn[SB * 8];
    static_assert(SB * 8 == 32, "block_reduce_lds<32>");
    __shared__ float s_part[32 * BR_STRIDE];
    __shared__ float s_parts8[32 * 8];
    extern __shared__ __attribute__((aligned(16))) float s_dl[];         // [SB][SP]
    const int C = C8 ? 8 : hd.C, CP = C8 ? 8 : hd.CP, Hd = hd.Hd, SP = hd.SP;     // (C8: C == CP == 8 at compile time)
    const int tid = threadIdx.x;
    // ---- loss: one EXTRA block (the launch has one more block than sample groups when n_loss > 0).  Inside a working block
    // the sum -- an HBM round trip, a float64 wave reduction -- was a tail every other block had already left behind.
    if (n_loss > 0 && blockIdx.x == gridDim.x - 1) {
        double a = 0.0;
        for (int64_t e0 = 0; e0 < n_loss; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int64_t e = e0 + tid + 256 * u; v[u] = e < n_loss ? losspart[e] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)v[u];
        }
        a = wave_sum_all_f64(a);
        __shared__ double s_l[4];
        if ((tid & 63) == 0) s_l[tid >> 6] = a;
        __syncthreads();
        if (tid == 0) {
            const double tot = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
            loss_acc[0] += tot;
            loss_acc[1] = tot;
        }
        return;
    }
    const int i0 = blockIdx.x * SB;
    const int ns = min(SB, b - i0);

    // ---- everything the short single-thread sections below read from global memory is fetched HERE, by many threads at once,
    // and parked in LDS: Q of the block's samples (softmax backward), Z, rinv and the RMSNorm weight (RMSNorm backward).
    // Read where they are used -- a handful of threads looping over k or c -- every element is a dependent L2 round trip of
    // its own while the whole block waits at the next barrier (8 + 8 + 8 round trips = most of the kernel's 33 us).
    __shared__ float s_z[SB * 8], s_g[8], s_ri[SB];
    float* const s_q = s_dl + SB * SP;                               // [SB][SP], behind s_dl in the dynamic allocation
    for (int e = tid; e < SB * SP; e += 256) s_q[e] = (e / SP < ns) ? Q[(int64_t)(i0 + e / SP) * SP + e % SP] : 0.f;
    if (tid < SB * 8) s_z[tid] = (tid / 8 < ns && (tid & 7) < CP) ? Z[(int64_t)(i0 + tid / 8) * CP + (tid & 7)] : 0.f;
    if (tid >= 64 && tid < 72) s_g[tid - 64] = (tid - 64 < C) ? small[hd.g_off + tid - 64] : 0.f;
    if (tid >= 128 && tid < 128 + SB) s_ri[tid - 128] = (tid - 128 < ns) ? rinv[i0 + tid - 128] : 0.f;
    // ---- this thread's hidden units: forward activations (relu mask) and W1 rows, issued first ----
    const float* W1 = small + hd.w1_off;
    float hact[JH][SB], w1[JH][8];
#pragma unroll
    for (int j = 0; j < JH; ++j) {                                   // unconditional, straight-line: see mlp_fwd_fast_kernel
        const int h = tid + 256 * j, hc = h < Hd ? h : Hd - 1;
#pragma unroll
        for (int s = 0; s < SB; ++s) hact[j][s] = H[(int64_t)(i0 + (s < ns ? s : ns - 1)) * Hd + hc];
        if constexpr (C8) {
            const float4 lo4 = reinterpret_cast<const float4*>(W1)[hc * 2];
            const float4 hi4 = reinterpret_cast<const float4*>(W1)[hc * 2 + 1];
            w1[j][0] = lo4.x; w1[j][1] = lo4.y; w1[j][2] = lo4.z; w1[j][3] = lo4.w;
            w1[j][4] = hi4.x; w1[j][5] = hi4.y; w1[j][6] = hi4.z; w1[j][7] = hi4.w;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) w1[j][c] = W1[hc * C + (c < C ? c : 0)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t cmask[8], smask[SB];                                    // column c < C, sample s < ns (bit masks instead of selects: lt_mask)
#pragma unroll
    for (int c = 0; c < 8; ++c) cmask[c] = lt_mask(c, C);
#pragma unroll
    for (int s = 0; s < SB; ++s) smask[s] = lt_mask(s, ns);
#pragma unroll
    for (int j = 0; j < JH; ++j) {
        const uint32_t in = lt_mask(tid + 256 * j, Hd);
#pragma unroll
        for (int s = 0; s < SB; ++s) hact[j][s] = keepf(hact[j][s], in & smask[s]);
#pragma unroll
        for (int c = 0; c < 8; ++c) w1[j][c] = keepf(w1[j][c], in & cmask[c]);
    }
    // ---- dQ = sum over chunks, per head ----
    {
        int64_t base = 0;
        for (int hh = 0; hh < hd.n_heads; ++hh) {
            const int kp = hd.kp[hh];
            const int row = SB * kp, row4 = row / 4;
            const int64_t nch = dq_chunks.n[hh];
            const int G = 256 / row4;
            const int e4 = tid % row4, g = tid / row4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < G && e4 * 4 < ns * kp) {
                const float4* src = reinterpret_cast<const float4*>(dqpart + base + (int64_t)i0 * kp) + e4;
                const int64_t stride4 = (int64_t)b * kp / 4;
                constexpr int DQD = 16;     // rows per trip: loaded unconditionally (clamped row), added under a mask -- see mlp_fwd_fast_kernel
                for (int64_t ch0 = g; ch0 < nch; ch0 += (int64_t)DQD * G) {
                    float4 v[DQD];
#pragma unroll
                    for (int u = 0; u < DQD; ++u) { const int64_t ch = ch0 + (int64_t)u * G; v[u] = src[(ch < nch ? ch : nch - 1) * stride4]; }
                    __builtin_amdgcn_sched_barrier(0);        // every load of the trip in flight before the first addition (left to itself the scheduler may interleave them six at a time: +3 us)
#pragma unroll
                    for (int u = 0; u < DQD; ++u) {
                        const uint32_t ok = lt_mask64(ch0 + (int64_t)u * G, nch);
                        a.x += keepf(v[u].x, ok); a.y += keepf(v[u].y, ok); a.z += keepf(v[u].z, ok); a.w += keepf(v[u].w, ok);
                    }
                }
            }
            reinterpret_cast<float4*>(s_grp)[tid] = a;
            __syncthreads();
            if (tid < row) {
                float t = 0.f;
                for (int gg = 0; gg < G; ++gg) t += s_grp[(gg * row4 + tid / 4) * 4 + (tid & 3)];
                s_dl[(tid / kp) * SP + hd.qoff[hh] + (tid % kp)] = t;
            }
            __syncthreads();
            base += dq_chunks.full[hh] * b * kp;
        }
    }
    // ---- softmax backward: one thread per (sample, column); every thread of a head forms the head's dot product itself, in
    // order (the same value in all of them) and writes its element to a third LDS image (no in-place update, one barrier) ----
    float* const s_dlv = s_dl + 2 * SB * SP;                          // [SB][SP]: dL of the block's samples, read by the dH phase
    for (int e = tid; e < SB * SP; e += 256) {
        const int s = e / SP, c = e % SP;
        int hh = 0;
        while (hh + 1 < hd.n_heads && c >= hd.qoff[hh + 1]) ++hh;
        const int k = hd.k[hh], o = hd.qoff[hh], j = c - o;
        float v = 0.f;
        if (s < ns && j < k) {
            const float* dl = s_dl + s * SP + o;
            const float* q = s_q + s * SP + o;
            float dot = 0.f;
            for (int jj = 0; jj < k; ++jj) dot = fmaf(dl[jj], q[jj], dot);
            v = q[j] * (dl[j] - dot);
        }
        s_dlv[e] = v;
        if (s < ns) dL[(int64_t)(i0 + s) * SP + c] = v;
    }
    __syncthreads();
    // ---- dH (relu-masked) in registers: head columns in passes of MLP_KT, accumulation order = head, column ----
    float dh[JH][SB];
#pragma unroll
    for (int j = 0; j < JH; ++j)
#pragma unroll
        for (int s = 0; s < SB; ++s) dh[j][s] = 0.f;
    for (int hh = 0; hh < hd.n_heads; ++hh) {
        const float* Wk = small + hd.wk_off[hh];
        const int K = hd.k[hh], o = hd.qoff[hh];
        for (int k0 = 0; k0 < K; k0 += MLP_KT) {
            float wk[JH][MLP_KT];
#pragma unroll
            for (int kk = 0; kk < MLP_KT; ++kk) {                              // unconditional, all in flight together (see the W1 loads)
                const float* row = Wk + (k0 + kk < K ? k0 + kk : K - 1) * Hd;   // wave-uniform row, clamped
#pragma unroll
                for (int j = 0; j < JH; ++j) { const int h = tid + 256 * j; wk[j][kk] = row[h < Hd ? h : Hd - 1]; }
            }
            __builtin_amdgcn_sched_barrier(0);                                 // keep the loads above their first use
#pragma unroll
            for (int kk = 0; kk < MLP_KT; ++kk) {
                const uint32_t kin = lt_mask(k0 + kk, K);
#pragma unroll
                for (int j = 0; j < JH; ++j) wk[j][kk] = keepf(wk[j][kk], kin & lt_mask(tid + 256 * j, Hd));
            }
#pragma unroll
            for (int kk = 0; kk < MLP_KT; ++kk) {
                float dls[SB];
#pragma unroll
                for (int s = 0; s < SB; ++s) dls[s] = (k0 + kk < K) ? s_dlv[s * SP + o + k0 + kk] : 0.f;
#pragma unroll
                for (int j = 0; j < JH; ++j)
#pragma unroll
                    for (int s = 0; s < SB; ++s) dh[j][s] = fmaf(dls[s], wk[j][kk], dh[j][s]);
            }
        }
    }
    float part[SB][8];
#pragma unroll
    for (int s = 0; s < SB; ++s)
#pragma unroll
        for (int c = 0; c < 8; ++c) part[s][c] = 0.f;
#pragma unroll
    for (int j = 0; j < JH; ++j) {
        const int h = tid + 256 * j;
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            // relu mask: hact is a relu output (>= +0), so "> 0" is "bits != 0": -bits is negative exactly then
            uint32_t pos = (uint32_t)((int)(0u - __float_as_uint(hact[j][s])) >> 31);
            asm("" : "+v"(pos));
            const float v = keepf(dh[j][s], pos);
            if (h < Hd && s < ns) dHpre[(int64_t)(i0 + s) * Hd + h] = v;
#pragma unroll
            for (int c = 0; c < 8; ++c) part[s][c] = fmaf(v, w1[j][c], part[s][c]);
        }
    }
    // ---- dZn[s][c] = sum_h dHpre[s][h] W1[h][c] ----
    block_reduce_lds<32>(reinterpret_cast<const float (&)[32]>(part), s_part, s_parts8);                      // value v = s * 8 + c
    if (tid < SB * 8) s_dzn[tid] = block_reduce_result(s_parts8, tid);
    __syncthreads();
    if (tid < ns) {
        const int64_t i = i0 + tid;
        const float ri = s_ri[tid];
        const float* g = s_g;
        const float* zr = s_z + tid * 8;
        const float* dzn = s_dzn + tid * 8;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(dzn[c] * g[c], zr[c], dot);
        const float mean_tz = dot / (float)C;
        const float ri3 = ri * ri * ri;
        for (int c = 0; c < CP; ++c) {
            float dz = 0.f, dgv = 0.f;
            if (c < C) {
                const float z = zr[c];
                dz = ri * (dzn[c] * g[c]) - z * ri3 * mean_tz;
                dgv = dzn[c] * z * ri;
            }
            // (image path: written THROUGH to memory -- the group's last block, possibly on another XCD with its own L2, reads it back)
            if (dzimg != nullptr) __hip_atomic_store(dZ + i * CP + c, dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dZ[i * CP + c] = dz;
            dgp[i * CP + c] = dgv;
        }
    }
    // ---- dZ as the operand image of pass 3 (nadm_common.h, dzi_build_piece): a lane of the matrix instruction holds 32 consecutive
    // samples of one column = the rows of 32 / SB blocks of this launch.  The block that finishes a group of 32 samples LAST builds
    // the group's image (one wave: 8 columns x 8 pieces); the counters return to zero for the next launch.  (As a launch of its own
    // the image cost 5 us + a launch gap per step, more than the matrix instruction saved in pass 3.)
    // Ordering without a fence: the dZ stores above go through to memory (a __threadfence here is a write-back of the XCD's whole L2,
    // which this kernel has just filled with dHpre: 200 of them took the kernel from 17 to 36-42 us), the wave waits for their
    // acknowledgement, then the block is counted with a device-scope atomic; the last block reads with device-scope loads.
    if (dzimg != nullptr) {
        __shared__ int s_last;
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): this wave's dZ stores have been acknowledged ...
        __syncthreads();
        const int grp = i0 / 32;
        if (tid == 0) {
            const int in_group = (min(b, 32 * grp + 32) - 32 * grp + SB - 1) / SB;
            const int old = __hip_atomic_fetch_add(&dzcnt[grp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before the block is counted
            s_last = old == in_group - 1;
            if (s_last) __hip_atomic_store(&dzcnt[grp], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last) {                                                 // block-uniform
            // the group's 32 x 8 values: ONE load per thread, parked in LDS; 64 threads then cut a column's 32 values into a piece each
            float* const s_grpz = s_part;                             // (free since the block reduction)
            const int smp = 32 * grp + (tid >> 3), cc = tid & 7;
            float x = 0.f;
            if (smp < b && cc < CP) x = __hip_atomic_load(dZ + (int64_t)smp * CP + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_grpz[tid] = x;
            __syncthreads();
            if (tid < 64) dzi_build_piece<true>(s_grpz, b, CP, dzimg, grp >> 2, grp & 3, tid >> 3, tid & 7);
        }
    }
}

// =================================================================================================
// loglik: the reference's post-training report (src/utils_c/utils.pyx:15-40) from the packed matrix,
//   sum over non-missing (i,j) of g*log(rec) + (2-g)*log1p(-rec),  rec = clip(sum_k Q[i,k] P[j,k], eps, 1-eps),
//   g = clip(code, eps, 2-eps), everything in float64 like the Cython loop (P and Q are the float32 results
//   widened, train.py:137-139).  block = 256 threads = 256 byte columns = 1024 SNPs; thread t keeps the K
//   frequencies of its 4 SNPs in registers (doubles), rows stream through with Q tiles broadcast from LDS.
//   Per-thread double accumulators, fixed-order block reduction, one partial per block (summed by the caller).
//   blockIdx.y = one of LOGLIK_ROW_SLICES slices of the rows (whole Q tiles): M / 1024 blocks alone are two blocks per CU at
//   500k SNPs, two waves per SIMD for a loop whose every genotype waits for two software float64 logarithms (r05).
// =================================================================================================
constexpr int LOGLIK_ROW_SLICES = 8;
template <int KP>
__global__ __launch_bounds__(256) void loglik_kernel(const uint8_t* __restrict__ xp, int64_t ld, int64_t rows, int64_t M,
                                                     const float* __restrict__ P, const float* __restrict__ Q, int K, int qstride,
                                                     double eps, double* __restrict__ partial) {
    constexpr int RT = 64;                                   // rows per Q tile
    __shared__ double s_q[RT * KP];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    const int64_t col = (int64_t)blockIdx.x * 256 + tid;     // byte column
    const bool col_ok = col * 4 < M;
    double p[4][KP];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t m = col * 4 + j;
#pragma unroll
        for (int k = 0; k < KP; ++k) p[j][k] = (m < M && k < K) ? (double)P[m * K + k] : 0.0;
    }
    // The sum of logarithms as the logarithm of products (r05).  With c the code and s = 1 - rec, the term of a genotype is
    //   c = 0: eps*log r + (2 - eps)*log s      c = 1: log r + log s      c = 2: (2 - eps)*log r + eps*log s
    //   = log f + eps * (log bn - log bd),   f = s*s | r*s | r*r,   bn / bd = r / s | 1 | s / r.
    // f, bn and bd are multiplied up over 4 rows x 4 SNPs (every factor lies in [eps, 1], eps = 1e-6: the products stay above 1e-192)
    // and ONE triple of float64 logarithms is taken per 16 genotypes instead of 32 -- the software logarithms were nine tenths of
    // this kernel's arithmetic.  Same float64 quantities as the Cython loop up to the rounding of the products (1e-15 relative);
    // log s stands in for log1p(-r): s = 1 - r is exact to 1e-16 absolute and >= eps.
    double acc = 0.0, pf = 1.0, pn = 1.0, pd = 1.0;
    const int64_t tiles = (rows + RT - 1) / RT, tps = (tiles + gridDim.y - 1) / gridDim.y;
    const int64_t row_lo = (int64_t)blockIdx.y * tps * RT;
    const int64_t row_hi = row_lo + tps * RT < rows ? row_lo + tps * RT : rows;
    for (int64_t r0 = row_lo; r0 < row_hi; r0 += RT) {
        const int nr = (int)(row_hi - r0 < RT ? row_hi - r0 : RT);
        __syncthreads();
        for (int e = tid; e < nr * KP; e += 256) {
            const int r = e / KP, k = e % KP;
            s_q[e] = k < K ? (double)Q[(r0 + r) * qstride + k] : 0.0;
        }
        __syncthreads();
        if (col_ok) {
            for (int r = 0; r < nr; ++r) {
                const uint32_t byte = xp[(r0 + r) * ld + col];
                double rec[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const double q = s_q[r * KP + k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) rec[j] = fma(q, p[j][k], rec[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c = (byte >> (2 * j)) & 3u;
                    if (c != 3u && col * 4 + j < M) {
                        const double rr = fmax(eps, fmin(rec[j], 1.0 - eps));
                        const double om = 1.0 - rr;
                        pf *= (c >= 1u ? rr : om) * (c == 2u ? rr : om);
                        pn *= c == 0u ? rr : (c == 2u ? om : 1.0);
                        pd *= c == 0u ? om : (c == 2u ? rr : 1.0);
                    }
                }
                if ((r & 3) == 3 || r == nr - 1) {
                    acc += log(pf) + eps * (log(pn) - log(pd));
                    pf = pn = pd = 1.0;
                }
            }
        }
    }
    acc = wave_sum_all_f64(acc);
    if ((tid & 63) == 0) s_red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// =================================================================================================
// bed_to_packed on the device: PLINK .bed is SNP-major, 4 samples per byte; the engine wants sample-major, 4 SNPs per
// byte -- a transpose of a 2-bit matrix plus the reference's recode [2,3,1,0] (src/utils_c/utils.pyx:52).
//   A 32-bit word made of the bytes of 4 consecutive SNPs at one byte column holds a 4 x 4 block (SNP r, sample i) of
//   2-bit fields at bit 8r + 2i; two delta swaps transpose it to (sample i, SNP r) at bit 8i + 2r = one output byte per
//   sample.  The recode is bitwise: g_hi = ~c_hi, g_lo = c_lo ^ c_hi.
//   block = 256 threads, tile = 512 SNPs (128 output byte columns) x 256 samples (64 input byte columns); the tile goes
//   through LDS so that each sample row is written as 128 contiguous bytes (bed_to_packed_kernel below).
//   Code counts (for the reader's "flip if the mean code is >= 1" rule) are integer atomics: order-independent.
// =================================================================================================
__device__ __forceinline__ uint32_t transpose4x4_2bit(uint32_t w) {
    uint32_t t = ((w >> 6) ^ w) & 0x00CC00CCu;
    w ^= t ^ (t << 6);
    t = ((w >> 12) ^ w) & 0x0000F0F0u;
    return w ^ t ^ (t << 12);
}

// 4 x 4 byte transpose: out[c] = byte c of in[0..3] (in[r] = one dword of row r)
__device__ __forceinline__ void transpose4x4_bytes(const uint32_t (&in)[4], uint32_t (&out)[4]) {
    const uint32_t t01l = __builtin_amdgcn_perm(in[1], in[0], 0x05010400u), t01h = __builtin_amdgcn_perm(in[1], in[0], 0x07030602u);
    const uint32_t t23l = __builtin_amdgcn_perm(in[3], in[2], 0x05010400u), t23h = __builtin_amdgcn_perm(in[3], in[2], 0x07030602u);
    out[0] = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);
    out[1] = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
}

// r05: tile = 512 SNPs x 256 samples (64 bytes of every SNP row in -- the tiles next to it in the grid read the rest of the line --
// 128 bytes of every sample row out); a thread task = 16 SNP rows x one dword of sample bytes (16 samples): 16 dword loads whose
// lanes run along the SNP row, a 16 x 16 transpose of 2-bit fields in registers (byte transposes by v_perm around the word-level
// delta swaps), 16 dword stores into the LDS tile.  100k x 500k (25 GB moved): the first version, single bytes read and single
// bytes stored to LDS, 14.5 ms (1.7 TB/s: bound by its instruction count); this one 6.9 ms (3.6 TB/s) with 256-sample tiles, 8.4
// with 512 (68 KB of LDS: two blocks per CU), 13.0 with 128 (32-byte reads).  The flip pass, when the data ask for it, 4.5 ms.
constexpr int BT_SNPS = 512, BT_JD = 16, BT_PITCH = 33;       // SNPs per tile, dword columns of sample bytes per tile, LDS row pitch (dwords)
__global__ __launch_bounds__(256) void bed_to_packed_kernel(const uint8_t* __restrict__ bed, int64_t N, int64_t M, int64_t nb,
                                                            uint8_t* __restrict__ out, int64_t ld, unsigned long long* __restrict__ counts,
                                                            int64_t tile_y0) {
    __shared__ uint32_t s_t[16 * BT_JD * BT_PITCH];            // row (s % 16) * 32 + s / 16 holds sample s of the tile: 32 dwords = 512 SNPs
    __shared__ unsigned int s_cnt[4];
    const int tid = threadIdx.x;
    const int64_t m0 = (tile_y0 + blockIdx.y) * BT_SNPS, jd0 = (int64_t)blockIdx.x * BT_JD;      // (sample tiles run fastest over the grid)
    if (tid < 4) s_cnt[tid] = 0;
    unsigned int c1 = 0, c2 = 0, c3 = 0, cv = 0;
    for (int t = tid; t < (BT_SNPS / 16) * BT_JD; t += 256) {
        const int jd = t % BT_JD, rg = t / BT_JD;
        const int64_t j = 4 * (jd0 + jd), m = m0 + 16 * rg;      // first sample byte column, first SNP row of the task
        uint32_t d[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d[r] = 0u;
            if (m + r < M && j < nb) {
                const uint8_t* src = bed + (m + r) * nb + j;
                if (j + 4 <= nb) __builtin_memcpy(&d[r], src, 4);            // (rows start at any byte: an unaligned dword load)
                else for (int c = 0; c < (int)(nb - j); ++c) d[r] |= (uint32_t)src[c] << (8 * c);
            }
        }
        // e[q][c]: the bytes of rows 4q..4q+3 at byte column c = one 4 SNP x 4 sample block of 2-bit fields
        uint32_t e[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t in[4] = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
            transpose4x4_bytes(in, e[q]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t jc = j + c;
            uint32_t smask = 0u;
            if (jc < nb) {
                const int ns = (int)((N - 4 * jc < 4) ? (N - 4 * jc) : 4);      // samples in this byte
                smask = ns == 4 ? 0xFFu : ((1u << (2 * ns)) - 1u);
            }
            uint32_t tq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t valid = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + 4 * q + r < M) valid |= smask << (8 * r);
                const uint32_t w = e[q][c];
                const uint32_t hi = w & 0xAAAAAAAAu, lo = w & 0x55555555u;        // PLINK 00,01,10,11 -> 2,3,1,0
                uint32_t g = ((~hi) & 0xAAAAAAAAu) | (lo ^ (hi >> 1));
                g &= valid;                                          // padding samples / SNPs past M stay 0
                const uint32_t gl = g & 0x55555555u, gh = (g >> 1) & 0x55555555u;
                c3 += __popc(gl & gh); c2 += __popc(gh & ~gl); c1 += __popc(gl & ~gh); cv += __popc(valid & 0x55555555u);
                tq[q] = transpose4x4_2bit(g);                        // byte i = sample 4 jc + i, its 4 SNPs of rows 4q..4q+3
            }
            uint32_t o[4];                                           // o[i] = sample 4 jc + i: the 16 SNPs of the task
            transpose4x4_bytes(tq, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) s_t[((4 * c + i) * BT_JD + jd) * BT_PITCH + rg] = o[i];
        }
    }
    __syncthreads();
    for (int e4 = tid; e4 < 16 * BT_JD * (BT_SNPS / 64); e4 += 256) {        // 512 sample rows x 128 bytes, 16 B per store
        const int srow = e4 / (BT_SNPS / 64), c16 = e4 % (BT_SNPS / 64);
        const int64_t smp = 16 * jd0 + srow, col = m0 / 4 + 16 * c16;
        if (smp < N && col < ld) {
            const uint32_t* src = &s_t[((srow % 16) * BT_JD + srow / 16) * BT_PITCH + 4 * c16];
            *reinterpret_cast<uint4*>(out + smp * ld + col) = make_uint4(src[0], src[1], src[2], src[3]);
        }
    }
    atomicAdd(&s_cnt[1], c1); atomicAdd(&s_cnt[2], c2); atomicAdd(&s_cnt[3], c3); atomicAdd(&s_cnt[0], cv - c1 - c2 - c3);
    __syncthreads();
    if (tid < 4 && s_cnt[tid]) atomicAdd(&counts[tid], (unsigned long long)s_cnt[tid]);
}

// flip 0 <-> 2 (1 and 3 unchanged) on every byte if the mean code is >= 1 (src/snp_reader.py:109-110, with the missing
// code kept at 3 like the packed path of the reference, pack2bit.cu:29); decided on the device from the counts.
__global__ __launch_bounds__(256) void bed_flip_kernel(uint8_t* __restrict__ out, int64_t N, int64_t M, int64_t ld,
                                                       const unsigned long long* __restrict__ counts, int32_t* __restrict__ flipped) {
    const double mean = (double)(counts[1] + 2 * counts[2] + 3 * counts[3]) / ((double)N * (double)M);
    const bool flip = mean >= 1.0;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *flipped = flip ? 1 : 0;
    if (!flip) return;
    const int64_t mp16 = ((M + 3) / 4 + 15) / 16;            // 16-byte pieces that hold SNPs
    const int64_t r = blockIdx.y;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < mp16; c += (int64_t)gridDim.x * 256) {
        uint4* pz = reinterpret_cast<uint4*>(out + r * ld + 16 * c);
        uint4 v = *pz;
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t snp0 = (16 * c + 4 * k) * 4;       // first SNP of this word
            uint32_t keep = 0xFFFFFFFFu;                      // fields that hold real SNPs (the row tail must stay 0)
            if (snp0 + 16 > M) keep = snp0 >= M ? 0u : ((1u << (2 * (M - snp0))) - 1u);
            w[k] = (w[k] ^ ((~w[k] & 0x55555555u) << 1)) & keep;
        }
        *pz = v;
    }
}

// =================================================================================================
// supervised_ce: weight * CrossEntropyLoss(sum) applied to the softmax OUTPUT of head 0 as logits
// (neural_admixture.py:293,470-473).  One block; thread t takes samples t, t+256, ...
//   loss_i = logsumexp(q_i) - q_i[y_i],  d/dq_ij = softmax(q_i)_j - [j == y_i]
// The gradient is added to chunk 0 of head 0's dQ slab (mlp_bwd sums the chunks), the weighted loss
// goes to one slot of the losspart array.  Fixed order -> deterministic.
// =================================================================================================
__global__ __launch_bounds__(256) void supervised_ce_kernel(const float* __restrict__ Q, int SP, int k, int kp,
                                                           const int32_t* __restrict__ labels, const int32_t* __restrict__ idx,
                                                           int b, float weight, float* __restrict__ dq0, float* __restrict__ loss_slot) {
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int i = tid; i < b; i += 256) {
        const float* q = Q + (int64_t)i * SP;
        const int y = labels[idx ? idx[i] : i];
        float qm = q[0];
        for (int j = 1; j < k; ++j) qm = fmaxf(qm, q[j]);
        float se = 0.f;
        for (int j = 0; j < k; ++j) se += expf(q[j] - qm);
        const float inv = 1.f / se;
        float qy = 0.f;
        for (int j = 0; j < k; ++j) {
            const float sm = expf(q[j] - qm) * inv;
            const float onehot = (j == y) ? 1.f : 0.f;
            if (j == y) qy = q[j];
            dq0[(int64_t)i * kp + j] += weight * (sm - onehot);
        }
        acc += (double)(qm + logf(se)) - (double)qy;
    }
    acc = wave_sum_all_f64(acc);
    __shared__ double s_l[4];
    if ((tid & 63) == 0) s_l[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) *loss_slot = (float)((double)weight * ((s_l[0] + s_l[1]) + (s_l[2] + s_l[3])));
}

// =================================================================================================
// mlp_bwd_b: weight gradients, split over samples: grid (ceil(Hd/256), splits of SJ samples).
//   dWk[k][h] = sum_i dL[i][k] H[i][h] ; dW1[h][c] = sum_i dHpre[i][h] Zn[i][c] ; db1[h] = sum_i dHpre[i][h]
//   dbk[k] = sum_i dL[i][k] ; dg[c] = sum_i dgp[i][c]      (block x == 0)
// writes small_part[split][n_small]; small_reduce sums the splits in fixed order.
// =================================================================================================
__global__ __launch_bounds__(256) void mlp_bwd_b_kernel(nadm_heads_t hd, int b, const float* __restrict__ Zn,
                                                        const float* __restrict__ H, const float* __restrict__ dL,
                                                        const float* __restrict__ dHpre, const float* __restrict__ dgp,
                                                        float* __restrict__ small_part) {
    mlp_bwd_b_block(hd, b, Zn, H, dL, dHpre, dgp, small_part, blockIdx.x, blockIdx.y);
}

// sum over the sample splits of element e, rows in order (bit-identical whichever kernel calls it); 32 rows per trip are
// loaded unconditionally (clamped row) so that they are in flight together -- the plain loop over a run-time count waited per
// unrolled group
__device__ __forceinline__ float sum_splits(const float* __restrict__ part, int splits, int n, int e) {
    float a = 0.f;
    for (int j0 = 0; j0 < splits; j0 += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = part[(int64_t)(j0 + u < splits ? j0 + u : splits - 1) * n + e];
#pragma unroll
        for (int u = 0; u < 32; ++u) if (j0 + u < splits) a += v[u];
    }
    return a;
}
__global__ void small_reduce_kernel(const float* __restrict__ part, int splits, int n, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = sum_splits(part, splits, n, e);
}

// =================================================================================================
// Adam + clamp (torch.optim.Adam fused kernel closed form; neural_admixture.py:187-204,411-412)
// =================================================================================================
__device__ __forceinline__ void adam_segment(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                             int64_t n, int64_t clamp_from, float step_size, float inv_bc2, float grad_scale,
                                             int64_t first, int64_t stride) {
    for (int64_t e = first; e < n; e += stride) {
        if (e + 4 <= n) {
            const float4 P4 = *reinterpret_cast<const float4*>(p + e);
            const float4 G4 = *reinterpret_cast<const float4*>(g + e);
            const float4 M4 = *reinterpret_cast<const float4*>(m + e);
            const float4 V4 = *reinterpret_cast<const float4*>(v + e);
            float pp[4] = {P4.x, P4.y, P4.z, P4.w}, gg[4] = {G4.x, G4.y, G4.z, G4.w};
            float mm[4] = {M4.x, M4.y, M4.z, M4.w}, vv[4] = {V4.x, V4.y, V4.z, V4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) pp[q] = adam_element(pp[q], gg[q], mm[q], vv[q], step_size, inv_bc2, grad_scale, e + q >= clamp_from);
            *reinterpret_cast<float4*>(p + e) = make_float4(pp[0], pp[1], pp[2], pp[3]);
            *reinterpret_cast<float4*>(m + e) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<float4*>(v + e) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            for (int64_t q = e; q < n; ++q) {
                float mq = m[q], vq = v[q];
                p[q] = adam_element(p[q], g[q], mq, vq, step_size, inv_bc2, grad_scale, q >= clamp_from);
                m[q] = mq; v[q] = vq;
            }
        }
    }
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   int64_t clamp_from, float step_size, float inv_bc2,
                                                   float grad_scale) {
    adam_segment(p, g, m, v, n, clamp_from, step_size, inv_bc2, grad_scale, ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4,
                 (int64_t)gridDim.x * blockDim.x * 4);
}

// =================================================================================================
// pack / unpack (pack2bit.cu:10-62).  One thread per output dword (16 genotypes).
// =================================================================================================
__global__ void pack2bit_kernel(const uint8_t* __restrict__ g, uint8_t* __restrict__ out, int64_t rows, int64_t M,
                                int64_t ld) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // dword within row
    const int64_t r = blockIdx.y;
    if (r >= rows || w * 4 >= ld) return;
    const uint8_t* src = g + r * M + w * 16;
    uint32_t word = 0;
    const int64_t left = M - w * 16;
    if (left >= 16) {
#pragma unroll
        for (int s = 0; s < 16; ++s) word |= (uint32_t)(src[s] & 3u) << (2 * s);
    } else {
        for (int s = 0; s < left; ++s) word |= (uint32_t)(src[s] & 3u) << (2 * s);
    }
    *reinterpret_cast<uint32_t*>(out + r * ld + w * 4) = word;
}

// (interop only: no kernel of the step reads unpacked genotypes.)  A thread takes 4 packed bytes = 16 genotypes; a byte b spreads into a
// dword of four codes as (b | b << 6 | b << 12 | b << 18) & 0x03030303 (the shifts put the four fields at bit 0 of bytes 0..3).  One
// packed byte per thread and four single-byte stores moved 1.7 TB/s (profiles/r05_io_timing.txt).
__global__ void unpack2bit_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t rows, int64_t M,
                                  int64_t ld) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // packed dword within the row
    const int64_t r = blockIdx.y;
    if (r >= rows || c * 16 >= M) return;
    const uint8_t* src = in + r * ld + 4 * c;
    uint32_t v = 0;
    if (4 * c + 4 <= ld) __builtin_memcpy(&v, src, 4);
    else for (int k = 0; k < (int)(ld - 4 * c); ++k) v |= (uint32_t)src[k] << (8 * k);
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t bq = (v >> (8 * k)) & 0xFFu;
        o[k] = (bq | (bq << 6) | (bq << 12) | (bq << 18)) & 0x03030303u;
    }
    uint8_t* dst = out + r * M + 16 * c;
    const int64_t n = M - 16 * c;
    if (n >= 16) {
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
        else __builtin_memcpy(dst, o, 16);                                 // (M need not be a multiple of 16: rows start at any byte)
    } else {
        for (int k = 0; k < (int)n; ++k) dst[k] = (uint8_t)(o[k >> 2] >> (8 * (k & 3)));
    }
}

// =================================================================================================
// synthetic genotypes: G ~ Binomial(2, Qt.F), missing -> 3, written packed.
// Counter-based hash RNG keyed by (seed, global row, SNP): independent of launch geometry.
// =================================================================================================
__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void synth_kernel(uint8_t* __restrict__ xp, int64_t rows, int64_t row0, int64_t M, int64_t ld,
                             const float* __restrict__ Qt, const float* __restrict__ Fq, int K, float missing,
                             uint64_t seed) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // packed byte
    const int64_t r = blockIdx.y;
    if (r >= rows || c >= ld) return;
    uint32_t byte = 0;
    for (int s = 0; s < 4; ++s) {
        const int64_t m = c * 4 + s;
        if (m >= M) break;
        float f = 0.f;
        for (int k = 0; k < K; ++k) f = fmaf(Qt[r * K + k], Fq[(int64_t)k * M + m], f);
        const uint64_t h = mix64(seed ^ mix64((uint64_t)(row0 + r) * 0x9E3779B97F4A7C15ull + (uint64_t)m));
        const float u1 = (float)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
        const float u2 = (float)((h >> 24) & 0xFFFFFF) * (1.0f / 16777216.0f);
        const float u3 = (float)((h >> 48) & 0xFFFF) * (1.0f / 65536.0f);
        uint32_t gcode = (u1 < f ? 1u : 0u) + (u2 < f ? 1u : 0u);
        if (u3 < missing) gcode = 3u;
        byte |= gcode << (2 * s);
    }
    xp[r * ld + c] = (uint8_t)byte;
}

}  // namespace nadm

using namespace nadm;

// -------------------------------------------------------------------------------------------------
extern "C" int nadm_abi_version(void) { return NADM_ABI_VERSION; }
extern "C" const char* nadm_last_error(void) { return err_buf(); }

extern "C" int nadm_pad_k(int k) {
    if (k <= 0 || k > NADM_MAX_K) return -1;
    if (k <= 16) return (k + 3) & ~3;
    if (k <= 24) return 24;
    if (k <= 32) return 32;
    if (k <= 48) return 48;
    return 64;
}
static int pad_c(int c) {
    if (c <= 0 || c > 32) return -1;
    if (c <= 16) return (c + 3) & ~3;
    return c <= 24 ? 24 : 32;
}

extern "C" int nadm_heads_init(nadm_heads_t* out, int C, int Hd, const int32_t* ks, int n) {
    if (!out || !ks) return fail("nadm_heads_init: null pointer");
    if (n <= 0 || n > NADM_MAX_HEADS) return fail("nadm_heads_init: 1..32 heads supported");
    if (pad_c(C) < 0) return fail("nadm_heads_init: n_components must be in 1..32");
    if (Hd <= 0 || Hd > 8192) return fail("nadm_heads_init: hidden size must be in 1..8192");
    memset(out, 0, sizeof(*out));
    out->n_heads = n; out->C = C; out->CP = pad_c(C); out->Hd = Hd;
    int off = 0;
    out->g_off = off; off += C;
    out->w1_off = off; off += Hd * C;
    out->b1_off = off; off += Hd;
    int q = 0;
    for (int h = 0; h < n; ++h) {
        const int kp = nadm_pad_k(ks[h]);
        if (kp < 0) return fail("nadm_heads_init: K must be in 1..64");
        if (h > 0 && ks[h] <= ks[h - 1]) return fail("nadm_heads_init: ks must be strictly ascending");
        out->k[h] = ks[h]; out->kp[h] = kp; out->qoff[h] = q; q += kp;
        out->wk_off[h] = off; off += ks[h] * Hd;
        out->bk_off[h] = off; off += ks[h];
    }
    out->SP = q;
    out->n_small = off;
    return 0;
}

extern "C" int32_t nadm_sample_splits(int b) { return (b + SJ - 1) / SJ; }

extern "C" int nadm_pack2bit_host(const uint8_t* g, uint8_t* out, int64_t N, int64_t M, int64_t ld) {
    if (!g || !out) return fail("nadm_pack2bit_host: null pointer");
    if (ld * 4 < M) return fail("nadm_pack2bit_host: ld < ceil(M/4)");
    auto work = [=](int64_t r_begin, int64_t r_end) {
        for (int64_t r = r_begin; r < r_end; ++r) {
            const uint8_t* src = g + r * M;
            uint8_t* dst = out + r * ld;
            const int64_t full = M / 4;
            for (int64_t c = 0; c < full; ++c) {
                const uint8_t* s = src + 4 * c;
                dst[c] = (uint8_t)((s[0] & 3) | ((s[1] & 3) << 2) | ((s[2] & 3) << 4) | ((s[3] & 3) << 6));
            }
            if (full * 4 < M) {
                uint8_t v = 0;
                for (int64_t s = full * 4; s < M; ++s) v |= (uint8_t)((src[s] & 3) << (2 * (s - full * 4)));
                dst[full] = v;
            }
            for (int64_t c = (M + 3) / 4; c < ld; ++c) dst[c] = 0;
        }
    };
    int nt = (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    if ((int64_t)nt > N) nt = (int)(N > 0 ? N : 1);
    if (N * M < (1 << 22)) nt = 1;
    if (nt == 1) {
        work(0, N);
    } else {
        std::vector<std::thread> th;
        const int64_t per = (N + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            const int64_t r0 = t * per, r1 = r0 + per < N ? r0 + per : N;
            if (r0 < r1) th.emplace_back(work, r0, r1);
        }
        for (auto& t : th) t.join();
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// PLINK .bed (SNP-major, 4 samples/byte) -> sample-major packed (4 SNPs/byte): a 2-bit matrix transpose
// with the reference's recode table [2,3,1,0] (utils.pyx:52).  Each worker owns blocks of 256 SNPs so that
// it writes 64 contiguous bytes per sample row.
// -------------------------------------------------------------------------------------------------
extern "C" int nadm_bed_to_packed(const uint8_t* bed, int64_t N, int64_t M, uint8_t* out, int64_t ld, int64_t* counts,
                                  int32_t flip_if_mean_ge1, int32_t* flipped) {
    if (!bed || !out || !counts) return fail("nadm_bed_to_packed: null pointer");
    if (ld * 4 < M) return fail("nadm_bed_to_packed: ld < ceil(M/4)");
    const int64_t nb = (N + 3) / 4;                       // bytes per SNP in the .bed
    int nt = (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    const int64_t nblk = (M + 255) / 256;
    if (nblk < nt) nt = (int)(nblk > 0 ? nblk : 1);
    std::vector<std::array<int64_t, 4>> cnt(nt, std::array<int64_t, 4>{0, 0, 0, 0});
    // word-level version of the device kernel: the bytes of 4 consecutive SNPs at one sample-byte column form a 4 x 4 block
    // of 2-bit fields; recode bitwise, transpose with two delta swaps, one output byte per sample
    auto tr = [](uint32_t w) -> uint32_t {
        uint32_t t = ((w >> 6) ^ w) & 0x00CC00CCu;
        w ^= t ^ (t << 6);
        t = ((w >> 12) ^ w) & 0x0000F0F0u;
        return w ^ t ^ (t << 12);
    };
    auto work = [&](int t) {
        uint8_t rows[4][64];
        for (int64_t blk = t; blk < nblk; blk += nt) {
            const int64_t m0 = blk * 256;
            const int64_t nm = (M - m0 < 256) ? (M - m0) : 256;
            const int64_t ncol = (nm + 3) / 4;            // output bytes per row in this block
            for (int64_t bi = 0; bi < nb; ++bi) {
                const int ns = (int)((N - 4 * bi < 4) ? (N - 4 * bi) : 4);
                const uint32_t smask = ns == 4 ? 0xFFu : ((1u << (2 * ns)) - 1u);
                for (int64_t g = 0; g < ncol; ++g) {
                    uint32_t w = 0, valid = 0;
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < nm) {
                            w |= (uint32_t)bed[(m0 + 4 * g + r) * nb + bi] << (8 * r);
                            valid |= smask << (8 * r);
                        }
                    const uint32_t hi = w & 0xAAAAAAAAu, lo = w & 0x55555555u;
                    const uint32_t gq = (((~hi) & 0xAAAAAAAAu) | (lo ^ (hi >> 1))) & valid;
                    const uint32_t gl = gq & 0x55555555u, gh = (gq >> 1) & 0x55555555u;
                    const int c3 = __builtin_popcount(gl & gh), c2 = __builtin_popcount(gh & ~gl), c1 = __builtin_popcount(gl & ~gh);
                    cnt[t][3] += c3; cnt[t][2] += c2; cnt[t][1] += c1;
                    cnt[t][0] += __builtin_popcount(valid & 0x55555555u) - c1 - c2 - c3;
                    const uint32_t q = tr(gq);
                    rows[0][g] = (uint8_t)q; rows[1][g] = (uint8_t)(q >> 8); rows[2][g] = (uint8_t)(q >> 16); rows[3][g] = (uint8_t)(q >> 24);
                }
                for (int s4 = 0; s4 < ns; ++s4) memcpy(out + (4 * bi + s4) * ld + (m0 >> 2), rows[s4], (size_t)ncol);
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    for (int c = 0; c < 4; ++c) { counts[c] = 0; for (int t = 0; t < nt; ++t) counts[c] += cnt[t][c]; }
    // zero the row padding
    const int64_t mp = (M + 3) / 4;
    if (ld > mp)
        for (int64_t r = 0; r < N; ++r) memset(out + r * ld + mp, 0, (size_t)(ld - mp));
    int did = 0;
    if (flip_if_mean_ge1 && N > 0 && M > 0) {
        const double mean = (double)(counts[1] + 2 * counts[2] + 3 * counts[3]) / ((double)N * (double)M);
        if (mean >= 1.0) {
            did = 1;                                      // 0 <-> 2, 1 and 3 unchanged: c ^= ((~c & 1) << 1) on every 2-bit field
            const int64_t tail_fields = M & 3;
            for (int64_t r = 0; r < N; ++r) {
                uint8_t* row = out + r * ld;
                for (int64_t c = 0; c < mp; ++c) row[c] ^= (uint8_t)((~row[c] & 0x55) << 1);
                if (tail_fields) row[mp - 1] &= (uint8_t)((1u << (2 * tail_fields)) - 1);    // keep the tail bits zero
            }
        }
    }
    if (flipped) *flipped = did;
    return 0;
}

extern "C" int nadm_bed_to_packed_dev(const uint8_t* bed_dev, int64_t N, int64_t M, uint8_t* out_dev, int64_t ld, uint64_t* counts_dev,
                                      int32_t flip_if_mean_ge1, int32_t* flipped_dev, void* stream) {
    if (!bed_dev || !out_dev || !counts_dev || !flipped_dev) return fail("nadm_bed_to_packed_dev: null pointer");
    if (ld % 16 != 0 || ld * 4 < M) return fail("nadm_bed_to_packed_dev: ld must be a multiple of 16 and >= ceil(M/4)");
    if (N <= 0 || M <= 0) return fail("nadm_bed_to_packed_dev: empty matrix");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = (N + 3) / 4;
    if (hipMemsetAsync(counts_dev, 0, 4 * sizeof(uint64_t), st) != hipSuccess || hipMemsetAsync(flipped_dev, 0, sizeof(int32_t), st) != hipSuccess)
        return fail("nadm_bed_to_packed_dev: memset failed");
    // the tile grid covers ld bytes per row, so the row padding is written (as zeros) too
    const int64_t tiles_y = (ld + BT_SNPS / 4 - 1) / (BT_SNPS / 4);
    for (int64_t y0 = 0; y0 < tiles_y; y0 += 65535) {         // grid.y limit: SNP tiles in chunks (33.5 M SNPs each; whole-genome call sets have more)
        const int64_t ny = tiles_y - y0 < 65535 ? tiles_y - y0 : 65535;
        dim3 grid((unsigned)((nb + 4 * BT_JD - 1) / (4 * BT_JD)), (unsigned)ny);
        hipLaunchKernelGGL(bed_to_packed_kernel, grid, dim3(256), 0, st, bed_dev, N, M, nb, out_dev, ld, (unsigned long long*)counts_dev, y0);
    }
    if (flip_if_mean_ge1) {
        const int64_t mp16 = ((M + 3) / 4 + 15) / 16;
        for (int64_t r0 = 0; r0 < N; r0 += 65535) {           // grid.y limit: rows in chunks
            const int64_t nr = N - r0 < 65535 ? N - r0 : 65535;
            dim3 g2((unsigned)((mp16 + 255) / 256 < 64 ? (mp16 + 255) / 256 : 64), (unsigned)nr);
            hipLaunchKernelGGL(bed_flip_kernel, g2, dim3(256), 0, st, out_dev + r0 * ld, N, M, ld, (const unsigned long long*)counts_dev, flipped_dev);
        }
    }
    return check_launch("bed_to_packed_dev");
}

extern "C" int nadm_pack2bit(const uint8_t* g_dev, uint8_t* out_dev, int64_t rows, int64_t M, int64_t ld, void* stream) {
    if (!g_dev || !out_dev) return fail("nadm_pack2bit: null pointer");
    if (ld % 4 != 0 || ld * 4 < M) return fail("nadm_pack2bit: ld must be a multiple of 4 and >= ceil(M/4)");
    if (rows == 0 || M == 0) return 0;
    if (rows > 65535 * 1024ll) return fail("nadm_pack2bit: too many rows for one call");
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid((unsigned)((ld / 4 + 255) / 256), (unsigned)nr), block(256);
        hipLaunchKernelGGL(pack2bit_kernel, grid, block, 0, (hipStream_t)stream, g_dev + r0 * M, out_dev + r0 * ld, nr, M, ld);
    }
    return check_launch("pack2bit");
}

extern "C" int nadm_unpack2bit(const uint8_t* in_dev, uint8_t* out_dev, int64_t rows, int64_t M, int64_t ld, void* stream) {
    if (!in_dev || !out_dev) return fail("nadm_unpack2bit: null pointer");
    if (ld * 4 < M) return fail("nadm_unpack2bit: ld < ceil(M/4)");
    if (rows == 0 || M == 0) return 0;
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid((unsigned)(((M + 15) / 16 + 255) / 256), (unsigned)nr), block(256);
        hipLaunchKernelGGL(unpack2bit_kernel, grid, block, 0, (hipStream_t)stream, in_dev + r0 * ld, out_dev + r0 * M, nr, M, ld);
    }
    return check_launch("unpack2bit");
}

// test hook (nadm_test_force_generic_mlp): the generic kernels also where the register-resident ones apply
#ifdef NADM_TEST_HOOKS          // the test build only (csrc/build.sh -> libnadm_testhooks.so)
static int g_force_generic_mlp = 0;
extern "C" void nadm_test_force_generic_mlp(int32_t on) { g_force_generic_mlp = on != 0; }
#else
constexpr int g_force_generic_mlp = 0;
#endif

static int mlp_fwd_impl(const nadm_heads_t* hd, const float* small, const float* zpart, int64_t n_chunks, int32_t b,
                        float* Z, float* rinv, float* Zn, float* H, float* Q, uint4* qimg, int64_t qimg_head_u4, void* stream) {
    if (!hd || !small || !zpart || !Z || !rinv || !Zn || !H || !Q) return fail("nadm_mlp_fwd: null pointer");
    if (b <= 0) return fail("nadm_mlp_fwd: empty batch");
    if (hd->Hd <= 256 * MLP_JMAX && hd->C <= 8 && !g_force_generic_mlp) {
        const dim3 grid((b + MLP_SB - 1) / MLP_SB);
        const size_t lds = (size_t)(MLP_SB + 1) * hd->SP * 4;                    // s_logit + the head biases
        const bool c8 = hd->C == 8 && ((reinterpret_cast<uintptr_t>(small) + 4 * (size_t)hd->w1_off) & 15) == 0;
#define NADM_FWD_LAUNCH(JH, C8) hipLaunchKernelGGL((mlp_fwd_fast_kernel<MLP_SB, JH, C8>), grid, dim3(256), lds, (hipStream_t)stream, *hd, small, zpart, n_chunks, b, Z, rinv, Zn, H, Q, qimg, qimg_head_u4)
        if (hd->Hd <= 1024) { if (c8) NADM_FWD_LAUNCH(4, true); else NADM_FWD_LAUNCH(4, false); }
        else { if (c8) NADM_FWD_LAUNCH(8, true); else NADM_FWD_LAUNCH(8, false); }
#undef NADM_FWD_LAUNCH
    } else if (hd->Hd <= 2048) {
        const size_t lds = (size_t)(1024 + MLP_SB * (hd->CP + hd->Hd + hd->SP)) * 4;
        hipLaunchKernelGGL((mlp_fwd_kernel<MLP_SB>), dim3((b + MLP_SB - 1) / MLP_SB), dim3(256), lds, (hipStream_t)stream, *hd, small, zpart, n_chunks, b, Z, rinv, Zn, H, Q, qimg, qimg_head_u4);
    } else {
        const size_t lds = (size_t)(1024 + hd->CP + hd->Hd + hd->SP) * 4;
        hipLaunchKernelGGL((mlp_fwd_kernel<1>), dim3(b), dim3(256), lds, (hipStream_t)stream, *hd, small, zpart, n_chunks, b, Z, rinv, Zn, H, Q, qimg, qimg_head_u4);
    }
    return check_launch("mlp_fwd");
}

extern "C" int nadm_mlp_fwd(const nadm_heads_t* hd, const float* small, const float* zpart, int64_t n_chunks, int32_t b,
                            float* Z, float* rinv, float* Zn, float* H, float* Q, void* stream) {
    return mlp_fwd_impl(hd, small, zpart, n_chunks, b, Z, rinv, Zn, H, Q, nullptr, 0, stream);
}

extern "C" int nadm_mlp_fwd_images(const nadm_heads_t* hd, const float* small, const float* zpart, int64_t n_chunks, int32_t b,
                                   float* Z, float* rinv, float* Zn, float* H, float* Q, void* qimg, int64_t qimg_head_bytes, void* stream) {
    if (!qimg) return fail("nadm_mlp_fwd_images: qimg is NULL (use nadm_mlp_fwd)");
    if (((uintptr_t)qimg & 15) || (qimg_head_bytes & 15)) return fail("nadm_mlp_fwd_images: qimg and the head stride must be multiples of 16 bytes");
    if (qimg_head_bytes < nadm_q_image_bytes(b)) return fail("nadm_mlp_fwd_images: head stride smaller than nadm_q_image_bytes(b)");
    return mlp_fwd_impl(hd, small, zpart, n_chunks, b, Z, rinv, Zn, H, Q, static_cast<uint4*>(qimg), qimg_head_bytes / 16, stream);
}

extern "C" int nadm_mlp_bwd_weights(const nadm_heads_t* hd, int32_t b, const float* Zn, const float* H, const float* dL,
                                    const float* dHpre, const float* dgp, float* small_part, float* grad_small, void* stream);

// First level of the dQ reduction for TALL slabs.  Pass 2 leaves one slab row [b, kp] per SNP chunk (1954 rows = 50 MB at
// M = 500k, K = 8); the MLP backward runs one block per 4 samples and each of its blocks walks all those rows 128 bytes at a
// time, 16 rows in flight per thread: latency-bound, ~1 us per trip -- 8 us at 1954 rows, growing with M.  Here every thread
// owns one float4 column of the slab and adds the rows y, y + DQ_R, y + 2 DQ_R, ... into row y -- whole rows are read with full
// lines by ~450 blocks at 4.5 TB/s (11 us per 50 MB) -- so the MLP backward is left with DQ_R rows per head.  Measured at
// M = 500k: 13.2 + 11.1 us with the fold against 21.3 us without, hence the fold only above DQ_R_MIN_ROWS rows (M > 655k).
// In place: row y (< DQ_R) is read only by the threads that also write it.  Fixed order, no atomics.
constexpr int DQ_R_MIN_ROWS = 2560;
constexpr int DQ_R = 64;
__global__ __launch_bounds__(256) void dq_prereduce_kernel(float* __restrict__ dq, DqChunks n, nadm_heads_t hd, int b) {
    const int hh = blockIdx.z;
    const int64_t nch = n.n[hh];
    if (nch <= DQ_R) return;
    int64_t base = 0;
    for (int h = 0; h < hh; ++h) base += n.full[h] * b * hd.kp[h];
    const int64_t row4 = (int64_t)b * hd.kp[hh] / 4;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= row4) return;
    float4* slab = reinterpret_cast<float4*>(dq + base) + col;
    const int y = blockIdx.y;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int64_t ch = y; ch < nch; ch += DQ_R) {
        const float4 v = slab[ch * row4];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    slab[(int64_t)y * row4] = a;
}

// out[e] = sum over rows r of src[r * n + e], rows added in ascending order within each of 32 interleaved groups, groups combined
// in ascending order: fixed order, no atomics.  The SNP-sharded step folds its partial Z / dQ slabs with it before the all-reduce.
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ src, int64_t rows, int64_t n4, float* __restrict__ out) {
    __shared__ float4 s_part[256];
    const int tid = threadIdx.x, c = tid & 7, g = tid >> 3;           // 8 float4 columns x 32 row groups per block
    const int64_t col = (int64_t)blockIdx.x * 8 + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n4) {
        const float4* base = reinterpret_cast<const float4*>(src) + col;
        constexpr int D = 8;                                           // rows per trip: loaded unconditionally (clamped), added under a mask
        for (int64_t r0 = g; r0 < rows; r0 += 32 * D) {
            float4 v[D];
#pragma unroll
            for (int u = 0; u < D; ++u) { const int64_t r = r0 + 32 * u; v[u] = base[(r < rows ? r : rows - 1) * n4]; }
#pragma unroll
            for (int u = 0; u < D; ++u)
                if (r0 + 32 * u < rows) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
    }
    s_part[tid] = a;
    __syncthreads();
    if (g == 0 && col < n4) {
        float4 t = s_part[c];
        for (int gg = 1; gg < 32; ++gg) { const float4 v = s_part[gg * 8 + c]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        reinterpret_cast<float4*>(out)[col] = t;
    }
}

extern "C" int nadm_sum_rows(const float* src, int64_t rows, int64_t n, float* out, void* stream) {
    if (!src || !out) return fail("nadm_sum_rows: null pointer");
    if (rows <= 0 || n <= 0 || (n & 3)) return fail("nadm_sum_rows: need rows > 0 and a row length that is a positive multiple of 4");
    if (((uintptr_t)src | (uintptr_t)out) & 15) return fail("nadm_sum_rows: 16-byte alignment");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((n4 + 7) / 8)), dim3(256), 0, (hipStream_t)stream, src, rows, n4, out);
    return check_launch("sum_rows");
}

static int mlp_bwd_impl(const nadm_heads_t* hd, const float* small, float* dqpart, int64_t M, int32_t b,
                        const float* Z, const float* rinv, const float* Zn, const float* H, const float* Q,
                        float* dL, float* dHpre, float* dgp, float* small_part, float* dZ, float* grad_small,
                        const float* losspart, int64_t n_loss, double* loss_acc, void* dzimg, int32_t* dz_counters, void* stream) {
    if (!hd || !small || !dqpart || !Z || !rinv || !Zn || !H || !Q || !dL || !dHpre || !dgp || !small_part || !dZ)
        return fail("nadm_mlp_bwd: null pointer");
    if (dzimg && (!dz_counters || hd->CP > 8 || ((uintptr_t)dzimg & 15)))
        return fail("nadm_mlp_bwd_image: the image needs its group counters, C <= 8 and 16-byte alignment");
    bool image_done = false;
    if (n_loss > 0 && (!losspart || !loss_acc)) return fail("nadm_mlp_bwd: n_loss > 0 needs losspart and loss_acc");
    if (b <= 0) return fail("nadm_mlp_bwd: empty batch");
    hipStream_t st = (hipStream_t)stream;
    DqChunks dqc;
    for (int h = 0; h < NADM_MAX_HEADS; ++h) dqc.n[h] = dqc.full[h] = h < hd->n_heads ? nadm_decode_chunks(M, hd->kp[h]) : 0;
    {   // tall slabs are folded to DQ_R rows first (dqpart is scratch of the step: reduced in place)
        int64_t max_rows = 0, max_row4 = 0;
        for (int h = 0; h < hd->n_heads; ++h) {
            if (dqc.n[h] > max_rows) max_rows = dqc.n[h];
            if ((int64_t)b * hd->kp[h] / 4 > max_row4) max_row4 = (int64_t)b * hd->kp[h] / 4;
        }
        if (max_rows > DQ_R_MIN_ROWS) {
            hipLaunchKernelGGL(dq_prereduce_kernel, dim3((unsigned)((max_row4 + 255) / 256), DQ_R, hd->n_heads), dim3(256), 0, st,
                               dqpart, dqc, *hd, b);
            if (check_launch("dq_prereduce")) return 1;
            for (int h = 0; h < hd->n_heads; ++h)                       // the kernel's rule: heads with more than DQ_R rows were folded
                if (dqc.n[h] > DQ_R) dqc.n[h] = DQ_R;
        }
    }
    if (hd->Hd <= 256 * MLP_JMAX && hd->C <= 8 && !g_force_generic_mlp) {
        const dim3 grid((b + MLP_SB - 1) / MLP_SB + (n_loss > 0 ? 1 : 0));          // + the loss block
        const size_t lds = (size_t)3 * MLP_SB * hd->SP * 4;                      // s_dl + the block's Q rows + dL
        const bool c8 = hd->C == 8 && ((reinterpret_cast<uintptr_t>(small) + 4 * (size_t)hd->w1_off) & 15) == 0;
#define NADM_BWD_LAUNCH(JH, C8) hipLaunchKernelGGL((mlp_bwd_a_fast_kernel<MLP_SB, JH, C8>), grid, dim3(256), lds, st, *hd, small, dqpart, dqc, b, Z, rinv, H, Q, \
                                                   dL, dHpre, dgp, dZ, losspart, n_loss, loss_acc, (uint4*)dzimg, dz_counters)
        image_done = dzimg != nullptr;
        if (hd->Hd <= 1024) { if (c8) NADM_BWD_LAUNCH(4, true); else NADM_BWD_LAUNCH(4, false); }
        else { if (c8) NADM_BWD_LAUNCH(8, true); else NADM_BWD_LAUNCH(8, false); }
#undef NADM_BWD_LAUNCH
    } else if (hd->Hd <= 2048) {
        const size_t lds = (size_t)(1024 + MLP_SB * (hd->SP + hd->CP + hd->Hd)) * 4;
        hipLaunchKernelGGL((mlp_bwd_a_kernel<MLP_SB>), dim3((b + MLP_SB - 1) / MLP_SB), dim3(256), lds, st, *hd, small, dqpart, dqc, b, Z, rinv, H, Q,
                           dL, dHpre, dgp, dZ, losspart, n_loss, loss_acc);
    } else {
        const size_t lds = (size_t)(1024 + hd->SP + hd->CP + hd->Hd) * 4;
        hipLaunchKernelGGL((mlp_bwd_a_kernel<1>), dim3(b), dim3(256), lds, st, *hd, small, dqpart, dqc, b, Z, rinv, H, Q,
                           dL, dHpre, dgp, dZ, losspart, n_loss, loss_acc);
    }
    if (check_launch("mlp_bwd")) return 1;
    if (dzimg && !image_done && nadm_dz_image(dZ, b, hd->CP, dzimg, stream)) return 1;     // the generic kernels: the image as a launch of its own
    if (!grad_small) return 0;                       // the caller runs nadm_mlp_bwd_weights itself (possibly on another stream)
    return nadm_mlp_bwd_weights(hd, b, Zn, H, dL, dHpre, dgp, small_part, grad_small, stream);
}

extern "C" int nadm_mlp_bwd(const nadm_heads_t* hd, const float* small, float* dqpart, int64_t M, int32_t b,
                            const float* Z, const float* rinv, const float* Zn, const float* H, const float* Q,
                            float* dL, float* dHpre, float* dgp, float* small_part, float* dZ, float* grad_small,
                            const float* losspart, int64_t n_loss, double* loss_acc, void* stream) {
    return mlp_bwd_impl(hd, small, dqpart, M, b, Z, rinv, Zn, H, Q, dL, dHpre, dgp, small_part, dZ, grad_small, losspart, n_loss, loss_acc,
                        nullptr, nullptr, stream);
}

extern "C" int nadm_mlp_bwd_image(const nadm_heads_t* hd, const float* small, float* dqpart, int64_t M, int32_t b,
                                  const float* Z, const float* rinv, const float* Zn, const float* H, const float* Q,
                                  float* dL, float* dHpre, float* dgp, float* small_part, float* dZ, float* grad_small,
                                  const float* losspart, int64_t n_loss, double* loss_acc, void* dzimg, int32_t* dz_counters, void* stream) {
    if (!dzimg || !dz_counters) return fail("nadm_mlp_bwd_image: null pointer");
    return mlp_bwd_impl(hd, small, dqpart, M, b, Z, rinv, Zn, H, Q, dL, dHpre, dgp, small_part, dZ, grad_small, losspart, n_loss, loss_acc,
                        dzimg, dz_counters, stream);
}

extern "C" int nadm_mlp_bwd_weights(const nadm_heads_t* hd, int32_t b, const float* Zn, const float* H, const float* dL,
                                    const float* dHpre, const float* dgp, float* small_part, float* grad_small, void* stream) {
    if (!hd || !Zn || !H || !dL || !dHpre || !dgp || !small_part || !grad_small) return fail("nadm_mlp_bwd_weights: null pointer");
    if (b <= 0) return fail("nadm_mlp_bwd_weights: empty batch");
    hipStream_t st = (hipStream_t)stream;
    const int splits = nadm_sample_splits(b);
    hipLaunchKernelGGL(mlp_bwd_b_kernel, dim3((hd->Hd + 255) / 256, splits), dim3(256), 0, st, *hd, b, Zn, H, dL, dHpre, dgp, small_part);
    hipLaunchKernelGGL(small_reduce_kernel, dim3((hd->n_small + 255) / 256), dim3(256), 0, st, small_part, splits, hd->n_small, grad_small);
    return check_launch("mlp_bwd_weights");
}

// the partial sums only (the first kernel of nadm_mlp_bwd_weights): used by the variants of pass 3 that cannot host them
extern "C" int nadm_mlp_bwd_weight_parts(const nadm_heads_t* hd, int32_t b, const float* Zn, const float* H, const float* dL,
                                         const float* dHpre, const float* dgp, float* small_part, void* stream) {
    hipLaunchKernelGGL(mlp_bwd_b_kernel, dim3((hd->Hd + 255) / 256, nadm_sample_splits(b)), dim3(256), 0, (hipStream_t)stream, *hd, b, Zn, H, dL,
                       dHpre, dgp, small_part);
    return check_launch("mlp_bwd_weight_parts");
}

// sum of the sample-split partials (fixed order, as small_reduce_kernel) -> grad_small, then -- when Adam state is given --
// the Adam update of the small parameters in the same thread: one launch for the tail of a single-GPU step
__global__ void small_reduce_adam_kernel(const float* __restrict__ part, int splits, int n, float* __restrict__ out,
                                         float* __restrict__ p, AdamFused ad) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float mq = 0.f, vq = 0.f, pq = 0.f;
    if (ad.m != nullptr) { mq = ad.m[e]; vq = ad.v[e]; pq = p[e]; }            // in flight with the partial sums
    const float a = sum_splits(part, splits, n, e);
    out[e] = a;
    if (ad.m != nullptr) {
        p[e] = adam_element(pq, a, mq, vq, ad.step_size, ad.inv_bc2, ad.grad_scale, false);
        ad.m[e] = mq; ad.v[e] = vq;
    }
}

extern "C" int nadm_small_grads(const float* small_part, int32_t splits, int32_t n_small, float* grad_small, float* small,
                                const nadm_adam_t* adam, void* stream) {
    if (!small_part || !grad_small) return fail("nadm_small_grads: null pointer");
    if (splits <= 0 || n_small <= 0) return fail("nadm_small_grads: empty");
    AdamFused ad{nullptr, nullptr, 0.f, 0.f, 0.f};
    if (adam) {
        if (!adam->m || !adam->v || !small) return fail("nadm_small_grads: Adam state / parameters are NULL");
        if (adam->step < 1) return fail("nadm_small_grads: Adam step is 1-based");
        ad.m = adam->m; ad.v = adam->v; ad.grad_scale = adam->grad_scale;
        adam_scalars(adam->lr, adam->step, &ad.step_size, &ad.inv_bc2);
    }
    hipLaunchKernelGGL(small_reduce_adam_kernel, dim3((n_small + 255) / 256), dim3(256), 0, (hipStream_t)stream, small_part, splits, n_small,
                       grad_small, small, ad);
    return check_launch("small_grads");
}

extern "C" int nadm_supervised_ce(const float* Q, int32_t SP, int32_t k, int32_t kp, const int32_t* labels, const int32_t* idx,
                                  int32_t b, int32_t n_classes, float weight, float* dqpart0, float* loss_slot, void* stream) {
    if (!Q || !labels || !dqpart0 || !loss_slot) return fail("nadm_supervised_ce: null pointer");
    if (b <= 0) return fail("nadm_supervised_ce: empty batch");
    if (k <= 0 || k > kp || kp > SP) return fail("nadm_supervised_ce: need 0 < k <= kp <= SP");
    if (n_classes != k) return fail("nadm_supervised_ce: number of classes must equal K");   // train.py:79
    hipLaunchKernelGGL(supervised_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, Q, SP, k, kp, labels, idx, b, weight, dqpart0, loss_slot);
    return check_launch("supervised_ce");
}

// VCF text -> genotype codes, the semantics of the reference's reader (src/snp_reader.py:73-87): scikit-allel's
// read_vcf(fields=["calldata/GT"], types i1, fills -1) gives two allele indices per call (a missing or absent allele is -1),
// the reader sums them and maps negative sums to 3.  So 0/0 -> 0, 0/1 -> 1, 1|1 -> 2, ./. -> 3, and -- as there -- a
// half-missing call ./1 or a haploid call 1 sums to 0.  buf holds the whole (decompressed) file; out == NULL: only count.
// Output is sample-major uint8 [n_samples, n_variants] like the reference's G.  Variant lines are parsed by std::threads.
extern "C" int nadm_vcf_parse_gt(const char* buf, int64_t len, int64_t* n_samples, int64_t* n_variants, uint8_t* out) {
    if (!buf || !n_samples || !n_variants) return fail("nadm_vcf_parse_gt: null pointer");
    std::vector<int64_t> starts;                       // offsets of the variant lines
    int64_t N = -1;
    for (int64_t p = 0; p < len;) {
        const char* nl = (const char*)memchr(buf + p, '\n', (size_t)(len - p));
        const int64_t e = nl ? (nl - buf) : len;
        if (e > p && buf[p] != '#') starts.push_back(p);
        else if (e > p + 6 && memcmp(buf + p, "#CHROM", 6) == 0) {
            int tabs = 0;
            for (int64_t q = p; q < e; ++q) tabs += buf[q] == '\t';
            N = tabs >= 9 ? tabs - 8 : 0;
        }
        p = e + 1;
    }
    if (N < 0) return fail("nadm_vcf_parse_gt: no #CHROM header line");
    const int64_t M = (int64_t)starts.size();
    *n_samples = N; *n_variants = M;
    if (!out) return 0;
    int bad = 0;
    auto work = [&](int64_t v0, int64_t v1) {
        for (int64_t v = v0; v < v1; ++v) {
            int64_t p = starts[v];
            int col = 0;
            bool gt_first = false;
            while (p < len && buf[p] != '\n' && col < 9) {          // skip the 9 fixed columns; FORMAT must start with GT
                if (col == 8) gt_first = (p + 1 < len && buf[p] == 'G' && buf[p + 1] == 'T' && (p + 2 >= len || buf[p + 2] == ':' || buf[p + 2] == '\t'));
                while (p < len && buf[p] != '\t' && buf[p] != '\n') ++p;
                if (p < len && buf[p] == '\t') ++p;
                ++col;
            }
            for (int64_t s = 0; s < N; ++s) {
                int a[2] = {-1, -1}, na = 0;
                if (p < len && buf[p] != '\n') {
                    if (gt_first) {
                        while (p < len && buf[p] != '\t' && buf[p] != '\n' && buf[p] != ':') {
                            if (buf[p] == '/' || buf[p] == '|') { ++p; continue; }
                            int val = -1;
                            if (buf[p] == '.') ++p;
                            else if (buf[p] >= '0' && buf[p] <= '9') { val = 0; while (p < len && buf[p] >= '0' && buf[p] <= '9') val = val * 10 + (buf[p++] - '0'); }
                            else { __atomic_store_n(&bad, 1, __ATOMIC_RELAXED); ++p; }
                            if (na < 2) a[na] = val;
                            ++na;
                        }
                    }
                    while (p < len && buf[p] != '\t' && buf[p] != '\n') ++p;
                    if (p < len && buf[p] == '\t') ++p;
                }
                const int sum = a[0] + a[1];
                out[s * M + v] = (uint8_t)(sum < 0 ? 3 : (sum > 255 ? 255 : sum));
            }
        }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if ((int64_t)nt > M) nt = (unsigned)(M > 0 ? M : 1);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, M * t / nt, M * (t + 1) / nt);
    for (auto& t : th) t.join();
    return bad ? fail("nadm_vcf_parse_gt: unexpected character in a GT field") : 0;
}

// np.savetxt(path, A, delimiter=' ') for a float32 matrix, byte for byte: numpy formats every element with
// '%.18e' applied to the value widened to double, one row per line, '\n' line ends (reference: src/utils.py:56-66).
// Rows are formatted by std::threads into per-thread buffers and written in order.
extern "C" int nadm_savetxt_f32(const char* path, const float* a, int64_t rows, int64_t cols, int64_t row_stride) {
    if (!path || (!a && rows * cols > 0)) return fail("nadm_savetxt_f32: null pointer");
    if (rows < 0 || cols < 0 || row_stride < cols) return fail("nadm_savetxt_f32: bad shape");
    FILE* f = fopen(path, "wb");
    if (!f) return fail("nadm_savetxt_f32: cannot open output file");
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? (hw > 32 ? 32 : hw) : 4);
    const int64_t block_rows = 2048;                      // rows per thread per round
    std::vector<std::vector<char>> bufs(nt);
    bool ok = true;
    for (int64_t r0 = 0; r0 < rows && ok; r0 += block_rows * nt) {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) {
            const int64_t b0 = r0 + t * block_rows, b1 = b0 + block_rows < rows ? b0 + block_rows : rows;
            bufs[t].clear();
            if (b0 >= rows) continue;
            th.emplace_back([&, t, b0, b1] {
                std::vector<char>& o = bufs[t];
                o.resize((size_t)(b1 - b0) * (size_t)(cols * 26 + 1));
                char* w = o.data();
                for (int64_t r = b0; r < b1; ++r) {
                    const float* row = a + r * row_stride;
                    for (int64_t c = 0; c < cols; ++c) {
                        w += snprintf(w, 27, "%.18e", (double)row[c]);
                        *w++ = (c + 1 < cols) ? ' ' : '\n';
                    }
                    if (cols == 0) *w++ = '\n';
                }
                o.resize((size_t)(w - o.data()));
            });
        }
        for (auto& x : th) x.join();
        for (int t = 0; t < nt && ok; ++t)
            if (!bufs[t].empty() && fwrite(bufs[t].data(), 1, bufs[t].size(), f) != bufs[t].size()) ok = false;
    }
    if (fclose(f) != 0) ok = false;
    return ok ? 0 : fail("nadm_savetxt_f32: write failed");
}

extern "C" int64_t nadm_loglik_blocks(int64_t M) { return (M + 1023) / 1024 * LOGLIK_ROW_SLICES; }

extern "C" int nadm_loglik(const uint8_t* xp, int64_t ld, int64_t rows, int64_t M, const float* P, const float* Q, int32_t K,
                           int32_t q_stride, double eps, double* partial, void* stream) {
    if (!xp || !P || !Q || !partial) return fail("nadm_loglik: null pointer");
    if (ld * 4 < M) return fail("nadm_loglik: ld < ceil(M/4)");
    if (K <= 0 || K > 16) return fail("nadm_loglik: K must be in 1..16");
    if (q_stride < K) return fail("nadm_loglik: q_stride < K");
    if (rows <= 0 || M <= 0) return fail("nadm_loglik: empty matrix");
    if (!(eps >= 1e-9 && eps < 0.5)) return fail("nadm_loglik: eps must be in [1e-9, 0.5) (the reference's is 1e-6; products of 32 factors >= eps must stay normal)");
    dim3 grid((unsigned)((M + 1023) / 1024), LOGLIK_ROW_SLICES), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (K <= 4) hipLaunchKernelGGL((loglik_kernel<4>), grid, block, 0, st, xp, ld, rows, M, P, Q, K, q_stride, eps, partial);
    else if (K <= 8) hipLaunchKernelGGL((loglik_kernel<8>), grid, block, 0, st, xp, ld, rows, M, P, Q, K, q_stride, eps, partial);
    else hipLaunchKernelGGL((loglik_kernel<16>), grid, block, 0, st, xp, ld, rows, M, P, Q, K, q_stride, eps, partial);
    return check_launch("loglik");
}

extern "C" int nadm_adam(float* param, const float* grad, float* m, float* v, int64_t n, int64_t clamp_from, float lr,
                         int32_t step, float grad_scale, void* stream) {
    if (!param || !grad || !m || !v) return fail("nadm_adam: null pointer");
    if (step < 1) return fail("nadm_adam: step is 1-based");
    if (n <= 0) return 0;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) return fail("nadm_adam: buffers must be 16-byte aligned");
    float step_size, inv_bc2;
    adam_scalars(lr, step, &step_size, &inv_bc2);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, clamp_from, step_size,
                       inv_bc2, grad_scale);
    return check_launch("adam");
}

extern "C" int nadm_synth_packed(uint8_t* xp, int64_t rows, int64_t row0, int64_t M, int64_t ld, const float* Qt, const float* Fq,
                                 int32_t K, float missing, uint64_t seed, void* stream) {
    if (!xp || !Qt || !Fq) return fail("nadm_synth_packed: null pointer");
    if (ld * 4 < M) return fail("nadm_synth_packed: ld < ceil(M/4)");
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid((unsigned)((ld + 255) / 256), (unsigned)nr), block(256);
        hipLaunchKernelGGL(synth_kernel, grid, block, 0, (hipStream_t)stream, xp + r0 * ld, nr, row0 + r0, M, ld, Qt + r0 * K, Fq, K, missing, seed);
    }
    return check_launch("synth_packed");
}
