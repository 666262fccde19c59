// Decoder init on the GPU: the EM iterations of csrc/nadm_gmm.cpp (the reference's GaussianMixture call, model/train.py:61-66; the
// algorithm and the library's conventions are stated there) with the sums over the samples on the device.  Same arithmetic, float64,
// the restarts side by side in one grid; only the ORDER of the sums differs from the host form (threads stride a part's samples, a
// block adds its threads in a fixed tree, the parts are added in part order), so a fit is reproducible on the device and equal to
// the host's to rounding (tests: means to 1e-9, same iteration counts).
//
// Why: at N = 100k the host form is 0.77 s of a default run's 11 s (5 restarts x 100 iterations x 2 sweeps on 40 threads,
// profiles/r05_gmm_timing.txt); an iteration here is five small launches, ~35 us, and nothing returns to the host but the five
// `done` flags every fourth iteration.
//
//   E      grid (N / 1024, R)       log N(x_i | k) + log w_k -> responsibilities (global, [R, N, K]); the block's sum of log-likelihoods
//   M1     grid (P, K, R)           nk and sum r x of part q for component k
//   fin1   grid (R)                 parts in order -> nk (+ 10 eps), mu; the objective (mean log-likelihood) of the E step
//   M2     grid (P, K, R)           sum r (x - mu)(x - mu)^T, upper triangle
//   fin2   grid (R)                 parts in order -> covariance / nk + reg I -> U = (chol^-1)^T, log det, log w; convergence: the restart
//                                   is done when |objective change| < tol or max_iter E steps have run; done restarts skip every launch
// d = 8 (the reference's --pca_components default, entry.py:33) and K <= 16; gmm.fit_means keeps the host form for anything else.
#include "../../include/nadm.h"
#include "nadm_common.h"
#include "nadm_host.h"
#include <limits>
#include <vector>

namespace nadm {
namespace {

constexpr int GD = 8, GK = 16, GT = 256, GSPT = 4, GU = GD * (GD + 1) / 2;
constexpr double GMM_TIE_DEV = 1e-10;                             // as on the host: objectives closer than this are one optimum

struct GmmDev {
    const double* X;                                              // [N, GD]
    int64_t N;
    int K, R, P, PE, max_iter;
    double tol, reg;
    double *resp, *mu, *U, *logdet, *logw, *nk;                   // [R,N,K], [R,K,GD], [R,K,GD,GD] upper, [R,K], [R,K], [R,K]
    double *pb, *pnk, *psx, *pcov, *newbound, *bound;             // [R,PE], [R,P,K], [R,P,K,GD], [R,P,K,GU], [R], [R]
    int *done, *iters, *status;                                   // [R]
    const int32_t* picks;                                         // [R,K]
};

// NV sums over the block's 256 threads, fixed order: lanes by xor-shuffle, the four waves left to right
template <int NV>
__device__ __forceinline__ void block_sums(const double (&v)[NV], double (*s_red)[GT / 64], double* out_if_tid_lt_nv) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const double w = wave_sum_all_f64(v[j]);
        if ((tid & 63) == 0) s_red[j][tid >> 6] = w;
    }
    __syncthreads();
    if (tid < NV) *out_if_tid_lt_nv = (s_red[tid][0] + s_red[tid][1]) + (s_red[tid][2] + s_red[tid][3]);
}

__global__ void gmm_seed_kernel(GmmDev s) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= s.R * s.K) return;
    const int r = t / s.K, k = t % s.K;
    s.resp[((int64_t)r * s.N + s.picks[t]) * s.K + k] = 1.0;
}

__global__ __launch_bounds__(GT) void gmm_e_kernel(GmmDev s) {
    const int r = blockIdx.y, tid = threadIdx.x, K = s.K;
    if (s.done[r]) return;
    __shared__ double s_mu[GK * GD], s_U[GK * GD * GD], s_c[GK];
    __shared__ double s_red[1][GT / 64];
    for (int e = tid; e < K * GD; e += GT) s_mu[e] = s.mu[(int64_t)r * K * GD + e];
    for (int e = tid; e < K * GD * GD; e += GT) s_U[e] = s.U[(int64_t)r * K * GD * GD + e];
    if (tid < K) s_c[tid] = s.logdet[r * K + tid] + s.logw[r * K + tid];
    __syncthreads();
    const double c0 = (double)GD * 1.8378770664093453;            // d log 2 pi
    double total[1] = {0.0};
    for (int j = 0; j < GSPT; ++j) {
        const int64_t i = ((int64_t)blockIdx.x * GSPT + j) * GT + tid;
        if (i >= s.N) continue;
        double x[GD], lp[GK];
#pragma unroll
        for (int a = 0; a < GD; ++a) x[a] = s.X[i * GD + a];
        double mx = -__builtin_inf();
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            lp[k] = -__builtin_inf();
            if (k < K) {
                double df[GD];
#pragma unroll
                for (int a = 0; a < GD; ++a) df[a] = x[a] - s_mu[k * GD + a];
                double maha = 0.0;
#pragma unroll
                for (int b = 0; b < GD; ++b) {
                    double y = 0.0;
#pragma unroll
                    for (int a = 0; a <= b; ++a) y += df[a] * s_U[(k * GD + a) * GD + b];
                    maha += y * y;
                }
                lp[k] = -0.5 * (c0 + maha) + s_c[k];
                if (lp[k] > mx) mx = lp[k];
            }
        }
        double sm = 0.0;
#pragma unroll
        for (int k = 0; k < GK; ++k)
            if (k < K) { lp[k] = exp(lp[k] - mx); sm += lp[k]; }
        const double inv = 1.0 / sm;
        double* out = s.resp + ((int64_t)r * s.N + i) * K;
#pragma unroll
        for (int k = 0; k < GK; ++k)
            if (k < K) out[k] = lp[k] * inv;
        total[0] += mx + log(sm);
    }
    double v;
    block_sums<1>(total, s_red, &v);
    if (tid == 0) s.pb[(int64_t)r * s.PE + blockIdx.x] = v;
}

__global__ __launch_bounds__(GT) void gmm_m1_kernel(GmmDev s) {
    const int q = blockIdx.x, k = blockIdx.y, r = blockIdx.z, tid = threadIdx.x, K = s.K;
    if (s.done[r]) return;
    __shared__ double s_red[1 + GD][GT / 64];
    const int64_t i0 = s.N * q / s.P, i1 = s.N * (q + 1) / s.P;
    double acc[1 + GD];
#pragma unroll
    for (int a = 0; a <= GD; ++a) acc[a] = 0.0;
    for (int64_t i = i0 + tid; i < i1; i += GT) {
        const double rk = s.resp[((int64_t)r * s.N + i) * K + k];
        if (rk == 0.0) continue;
        acc[0] += rk;
#pragma unroll
        for (int a = 0; a < GD; ++a) acc[1 + a] += rk * s.X[i * GD + a];
    }
    double v = 0.0;
    block_sums<1 + GD>(acc, s_red, &v);
    const int64_t o = ((int64_t)r * s.P + q) * K + k;
    if (tid == 0) s.pnk[o] = v;
    else if (tid <= GD) s.psx[o * GD + tid - 1] = v;
}

__global__ __launch_bounds__(GT) void gmm_fin1_kernel(GmmDev s, int e_first) {
    const int r = blockIdx.x, tid = threadIdx.x, K = s.K;
    if (s.done[r]) return;
    __shared__ double s_nk[GK];
    if (tid < K) {
        double n = 0.0;
        for (int q = 0; q < s.P; ++q) n += s.pnk[((int64_t)r * s.P + q) * K + tid];
        n += 10.0 * 2.220446049250313e-16;
        s_nk[tid] = n;
        s.nk[r * K + tid] = n;
    }
    __syncthreads();
    if (tid < K * GD) {
        const int k = tid / GD, a = tid % GD;
        double m = 0.0;
        for (int q = 0; q < s.P; ++q) m += s.psx[(((int64_t)r * s.P + q) * K + k) * GD + a];
        s.mu[((int64_t)r * K + k) * GD + a] = m / s_nk[k];
    }
    if (e_first && tid == GT - 1) {
        double t = 0.0;
        for (int b = 0; b < s.PE; ++b) t += s.pb[(int64_t)r * s.PE + b];
        s.newbound[r] = t / (double)s.N;
    }
}

__global__ __launch_bounds__(GT) void gmm_m2_kernel(GmmDev s) {
    const int q = blockIdx.x, k = blockIdx.y, r = blockIdx.z, tid = threadIdx.x, K = s.K;
    if (s.done[r]) return;
    __shared__ double s_red[GU][GT / 64];
    const int64_t i0 = s.N * q / s.P, i1 = s.N * (q + 1) / s.P;
    double mu[GD], acc[GU];
#pragma unroll
    for (int a = 0; a < GD; ++a) mu[a] = s.mu[((int64_t)r * K + k) * GD + a];
#pragma unroll
    for (int e = 0; e < GU; ++e) acc[e] = 0.0;
    for (int64_t i = i0 + tid; i < i1; i += GT) {
        const double rk = s.resp[((int64_t)r * s.N + i) * K + k];
        if (rk == 0.0) continue;
        double df[GD];
#pragma unroll
        for (int a = 0; a < GD; ++a) df[a] = s.X[i * GD + a] - mu[a];
        int e = 0;
#pragma unroll
        for (int a = 0; a < GD; ++a) {
            const double w = rk * df[a];
#pragma unroll
            for (int b = a; b < GD; ++b) acc[e++] += w * df[b];
        }
    }
    double v = 0.0;
    block_sums<GU>(acc, s_red, &v);
    if (tid < GU) s.pcov[(((int64_t)r * s.P + q) * K + k) * GU + tid] = v;
}

__global__ __launch_bounds__(GT) void gmm_fin2_kernel(GmmDev s, int e_first, int seeding) {
    const int r = blockIdx.x, tid = threadIdx.x, K = s.K;
    if (s.done[r]) return;
    __shared__ double s_S[GK][GD * GD], s_L[GK][GD * GD], s_Li[GK][GD * GD];
    __shared__ double s_nsum;
    __shared__ int s_bad;
    if (tid == 0) {
        double n = 0.0;
        for (int k = 0; k < K; ++k) n += s.nk[r * K + k];
        s_nsum = n;
        s_bad = 0;
    }
    for (int t = tid; t < K * GU; t += GT) {                     // covariance sums: the parts in order, then / nk + reg on the diagonal
        const int k = t / GU, e = t % GU;
        double c = 0.0;
        for (int q = 0; q < s.P; ++q) c += s.pcov[(((int64_t)r * s.P + q) * K + k) * GU + e];
        int a = 0, rem = e;
        while (rem >= GD - a) { rem -= GD - a; ++a; }
        const int b = a + rem;
        const double v = c / s.nk[r * K + k] + (a == b ? s.reg : 0.0);
        s_S[k][a * GD + b] = v;
        s_S[k][b * GD + a] = v;
    }
    for (int t = tid; t < K * GD * GD; t += GT) { s_L[t / (GD * GD)][t % (GD * GD)] = 0.0; s_Li[t / (GD * GD)][t % (GD * GD)] = 0.0; }
    __syncthreads();
    if (tid < K) {                                               // U = (L^-1)^T, L = chol(S): one thread per component (8 x 8)
        const int k = tid;
        double* S = s_S[k];
        double* L = s_L[k];
        double* Li = s_Li[k];
        bool bad = false;
        for (int j = 0; j < GD && !bad; ++j) {
            double d = S[j * GD + j];
            for (int p = 0; p < j; ++p) d -= L[j * GD + p] * L[j * GD + p];
            if (!(d > 0.0) || !isfinite(d)) { bad = true; break; }
            const double ljj = sqrt(d);
            L[j * GD + j] = ljj;
            for (int i = j + 1; i < GD; ++i) {
                double t = S[i * GD + j];
                for (int p = 0; p < j; ++p) t -= L[i * GD + p] * L[j * GD + p];
                L[i * GD + j] = t / ljj;
            }
        }
        if (bad) atomicOr(&s_bad, 1);
        else {
            for (int c = 0; c < GD; ++c)
                for (int i = c; i < GD; ++i) {
                    double t = i == c ? 1.0 : 0.0;
                    for (int p = c; p < i; ++p) t -= L[i * GD + p] * Li[p * GD + c];
                    Li[i * GD + c] = t / L[i * GD + i];
                }
            double ld = 0.0;
            double* U = s.U + ((int64_t)r * K + k) * GD * GD;
            for (int a = 0; a < GD; ++a) {
                for (int b = 0; b < GD; ++b) U[a * GD + b] = b >= a ? Li[b * GD + a] : 0.0;
                ld += log(Li[a * GD + a]);
            }
            s.logdet[r * K + k] = ld;
            const double n = s.nk[r * K + k];
            s.logw[r * K + k] = log(seeding ? n / (double)s.N : n / s_nsum);
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_bad) { s.status[r] = 1; s.done[r] = 1; return; }
        if (e_first) {
            const double prev = s.bound[r], now = s.newbound[r];
            s.bound[r] = now;
            const int it = s.iters[r] + 1;
            s.iters[r] = it;
            if (fabs(now - prev) < s.tol || it >= s.max_iter) s.done[r] = 1;
        }
    }
}

#define GMM_HIP(call)                                                                                        \
    do {                                                                                                     \
        const hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                              \
            snprintf(err_buf(), 512, "nadm_gmm_fit_means_dev: %s: %s", #call, hipGetErrorString(e_));        \
            rc = 2;                                                                                          \
            goto out;                                                                                        \
        }                                                                                                    \
    } while (0)

}  // namespace
}  // namespace nadm

using namespace nadm;

extern "C" int nadm_gmm_fit_means_dev(const double* X, int64_t N, int32_t d, int32_t K, const int32_t* picks, int32_t n_init, double tol,
                                      int32_t max_iter, double reg_covar, double* means, double* lower_bound, int32_t* n_iter, void* stream) {
    if (!X || !picks || !means) return fail("nadm_gmm_fit_means_dev: null pointer");
    if (d != GD || K < 1 || K > GK) return fail("nadm_gmm_fit_means_dev: d must be 8 and K in 1..16 (nadm_gmm_fit_means takes the rest)");
    if (N < 1 || K > N || n_init < 1 || n_init > 64 || max_iter < 1) return fail("nadm_gmm_fit_means_dev: need 1 <= K <= N, 1 <= n_init <= 64, max_iter >= 1");
    for (int64_t j = 0; j < (int64_t)n_init * K; ++j)
        if (picks[j] < 0 || picks[j] >= N) return fail("nadm_gmm_fit_means_dev: a seed index lies outside the samples");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int R = n_init;
    int64_t Pl = (N + 2047) / 2048;
    const int P = (int)(Pl < 1 ? 1 : (Pl > 256 ? 256 : Pl));
    const int PE = (int)((N + (int64_t)GT * GSPT - 1) / ((int64_t)GT * GSPT));
    // one arena: doubles first, then the ints
    const size_t n_x = (size_t)N * GD, n_resp = (size_t)R * N * K, n_mu = (size_t)R * K * GD, n_U = (size_t)R * K * GD * GD, n_rk = (size_t)R * K;
    const size_t n_pb = (size_t)R * PE, n_pnk = (size_t)R * P * K, n_psx = n_pnk * GD, n_pcov = n_pnk * GU;
    const size_t n_dbl = n_x + n_resp + n_mu + n_U + 3 * n_rk + n_pb + n_pnk + n_psx + n_pcov + 2 * (size_t)R;
    const size_t n_int = 3 * (size_t)R + (size_t)R * K;
    char* arena = nullptr;
    int rc = 0;
    std::vector<int> h_int(3 * (size_t)R, 0);
    std::vector<double> h_bound(R), h_mu(n_mu);
    GmmDev s{};
    double* p = nullptr;
    int* ip = nullptr;
    GMM_HIP(hipMalloc((void**)&arena, n_dbl * sizeof(double) + n_int * sizeof(int)));
    p = reinterpret_cast<double*>(arena);
    s.N = N; s.K = K; s.R = R; s.P = P; s.PE = PE; s.max_iter = max_iter; s.tol = tol; s.reg = reg_covar;
    s.X = p; p += n_x;
    s.resp = p; p += n_resp;
    s.mu = p; p += n_mu;
    s.U = p; p += n_U;
    s.logdet = p; p += n_rk;
    s.logw = p; p += n_rk;
    s.nk = p; p += n_rk;
    s.pb = p; p += n_pb;
    s.pnk = p; p += n_pnk;
    s.psx = p; p += n_psx;
    s.pcov = p; p += n_pcov;
    s.newbound = p; p += R;
    s.bound = p; p += R;
    ip = reinterpret_cast<int*>(p);
    s.done = ip; s.iters = ip + R; s.status = ip + 2 * R;
    s.picks = ip + 3 * R;
    GMM_HIP(hipMemcpyAsync(const_cast<double*>(s.X), X, n_x * sizeof(double), hipMemcpyHostToDevice, st));
    GMM_HIP(hipMemcpyAsync(const_cast<int32_t*>(s.picks), picks, (size_t)R * K * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GMM_HIP(hipMemsetAsync(s.resp, 0, n_resp * sizeof(double), st));
    GMM_HIP(hipMemsetAsync(s.done, 0, 3 * (size_t)R * sizeof(int), st));
    for (int r = 0; r < R; ++r) h_bound[r] = -std::numeric_limits<double>::infinity();
    GMM_HIP(hipMemcpyAsync(s.bound, h_bound.data(), R * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gmm_seed_kernel, dim3((R * K + 63) / 64), dim3(64), 0, st, s);
    {
        const dim3 gm(P, K, R), ge(PE, R), gr(R), blk(GT);
        // the M step behind the seeding (responsibilities one-hot on the picked rows, weights nk / N)
        hipLaunchKernelGGL(gmm_m1_kernel, gm, blk, 0, st, s);
        hipLaunchKernelGGL(gmm_fin1_kernel, gr, blk, 0, st, s, 0);
        hipLaunchKernelGGL(gmm_m2_kernel, gm, blk, 0, st, s);
        hipLaunchKernelGGL(gmm_fin2_kernel, gr, blk, 0, st, s, 0, 1);
        for (int it = 0; it < max_iter; ++it) {
            hipLaunchKernelGGL(gmm_e_kernel, ge, blk, 0, st, s);
            hipLaunchKernelGGL(gmm_m1_kernel, gm, blk, 0, st, s);
            hipLaunchKernelGGL(gmm_fin1_kernel, gr, blk, 0, st, s, 1);
            hipLaunchKernelGGL(gmm_m2_kernel, gm, blk, 0, st, s);
            hipLaunchKernelGGL(gmm_fin2_kernel, gr, blk, 0, st, s, 1, 0);
            if ((it & 3) == 3 && it + 1 < max_iter) {           // every fourth iteration: has every restart stopped?
                GMM_HIP(hipMemcpyAsync(h_int.data(), s.done, R * sizeof(int), hipMemcpyDeviceToHost, st));
                GMM_HIP(hipStreamSynchronize(st));
                bool all = true;
                for (int r = 0; r < R; ++r) all = all && h_int[r] != 0;
                if (all) break;
            }
        }
    }
    if ((rc = check_launch("gmm_fit_means_dev"))) goto out;
    GMM_HIP(hipMemcpyAsync(h_int.data(), s.done, 3 * (size_t)R * sizeof(int), hipMemcpyDeviceToHost, st));
    GMM_HIP(hipMemcpyAsync(h_bound.data(), s.bound, R * sizeof(double), hipMemcpyDeviceToHost, st));
    GMM_HIP(hipMemcpyAsync(h_mu.data(), s.mu, n_mu * sizeof(double), hipMemcpyDeviceToHost, st));
    GMM_HIP(hipStreamSynchronize(st));
    {
        double top = -std::numeric_limits<double>::infinity();
        for (int r = 0; r < R; ++r) {
            if (h_int[2 * R + r]) {
                rc = fail("Fitting the mixture model failed because some components have ill-defined empirical covariance (for instance caused by "
                          "singleton or collapsed samples). Try to decrease the number of components, increase reg_covar, or scale the input data.");
                goto out;
            }
            if (h_bound[r] > top) top = h_bound[r];
        }
        int best = 0;
        while (best < R - 1 && !(h_bound[best] >= top - GMM_TIE_DEV)) ++best;       // the first restart within the tie band of the best objective
        memcpy(means, h_mu.data() + (size_t)best * K * GD, sizeof(double) * (size_t)K * GD);
        if (lower_bound) *lower_bound = h_bound[best];
        if (n_iter) *n_iter = h_int[R + best];
    }
out:
    if (arena) (void)hipFree(arena);
    return rc;
}
