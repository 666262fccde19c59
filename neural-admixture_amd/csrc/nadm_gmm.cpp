// Decoder init: the means of the Gaussian mixture the reference fits in the PCA subspace (model/train.py:61-66:
//   sklearn.mixture.GaussianMixture(n_components=K, n_init=5, init_params='k-means++', tol=1e-4, covariance_type='full',
//                                   max_iter=100, random_state=seed).fit(X_pca).means_ ).
// scikit-learn (>= 1.1.0, setup.cfg:29; 1.7.2 in this image) is a third-party dependency of the reference; this file restates the
// published algorithm it runs for exactly that call -- EM for a full-covariance mixture (Dempster, Laird & Rubin 1977; Bishop, PRML
// 9.2) with the library's conventions -- in float64 on the host.  On a 1000-Genomes-sized input the library's fit is ~480 EM iterations of
// numpy calls on a [2504, 8] matrix, 0.55 s of a 1.9 s default run whose 250 epochs take 0.34 s, plus a 1.0 s import
// (profiles/r05_init_profile_before.txt); here 0.02 s.  At N = 100k the library takes 22-45 s, this 0.8 s (tools/gmm_timing.py).  Host code
// only; _gmm_em.py holds the same algorithm in device ops (train.gmm_p_init(fit="em")).
//
//   seeding   the only random draws are the k-means++ picks of K samples per restart; the CALLER makes them (gmm.kmeanspp_picks:
//             numpy's RandomState stream consumed exactly as the library consumes it) and passes the K row indices of every restart.
//             Responsibilities start one-hot on those rows.
//   M step    nk = sum_i r_ik + 10 eps;  mu_k = sum_i r_ik x_i / nk;  S_k = sum_i r_ik (x_i - mu_k)(x_i - mu_k)^T / nk + reg I;
//             weights nk / N after the seeding step, nk / sum(nk) afterwards;  precision factor U_k = (L_k^-1)^T, L_k = chol(S_k).
//   E step    log N(x_i | k) = -(d log 2pi + |(x_i - mu_k)^T U_k|^2) / 2 + sum log diag U_k;  + log weights;  log-sum-exp over k ->
//             responsibilities;  the objective is the MEAN log-likelihood per sample, evaluated BEFORE the M step of the iteration.
//   stop      |objective change| < tol, at most max_iter iterations; of the restarts the one with the highest objective wins (the
//             first on ties), with the means as they were after its last M step.  TIES: on well-separated populations every restart
//             reaches the same optimum and the objectives agree to ~1e-15 -- in the library the winner is then decided by the rounding
//             of its sums (BLAS, thread count), i.e. which restart's component ORDER the reference reports is platform noise.  Here
//             objectives within 1e-10 of the best count as tied and the first of them wins: the rule of exact arithmetic, and the same
//             answer on every machine.
// The restarts are independent once their seeds are known: they run on threads of their own, and each splits its sums over the
// samples into up to 64 fixed ranges swept by a few worker threads (n_parts, Fit).
#include "../../include/nadm.h"
#include "nadm_host.h"
#include <atomic>
#include <cmath>
#include <limits>
#include <thread>
#include <vector>

using namespace nadm;

namespace {

constexpr int GMM_MAX_D = 32;
constexpr double GMM_TIE = 1e-10;                               // objectives (mean log-likelihood per sample) closer than this are one optimum

struct Restart {
    double bound = -std::numeric_limits<double>::infinity();
    int iters = 0, status = 0;                                  // status 1: a covariance was not positive definite
    std::vector<double> means;
};

struct Model {
    int K, d;
    std::vector<double> mu, U, logdet, logw;                    // [K,d], [K,d,d] (upper triangular, row-major), [K], [K]
};

// S [d,d] symmetric -> U = (L^-1)^T with L = chol(S) lower; returns 1 if S is not positive definite
int precision_factor(const double* S, int d, double* U, double* logdet) {
    double L[GMM_MAX_D * GMM_MAX_D] = {0.0}, Li[GMM_MAX_D * GMM_MAX_D] = {0.0};
    for (int j = 0; j < d; ++j) {
        double s = S[j * d + j];
        for (int p = 0; p < j; ++p) s -= L[j * d + p] * L[j * d + p];
        if (!(s > 0.0) || !std::isfinite(s)) return 1;
        const double ljj = std::sqrt(s);
        L[j * d + j] = ljj;
        for (int i = j + 1; i < d; ++i) {
            double t = S[i * d + j];
            for (int p = 0; p < j; ++p) t -= L[i * d + p] * L[j * d + p];
            L[i * d + j] = t / ljj;
        }
    }
    for (int c = 0; c < d; ++c) {                               // L Li = I, column by column (forward substitution)
        for (int i = c; i < d; ++i) {
            double t = i == c ? 1.0 : 0.0;
            for (int p = c; p < i; ++p) t -= L[i * d + p] * Li[p * d + c];
            Li[i * d + c] = t / L[i * d + i];
        }
    }
    double ld = 0.0;
    for (int a = 0; a < d; ++a) {
        for (int b = 0; b < d; ++b) U[a * d + b] = b >= a ? Li[b * d + a] : 0.0;
        ld += std::log(U[a * d + a]);
    }
    *logdet = ld;
    return 0;
}

// The sums over samples run over P fixed sample ranges ("parts": P depends on N only, never on the machine) whose partial sums are
// combined in range order -- the same bits whatever the number of hardware threads.  A restart keeps W worker threads for its lifetime
// (at most 8, see nadm_gmm_fit_means); a step is two sweeps over the parts with the combination in between,
// sequenced by a spinning barrier (a thread per sweep costs more than the sweep: 49 parts x 200 sweeps x 5 restarts at N = 100k).
int n_parts(int64_t N) {
    const int64_t p = (N + 2047) / 2048;
    return (int)(p < 1 ? 1 : (p > 64 ? 64 : p));
}

struct Barrier {
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        if (n == 1) return;
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
            return;
        }
        for (int spins = 0; gen.load(std::memory_order_acquire) == g; ++spins) {
            if (spins < 1024) __builtin_ia32_pause();
            else std::this_thread::yield();
        }
    }
    const int n;
    std::atomic<int> count{0}, gen{0};
};

struct Partial {                                                // one part's share of a step's sums
    double bound = 0.0;
    std::vector<double> nk, sx, cov;                            // [K], [K,d], [K,d,d] (upper triangle)
};

// E step on the samples [i0, i1): responsibilities -> resp, the part's sum of log-likelihoods -> bound
template <int D>                                                // D = the dimension at compile time (0: read it from the model)
void e_part_d(const double* X, int64_t i0, int64_t i1, const Model& m, double* resp, Partial* out) {
    const int K = m.K, d = D ? D : m.d;
    const double c0 = (double)d * std::log(2.0 * M_PI);
    double total = 0.0;
    for (int64_t i = i0; i < i1; ++i) {
        const double* x = X + i * d;
        double* lp = resp + i * K;
        double mx = -std::numeric_limits<double>::infinity();
        for (int k = 0; k < K; ++k) {
            const double* mu = &m.mu[(size_t)k * d];
            const double* U = &m.U[(size_t)k * d * d];
            double df[GMM_MAX_D];
            for (int a = 0; a < d; ++a) df[a] = x[a] - mu[a];
            double maha = 0.0;
            for (int b = 0; b < d; ++b) {
                double y = 0.0;
                for (int a = 0; a <= b; ++a) y += df[a] * U[a * d + b];
                maha += y * y;
            }
            lp[k] = -0.5 * (c0 + maha) + m.logdet[k] + m.logw[k];
            if (lp[k] > mx) mx = lp[k];
        }
        double s = 0.0;
        for (int k = 0; k < K; ++k) { lp[k] = std::exp(lp[k] - mx); s += lp[k]; }
        const double inv = 1.0 / s;
        for (int k = 0; k < K; ++k) lp[k] *= inv;                // = exp(lp - norm) up to an ulp, for half the exponentials
        total += mx + std::log(s);
    }
    out->bound = total;
}

// first half of the M step on [i0, i1): nk and sum r x
template <int D>
void m1_part_d(const double* X, int64_t i0, int64_t i1, int K, int d_, const double* resp, Partial* out) {
    const int d = D ? D : d_;
    out->nk.assign(K, 0.0);
    out->sx.assign((size_t)K * d, 0.0);
    for (int64_t i = i0; i < i1; ++i) {
        const double* x = X + i * d;
        const double* r = resp + i * K;
        for (int k = 0; k < K; ++k) {
            const double rk = r[k];
            if (rk == 0.0) continue;
            out->nk[k] += rk;
            double* sx = &out->sx[(size_t)k * d];
            for (int a = 0; a < d; ++a) sx[a] += rk * x[a];
        }
    }
}

// second half on [i0, i1): sum r (x - mu)(x - mu)^T, upper triangle
template <int D>
void m2_part_d(const double* X, int64_t i0, int64_t i1, const Model& m, const double* resp, Partial* out) {
    const int K = m.K, d = D ? D : m.d;
    out->cov.assign((size_t)K * d * d, 0.0);
    for (int64_t i = i0; i < i1; ++i) {
        const double* x = X + i * d;
        const double* r = resp + i * K;
        for (int k = 0; k < K; ++k) {
            const double rk = r[k];
            if (rk == 0.0) continue;
            const double* mu = &m.mu[(size_t)k * d];
            double df[GMM_MAX_D];
            for (int a = 0; a < d; ++a) df[a] = x[a] - mu[a];
            double* c = &out->cov[(size_t)k * d * d];
            for (int a = 0; a < d; ++a) {
                const double w = rk * df[a];
                for (int b = a; b < d; ++b) c[a * d + b] += w * df[b];
            }
        }
    }
}

// (n_components = 8 is the reference's default, entry.py:33: its loops are unrolled and vectorised at compile time)
void e_part(const double* X, int64_t i0, int64_t i1, const Model& m, double* resp, Partial* out) {
    if (m.d == 8) e_part_d<8>(X, i0, i1, m, resp, out); else e_part_d<0>(X, i0, i1, m, resp, out);
}
void m1_part(const double* X, int64_t i0, int64_t i1, int K, int d, const double* resp, Partial* out) {
    if (d == 8) m1_part_d<8>(X, i0, i1, K, d, resp, out); else m1_part_d<0>(X, i0, i1, K, d, resp, out);
}
void m2_part(const double* X, int64_t i0, int64_t i1, const Model& m, const double* resp, Partial* out) {
    if (m.d == 8) m2_part_d<8>(X, i0, i1, m, resp, out); else m2_part_d<0>(X, i0, i1, m, resp, out);
}

// One restart: the model, the responsibilities, the parts' partial sums and the worker threads that sweep them
struct Fit {
    const double* X;
    int64_t N;
    int P, W;
    Model m;
    std::vector<double> resp;
    std::vector<Partial> part;
    Barrier bar;
    bool with_e = false, stop = false;                          // published by the main thread before it releases the workers
    std::vector<std::thread> workers;

    Fit(const double* X_, int64_t N_, int d, int K, int W_)
        : X(X_), N(N_), P(n_parts(N_)), W(W_ < 1 ? 1 : (W_ > n_parts(N_) ? n_parts(N_) : W_)),
          m{K, d, std::vector<double>((size_t)K * d), std::vector<double>((size_t)K * d * d), std::vector<double>(K), std::vector<double>(K)},
          resp((size_t)N_ * K, 0.0), part(n_parts(N_)), bar(W) {
        for (int w = 1; w < W; ++w) workers.emplace_back([this, w] { work(w); });
    }
    ~Fit() {
        stop = true;
        bar.wait();
        for (auto& t : workers) t.join();
    }
    void range(int q, int64_t* i0, int64_t* i1) const { *i0 = N * q / P; *i1 = N * (q + 1) / P; }
    void sweep_a(int w) {                                       // [E step +] nk, sum r x of this worker's parts
        for (int q = w; q < P; q += W) {
            int64_t i0, i1;
            range(q, &i0, &i1);
            if (with_e) e_part(X, i0, i1, m, resp.data(), &part[q]);
            m1_part(X, i0, i1, m.K, m.d, resp.data(), &part[q]);
        }
    }
    void sweep_b(int w) {                                       // sum r (x - mu)(x - mu)^T
        for (int q = w; q < P; q += W) {
            int64_t i0, i1;
            range(q, &i0, &i1);
            m2_part(X, i0, i1, m, resp.data(), &part[q]);
        }
    }
    void work(int w) {
        for (;;) {
            bar.wait();                                         // a step begins (or the fit ends)
            if (stop) return;
            sweep_a(w);
            bar.wait();                                         // every part's first sums are in
            bar.wait();                                         // the means are combined
            sweep_b(w);
            bar.wait();                                         // every part's covariance sums are in
        }
    }
    // M step from the responsibilities, optionally behind the E step of the same samples; returns 1 on a covariance that is not positive
    // definite.  *bound receives the mean log-likelihood per sample when e_first
    int step(bool e_first, bool seeding, double reg, double* bound) {
        const int K = m.K, d = m.d;
        with_e = e_first;
        bar.wait();
        sweep_a(0);
        bar.wait();
        std::vector<double> nk(K, 0.0);
        std::fill(m.mu.begin(), m.mu.end(), 0.0);
        double total = 0.0;
        for (int q = 0; q < P; ++q) {                           // fixed order
            total += part[q].bound;
            for (int k = 0; k < K; ++k) nk[k] += part[q].nk[k];
            for (size_t e = 0; e < m.mu.size(); ++e) m.mu[e] += part[q].sx[e];
        }
        if (e_first) *bound = total / (double)N;
        double nsum = 0.0;
        for (int k = 0; k < K; ++k) {
            nk[k] += 10.0 * std::numeric_limits<double>::epsilon();
            nsum += nk[k];
            for (int a = 0; a < d; ++a) m.mu[(size_t)k * d + a] /= nk[k];
        }
        bar.wait();
        sweep_b(0);
        bar.wait();
        std::vector<double> cov((size_t)K * d * d, 0.0);
        for (int q = 0; q < P; ++q)
            for (size_t e = 0; e < cov.size(); ++e) cov[e] += part[q].cov[e];
        for (int k = 0; k < K; ++k) {
            double* c = &cov[(size_t)k * d * d];
            for (int a = 0; a < d; ++a)
                for (int b = a; b < d; ++b) {
                    const double v = c[a * d + b] / nk[k] + (a == b ? reg : 0.0);
                    c[a * d + b] = v; c[b * d + a] = v;
                }
            if (precision_factor(c, d, &m.U[(size_t)k * d * d], &m.logdet[k])) return 1;
            m.logw[k] = std::log(seeding ? nk[k] / (double)N : nk[k] / nsum);
        }
        return 0;
    }
};

void run_restart(const double* X, int64_t N, int d, int K, const int32_t* picks, double tol, int max_iter, double reg, int workers, Restart* out) try {
    Fit f(X, N, d, K, workers);
    for (int k = 0; k < K; ++k) f.resp[(size_t)picks[k] * K + k] = 1.0;
    double bound = -std::numeric_limits<double>::infinity();
    if (f.step(false, true, reg, &bound)) { out->status = 1; return; }
    int it = 0;
    for (; it < max_iter; ++it) {
        const double prev = bound;
        if (f.step(true, false, reg, &bound)) { out->status = 1; return; }
        if (std::fabs(bound - prev) < tol) { ++it; break; }
    }
    out->bound = bound; out->iters = it; out->means = f.m.mu;
} catch (...) {                                                 // (no memory, no thread: reported by the caller, never thrown through the C ABI)
    out->status = 2;
}

}  // namespace

extern "C" int nadm_gmm_fit_means(const double* X, int64_t N, int32_t d, int32_t K, const int32_t* picks, int32_t n_init, double tol,
                                  int32_t max_iter, double reg_covar, double* means, double* lower_bound, int32_t* n_iter) {
    if (!X || !picks || !means) return fail("nadm_gmm_fit_means: null pointer");
    if (N < 1 || d < 1 || d > GMM_MAX_D || K < 1 || K > N || n_init < 1 || max_iter < 1) return fail("nadm_gmm_fit_means: need 1 <= d <= 32, 1 <= K <= N, n_init, max_iter >= 1");
    for (int64_t j = 0; j < (int64_t)n_init * K; ++j)
        if (picks[j] < 0 || picks[j] >= N) return fail("nadm_gmm_fit_means: a seed index lies outside the samples");
    std::vector<Restart> res(n_init);
    std::vector<std::thread> th;
    try {
    // worker threads per restart: the restarts run side by side, and the barriers spin -- never more threads than the host has
    // (at most 8 each: the parts are ~2048 samples, a sweep is tens of microseconds per part, and a GPU host is shared -- 5 x 8 threads
    // is the measured sweet spot on a 256-thread host: 51 each spent their time spinning on each other's hyperthreads)
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    int workers = hw / (2 * n_init);
    if (workers > 8) workers = 8;
    for (int r = 1; r < n_init; ++r) th.emplace_back(run_restart, X, N, (int)d, (int)K, picks + (int64_t)r * K, tol, (int)max_iter, reg_covar, workers, &res[r]);
    run_restart(X, N, d, K, picks, tol, max_iter, reg_covar, workers, &res[0]);
    } catch (...) {                                             // a restart's thread could not be started: the others are joined, the fit fails
        for (auto& t : th) t.join();
        return fail("nadm_gmm_fit_means: out of memory or threads");
    }
    for (auto& t : th) t.join();
    double top = -std::numeric_limits<double>::infinity();
    for (int r = 0; r < n_init; ++r) {
        if (res[r].status == 2) return fail("nadm_gmm_fit_means: out of memory or threads");
        if (res[r].status)        // (the library raises from inside the restart that hits it; here every restart has run, the outcome is the same)
            return fail("Fitting the mixture model failed because some components have ill-defined empirical covariance (for instance caused by "
                        "singleton or collapsed samples). Try to decrease the number of components, increase reg_covar, or scale the input data.");
        if (res[r].bound > top) top = res[r].bound;
    }
    int best = 0;
    while (best < n_init - 1 && !(res[best].bound >= top - GMM_TIE)) ++best;      // the first restart within GMM_TIE of the best objective
    memcpy(means, res[best].means.data(), sizeof(double) * (size_t)K * d);
    if (lower_bound) *lower_bound = res[best].bound;
    if (n_iter) *n_iter = res[best].iters;
    return 0;
}
