// Decoder init: the means of the Gaussian mixture the reference fits in the PCA subspace (model/train.py:61-66:
//   sklearn.mixture.GaussianMixture(n_components=K, n_init=5, init_params='k-means++', tol=1e-4, covariance_type='full',
//                                   max_iter=100, random_state=seed).fit(X_pca).means_ ).
// scikit-learn (>= 1.1.0, setup.cfg:29; 1.7.2 in this image) is a third-party dependency of the reference; this file restates the
// published algorithm it runs for exactly that call -- EM for a full-covariance mixture (Dempster, Laird & Rubin 1977; Bishop, PRML
// 9.2) with the library's conventions -- in float64 on the host, for 1000-Genomes-sized inputs: there the library's fit is ~480 EM
// iterations of numpy calls on a [2504, 8] matrix, 0.55 s of a 1.9 s default run whose 250 epochs take 0.34 s, plus a 1.0 s import
// (profiles/r05_init_profile_c2.txt).  Host code only; N > 20000 runs the same algorithm in device ops (_gmm_em.py).
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
// samples into up to 16 fixed ranges that run concurrently (n_parts) -- 100k samples fit in ~0.2 s where the library takes 22-45 s.
#include "../../include/nadm.h"
#include "nadm_host.h"
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
// combined in range order -- the same bits whatever the number of hardware threads; the parts of a step run on threads of their own.
int n_parts(int64_t N) {
    const int64_t p = (N + 8191) / 8192;
    return (int)(p < 1 ? 1 : (p > 16 ? 16 : p));
}
template <typename F>
void for_parts(int P, F&& fn) {
    if (P == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int q = 1; q < P; ++q) th.emplace_back([&fn, q] { fn(q); });
    fn(0);
    for (auto& t : th) t.join();
}

struct Partial {                                                // one part's share of a step's sums
    double bound = 0.0;
    std::vector<double> nk, sx, cov;                            // [K], [K,d], [K,d,d] (upper triangle)
};

// E step on the samples [i0, i1): responsibilities -> resp, the part's sum of log-likelihoods -> bound
void e_part(const double* X, int64_t i0, int64_t i1, const Model& m, double* resp, Partial* out) {
    const int K = m.K, d = m.d;
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
        for (int k = 0; k < K; ++k) s += std::exp(lp[k] - mx);
        const double norm = mx + std::log(s);
        for (int k = 0; k < K; ++k) lp[k] = std::exp(lp[k] - norm);
        total += norm;
    }
    out->bound = total;
}

// first half of the M step on [i0, i1): nk and sum r x
void m1_part(const double* X, int64_t i0, int64_t i1, int K, int d, const double* resp, Partial* out) {
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
void m2_part(const double* X, int64_t i0, int64_t i1, const Model& m, const double* resp, Partial* out) {
    const int K = m.K, d = m.d;
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

// M step from the responsibilities (optionally fused behind the E step of the same samples: `with_e`); returns 1 on a covariance that
// is not positive definite.  *bound receives the mean log-likelihood per sample when with_e
int em_step(const double* X, int64_t N, int P, double* resp, bool with_e, bool seeding, double reg, Model* m, double* bound) {
    const int K = m->K, d = m->d;
    std::vector<Partial> part(P);
    auto range = [&](int q, int64_t* i0, int64_t* i1) { *i0 = N * q / P; *i1 = N * (q + 1) / P; };
    for_parts(P, [&](int q) {
        int64_t i0, i1;
        range(q, &i0, &i1);
        if (with_e) e_part(X, i0, i1, *m, resp, &part[q]);
        m1_part(X, i0, i1, K, d, resp, &part[q]);
    });
    std::vector<double> nk(K, 0.0);
    std::fill(m->mu.begin(), m->mu.end(), 0.0);
    double total = 0.0;
    for (int q = 0; q < P; ++q) {                               // fixed order
        total += part[q].bound;
        for (int k = 0; k < K; ++k) nk[k] += part[q].nk[k];
        for (size_t e = 0; e < m->mu.size(); ++e) m->mu[e] += part[q].sx[e];
    }
    if (with_e) *bound = total / (double)N;
    double nsum = 0.0;
    for (int k = 0; k < K; ++k) {
        nk[k] += 10.0 * std::numeric_limits<double>::epsilon();
        nsum += nk[k];
        for (int a = 0; a < d; ++a) m->mu[(size_t)k * d + a] /= nk[k];
    }
    for_parts(P, [&](int q) {
        int64_t i0, i1;
        range(q, &i0, &i1);
        m2_part(X, i0, i1, *m, resp, &part[q]);
    });
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
        if (precision_factor(c, d, &m->U[(size_t)k * d * d], &m->logdet[k])) return 1;
        m->logw[k] = std::log(seeding ? nk[k] / (double)N : nk[k] / nsum);
    }
    return 0;
}

void run_restart(const double* X, int64_t N, int d, int K, const int32_t* picks, double tol, int max_iter, double reg, Restart* out) {
    Model m{K, d, std::vector<double>((size_t)K * d), std::vector<double>((size_t)K * d * d), std::vector<double>(K), std::vector<double>(K)};
    std::vector<double> resp((size_t)N * K, 0.0);
    const int P = n_parts(N);
    for (int k = 0; k < K; ++k) resp[(size_t)picks[k] * K + k] = 1.0;
    double bound = -std::numeric_limits<double>::infinity();
    if (em_step(X, N, P, resp.data(), false, true, reg, &m, &bound)) { out->status = 1; return; }
    int it = 0;
    for (; it < max_iter; ++it) {
        const double prev = bound;
        if (em_step(X, N, P, resp.data(), true, false, reg, &m, &bound)) { out->status = 1; return; }
        if (std::fabs(bound - prev) < tol) { ++it; break; }
    }
    out->bound = bound; out->iters = it; out->means = m.mu;
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
    for (int r = 1; r < n_init; ++r) th.emplace_back(run_restart, X, N, (int)d, (int)K, picks + (int64_t)r * K, tol, (int)max_iter, reg_covar, &res[r]);
    run_restart(X, N, d, K, picks, tol, max_iter, reg_covar, &res[0]);
    for (auto& t : th) t.join();
    double top = -std::numeric_limits<double>::infinity();
    for (int r = 0; r < n_init; ++r) {
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
