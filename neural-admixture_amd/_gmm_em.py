"""Gaussian-mixture means for the decoder init (reference: model/train.py:61-66 calls
``sklearn.mixture.GaussianMixture(n_components=K, n_init=5, init_params='k-means++', tol=1e-4, covariance_type='full',
max_iter=100, random_state=seed).fit(X_pca).means_``).

scikit-learn (1.7.2 in this image) is a third-party dependency of the reference; this module restates the published
algorithm it runs for exactly that call -- EM for a full-covariance mixture (Dempster et al. 1977; Bishop PRML 9.2) with
scikit-learn's conventions -- in float64 tensor ops on the device that holds the projected samples, because on 100k
samples the library's per-component numpy loops take 22-45 s while the whole training run takes 14 s:

  * seeding: the ONLY random draws are k-means++ (Arthur & Vassilvitskii 2007) picks of K samples per restart, taken from
    ONE RandomState(seed) stream shared by the 5 restarts; they are made here by calling ``sklearn.cluster.kmeans_plusplus``
    itself, so the restarts start from the library's own picks.  Responsibilities start one-hot on those K samples.
  * M step: nk = sum_i r_ik + 10*eps, mu_k = sum_i r_ik x_i / nk, S_k = sum_i r_ik (x_i-mu_k)(x_i-mu_k)^T / nk + 1e-6*I,
    weights nk / sum(nk) (after the seeding step: nk / N, not renormalised), precision factor = (L_k^-1)^T, L_k = chol(S_k).
  * E step: log N(x_i | k) from the precision factors, + log weights, log-sum-exp over k -> log r_ik; the objective is the
    MEAN log-likelihood per sample, evaluated BEFORE the M step of the same iteration.
  * stop when |objective change| < tol, at most max_iter iterations; of the restarts keep the one with the highest
    objective (first wins ties), with the parameters as they were after its last M step.

The K tiny Cholesky factorisations run on the host with the same LAPACK calls the library uses; everything that touches
the N samples is a device op.  ``tests/test_abi_and_host.py`` checks the means against the library's on the CPU and
``tests/test_gpu_parity.py`` on the GPU; ``gmm_p_init(..., fit="sklearn")`` selects the library fit (_gmm_fit.py) instead."""
from __future__ import annotations

import math

import numpy as np
import torch


def _precision_factors(cov: np.ndarray) -> np.ndarray:
    from scipy import linalg
    K, d, _ = cov.shape
    out = np.empty_like(cov)
    for k in range(K):
        try:
            L = linalg.cholesky(cov[k], lower=True)
        except linalg.LinAlgError:
            raise ValueError("Fitting the mixture model failed because some components have ill-defined empirical "
                             "covariance (for instance caused by singleton or collapsed samples).")
        out[k] = linalg.solve_triangular(L, np.eye(d), lower=True).T
    return out


class _Mixture:
    CHUNK = 512          # the sums over samples are [8 x N] x [N x 8] products: as ONE GEMM each they run on a single
                         # workgroup (15 ms at N = 100k); as a batch over 512-sample chunks + a sum they take 0.3 ms

    def __init__(self, X: torch.Tensor, reg_covar: float):
        self.reg = reg_covar
        self.N, self.d = X.shape
        self.C = (self.N + self.CHUNK - 1) // self.CHUNK
        self.Np = self.C * self.CHUNK
        self.X = torch.zeros((self.Np, self.d), dtype=X.dtype, device=X.device)      # zero rows pad the last chunk; their
        self.X[: self.N] = X                                                           # responsibilities stay zero
        self.eye = torch.eye(self.d, dtype=X.dtype, device=X.device)

    def m_step(self, resp: torch.Tensor, seeding: bool) -> None:
        X, C, n, d = self.X, self.C, self.CHUNK, self.d
        K = resp.shape[1]
        if resp.shape[0] != self.Np:
            rp = torch.zeros((self.Np, K), dtype=resp.dtype, device=resp.device)
            rp[: self.N] = resp
            resp = rp
        nk = resp.sum(dim=0) + 10 * np.finfo(np.float64).eps
        self.means = torch.bmm(resp.reshape(C, n, K).transpose(1, 2), X.view(C, n, d)).sum(dim=0) / nk[:, None]
        diff = X[None, :, :] - self.means[:, None, :]                            # [K, Np, d]
        wd = resp.T[:, :, None] * diff
        cov = torch.bmm(wd.reshape(K * C, n, d).transpose(1, 2), diff.reshape(K * C, n, d)).view(K, C, d, d).sum(dim=1)
        cov = cov / nk[:, None, None] + self.reg * self.eye
        self.weights = nk / self.N if seeding else nk / nk.sum()
        pc = _precision_factors(cov.cpu().numpy())
        self.log_det = torch.from_numpy(np.log(np.diagonal(pc, axis1=1, axis2=2)).sum(axis=1)).to(X.device)
        self.prec_chol = torch.from_numpy(pc).to(X.device)

    def e_step(self):
        """(mean log-likelihood per sample, log responsibilities [Np, K] with the padding rows at -inf -> resp 0)."""
        y = torch.matmul(self.X[None], self.prec_chol) - torch.matmul(self.means[:, None, :], self.prec_chol)   # [K, Np, d]
        maha = (y * y).sum(dim=2).T                                              # [Np, K]
        logp = -0.5 * (self.d * math.log(2 * math.pi) + maha) + self.log_det + torch.log(self.weights)
        norm = torch.logsumexp(logp, dim=1)
        log_resp = logp - norm[:, None]
        log_resp[self.N:] = -math.inf
        return float(norm[: self.N].mean().item()), log_resp


def fit_means(X_pca: np.ndarray, k: int, seed: int, device=None, n_init: int = 5, tol: float = 1e-4, max_iter: int = 100,
              reg_covar: float = 1e-6) -> np.ndarray:
    """means_ [k, d] (float64) of the reference's GaussianMixture call on X_pca [N, d] (float64)."""
    from sklearn.cluster import kmeans_plusplus
    from sklearn.utils import check_random_state
    Xh = np.ascontiguousarray(X_pca, dtype=np.float64)
    N = Xh.shape[0]
    if N < k:
        raise ValueError(f"Expected n_samples >= n_components but got n_components = {k}, n_samples = {N}")
    X = torch.from_numpy(Xh).to(device if device is not None else "cpu")
    rs = check_random_state(seed)
    mix = _Mixture(X, reg_covar)
    runs = []
    for _ in range(n_init):
        _, picks = kmeans_plusplus(Xh, k, random_state=rs)
        resp = torch.zeros((N, k), dtype=X.dtype, device=X.device)
        resp[torch.from_numpy(np.asarray(picks)).to(X.device), torch.arange(k, device=X.device)] = 1
        mix.m_step(resp, seeding=True)
        bound = -math.inf
        for _it in range(max_iter):
            prev = bound
            bound, log_resp = mix.e_step()
            mix.m_step(torch.exp(log_resp), seeding=False)
            if abs(bound - prev) < tol:
                break
        runs.append((bound, mix.means.cpu().numpy().copy()))
    # the highest objective wins; restarts within 1e-10 of it reached the same optimum (in the library the winner among those is decided
    # by the rounding of its sums): the first of them, like csrc/nadm_gmm.cpp
    top = max(b for b, _ in runs)
    return next(m for b, m in runs if b >= top - 1e-10)
