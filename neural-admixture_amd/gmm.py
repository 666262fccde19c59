"""Decoder init for 1000-Genomes-sized inputs without importing scikit-learn: the means of the reference's mixture fit
(model/train.py:61-66: ``GaussianMixture(n_components=K, n_init=5, init_params='k-means++', tol=1e-4, covariance_type='full',
max_iter=100, random_state=seed).fit(X_pca).means_``) from

* ``kmeanspp_picks`` -- the seeding draws.  scikit-learn's k-means++ (Arthur & Vassilvitskii 2007, greedy variant with
  2 + floor(log K) local trials) is the only consumer of random numbers in that call: one ``RandomState(seed)`` stream, shared by
  the five restarts, from which every restart takes one ``choice`` for the first seed and ``uniform(size=trials)`` per further seed,
  turned into sample indices by a search in the cumulative squared distances.  Restated here on numpy's own ``RandomState`` with the
  same expressions for the distances (so that the searches see the same roundings); the library's function is the test oracle
  (tests/test_abi_and_host.py) and ``train.gmm_p_init(fit="sklearn")`` still calls the library itself.
* ``nadm_gmm_fit_means`` (csrc/nadm_gmm.cpp) -- the EM iterations of the five restarts, float64, one thread per restart.

Why: on a 2504 x 600k run the library fit is 0.55 s and its import 1.0 s of a 1.9 s default run whose 250 epochs take 0.34 s
(profiles/r05_init_profile_before.txt); this path takes ~0.05 s and the same means to 1e-12."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib, check


def kmeanspp_picks(X: np.ndarray, k: int, rs: np.random.RandomState) -> np.ndarray:
    """Row indices of the k seeds scikit-learn's ``kmeans_plusplus(X, k, random_state=rs)`` picks, consuming ``rs`` like it does."""
    n = X.shape[0]
    xsq = np.einsum("ij,ij->i", X, X)
    sw = np.ones(n, dtype=X.dtype)
    trials = 2 + int(np.log(k))

    def dist(Cn):                                           # squared distances of the rows Cn to every sample, the library's expression
        d = -2 * (Cn @ X.T)
        d += np.einsum("ij,ij->i", Cn, Cn)[:, None]
        d += xsq.reshape(1, -1)
        np.maximum(d, 0, out=d)
        return d

    picks = np.full(k, -1, dtype=np.int64)
    picks[0] = rs.choice(n, p=sw / sw.sum())
    closest = dist(X[picks[0], np.newaxis])                 # [1, n]
    pot = closest @ sw
    for c in range(1, k):
        rand_vals = rs.uniform(size=trials) * pot
        cand = np.searchsorted(np.cumsum(sw * closest, axis=None, dtype=np.float64), rand_vals)
        np.clip(cand, None, closest.size - 1, out=cand)
        dc = dist(X[cand])
        np.minimum(closest, dc, out=dc)
        cpot = dc @ sw.reshape(-1, 1)
        best = int(np.argmin(cpot))
        pot = cpot[best]
        closest = dc[best]
        picks[c] = cand[best]
    return picks


DEVICE_MIN_SAMPLES = 20_000      # (0.82 s on host threads against 0.033 s at N = 100k; at N = 2504 a warm call is 0.009 against 0.021 s, but the
                                 # first call's allocation and code loading make a default run no faster: profiles/r05_gmm_timing.txt, r05_full_runs.txt)


def device_form_applies(N: int, d: int, k: int) -> bool:
    """The shapes csrc/nadm_gmm_dev.hip is built for (the reference's 8 PCA components, K <= 16) at sizes where it pays."""
    return d == 8 and 1 <= k <= 16 and N >= DEVICE_MIN_SAMPLES


def fit_means(X_pca: np.ndarray, k: int, seed: int, n_init: int = 5, tol: float = 1e-4, max_iter: int = 100,
              reg_covar: float = 1e-6, stream: int | None = None) -> np.ndarray:
    """means_ [k, d] (float64) of the reference's GaussianMixture call on X_pca [N, d].  ``stream`` (a HIP stream handle, 0 = the
    default stream): run the EM sums on the GPU (nadm_gmm_fit_means_dev; d = 8, k <= 16 -- see ``device_form_applies``) instead of
    on host threads; the seeding draws are the same either way."""
    X = np.ascontiguousarray(X_pca, dtype=np.float64)
    N, d = X.shape
    if N < k:
        raise ValueError(f"Expected n_samples >= n_components but got n_components = {k}, n_samples = {N}")
    rs = np.random.RandomState(seed)                        # (= sklearn.utils.check_random_state(seed))
    picks = np.ascontiguousarray(np.stack([kmeanspp_picks(X, k, rs) for _ in range(n_init)]), dtype=np.int32)
    means = np.empty((k, d), dtype=np.float64)
    bound, iters = C.c_double(0.0), C.c_int32(0)
    if stream is not None:
        rc = lib.nadm_gmm_fit_means_dev(X.ctypes.data, N, d, k, picks.ctypes.data, n_init, tol, max_iter, reg_covar, means.ctypes.data,
                                        C.byref(bound), C.byref(iters), C.c_void_p(stream))
    else:
        rc = lib.nadm_gmm_fit_means(X.ctypes.data, N, d, k, picks.ctypes.data, n_init, tol, max_iter, reg_covar, means.ctypes.data,
                                    C.byref(bound), C.byref(iters))
    if rc:
        msg = (lib.nadm_last_error() or b"").decode()
        if "ill-defined empirical covariance" in msg:
            raise ValueError(msg)                            # the library's own error for this input
        check(rc, "gmm_fit_means")
    fit_means.last = {"lower_bound": bound.value, "n_iter": iters.value, "device": stream is not None}
    return means
