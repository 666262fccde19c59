"""Stand-alone worker: one sklearn GaussianMixture fit, exactly the reference's call (model/train.py:61,66).

Run as a script (``python _gmm_fit.py X.npy k seed out.npy``) by train.gmm_p_init when several K have to be fitted:
the fits are independent, sklearn's fit is single-threaded Python, and a child started with exec shares nothing with the
parent's GPU context.  Imports numpy and sklearn only (importing the package would load torch and libnadm)."""
import sys

import numpy as np


def fit_means(X_pca: np.ndarray, k: int, seed: int) -> np.ndarray:
    from sklearn.mixture import GaussianMixture
    gmm = GaussianMixture(n_components=k, n_init=5, init_params="k-means++", tol=1e-4, covariance_type="full",
                          max_iter=100, random_state=seed)
    limit = None
    try:                                    # the matrices are [N, n_components <= 8]: a BLAS/OpenMP pool of 64-256 threads only
        from threadpoolctl import threadpool_info, threadpool_limits     # adds overhead (2x on a 256-thread host, identical means)
        # only ever LOWER a pool: raising one that came up with fewer threads (OPENBLAS_NUM_THREADS=1 in the environment, which the CLI
        # sets like the reference's does, entry.py:138-146) crashes OpenBLAS inside the fit.  threadpoolctl limits per API, so the
        # API's limit is the smallest of 4 and what any of its libraries has now (r05: scipy's OpenBLAS, first loaded here, started with 1
        # while numpy's had 64 -- "limits=4" raised it and the fit died in trtrs)
        per_api = {}
        for p_ in threadpool_info():
            api, n = p_.get("user_api"), p_.get("num_threads")
            if api and n:
                per_api[api] = min(per_api.get(api, 4), n, 4)
        if per_api:
            limit = threadpool_limits(limits=per_api)
    except ImportError:
        pass
    try:
        gmm.fit(X_pca)
    finally:
        if limit is not None:
            limit.restore_original_limits()
    return gmm.means_


if __name__ == "__main__":
    x_path, k, seed, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    np.save(out_path, fit_means(np.load(x_path), k, seed))
