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
        pools = threadpool_info()
        if pools and max(p_["num_threads"] for p_ in pools) > 4:         # only ever LOWER the count: raising a pool that was
            limit = threadpool_limits(limits=4)                           # started with OMP_NUM_THREADS=1 crashes OpenBLAS
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
