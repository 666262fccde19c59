#!/usr/bin/env python3
"""Where the decoder-init mixture fit spends its time on this host: seeding draws (numpy) against the EM (csrc/nadm_gmm.cpp), for the
sample counts of the BASELINE configs, and -- with a GPU -- the EM with its sums on the device (csrc/nadm_gmm_dev.hip) beside it.
Usage: gmm_timing.py  -> stdout"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_admixture_amd import gmm                      # noqa: E402
from neural_admixture_amd._lib import lib                  # noqa: E402

rng = np.random.default_rng(0)
for N, k in ((2504, 7), (100_000, 8), (500_000, 16)):
    cent = rng.standard_normal((k, 8))
    X = (rng.dirichlet(np.full(k, 0.2), N) @ cent + 0.05 * rng.standard_normal((N, 8))).astype(np.float64)
    t = time.time()
    rs = np.random.RandomState(42)
    picks = np.ascontiguousarray(np.stack([gmm.kmeanspp_picks(X, k, rs) for _ in range(5)]), dtype=np.int32)
    t_seed = time.time() - t
    means = np.empty((k, 8))
    b, it = C.c_double(), C.c_int32()
    t = time.time()
    lib.nadm_gmm_fit_means(X.ctypes.data, N, 8, k, picks.ctypes.data, 5, 1e-4, 100, 1e-6, means.ctypes.data, C.byref(b), C.byref(it))
    t_em = time.time() - t
    print(f"N = {N:7d} K = {k:2d}: seeding {t_seed:.3f} s, EM (5 restarts) {t_em:.3f} s, winner: {it.value} iterations, objective {b.value:.6f}; "
          f"host threads {os.cpu_count()}")
    import torch
    if torch.cuda.is_available():
        torch.zeros(1, device="cuda:0")
        for rep in range(2):
            md = np.empty((k, 8))
            bd, itd = C.c_double(), C.c_int32()
            t = time.time()
            rc = lib.nadm_gmm_fit_means_dev(X.ctypes.data, N, 8, k, picks.ctypes.data, 5, 1e-4, 100, 1e-6, md.ctypes.data, C.byref(bd), C.byref(itd), None)
            t_dev = time.time() - t
        print(f"                   on the device: EM {t_dev:.3f} s (second call), rc {rc}, winner: {itd.value} iterations, objective {bd.value:.6f}, "
              f"max |means - host| {np.abs(md - means).max():.2e}")
