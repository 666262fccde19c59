"""Time nadm_loglik (the post-training log-likelihood report) on a resident packed matrix: random codes, random Q / P.
    python tools/loglik_timing.py [N M K]        default 100000 500000 8 (configs[3])"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from neural_admixture_amd.report import loglikelihood_hip  # noqa: E402

N, M, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (100_000, 500_000, 8)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
xp = torch.randint(0, 256, (N, (M + 3) // 4), dtype=torch.uint8, device=dev, generator=g)
rng = np.random.default_rng(2)
P = rng.uniform(0.01, 0.99, size=(M, K)).astype(np.float32)
Q = rng.dirichlet(np.ones(K), size=N).astype(np.float32)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v = loglikelihood_hip(xp, M, P, Q)
    torch.cuda.synchronize()
    print(f"N {N} M {M} K {K}: loglik {v:.6e}  {time.perf_counter() - t0:.4f} s (incl. the H2D copies of P and Q)")
