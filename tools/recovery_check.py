#!/usr/bin/env python3
"""End-to-end functional check at 1000-Genomes scale: synthetic admixed genotypes with KNOWN ancestry fractions
(G ~ Binomial(2, Qt.F), 1 % missing), default run (RSVD + GMM init + 250 epochs, batch 800), then the RMSE between the
estimated Q and the true one after matching columns.  Usage: recovery_check.py [N] [M] [K] [epochs]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    N, M, K, epochs = (int(a) for a in (sys.argv[1:5] + ["2504", "600000", "7", "250"][len(sys.argv) - 1:]))
    import neural_admixture_amd as na
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.io import PackedGenotypes
    from neural_admixture_amd.layout import ModelLayout
    from neural_admixture_amd.svd import RSVD
    from scipy.optimize import linear_sum_assignment
    dev = torch.device("cuda:0")
    ld = ModelLayout.row_stride(M)
    torch.manual_seed(7)
    Fq = (0.5 * torch.distributions.Beta(torch.tensor(0.5), torch.tensor(0.5)).sample((K, M))).clamp(0.005, 0.5).float().to(dev)
    Qt = torch.distributions.Dirichlet(torch.full((K,), 0.2)).sample((N,)).float().to(dev)
    xp = torch.empty((N, ld), dtype=torch.uint8, device=dev)
    check(lib.nadm_synth_packed(ptr(xp), N, 0, M, ld, ptr(Qt), ptr(Fq), K, 0.01, 99, None))
    torch.cuda.synchronize()
    data = PackedGenotypes(xp.cpu(), N, M)
    t0 = time.time()
    seed = int(os.environ.get("NADM_RC_SEED", "42"))              # (the trainer's seed: epoch orders and small-parameter init)
    V = RSVD(data, N, M, 8, 42)
    Ps, Qs, _ = na.train(epochs, 800, 2e-3, K, seed, data, dev, 1, 1024, True, V, None, None, None, 8)
    dt = time.time() - t0
    Q, Qtrue, F = Qs[0].astype(np.float64), Qt.cpu().numpy().astype(np.float64), Fq.cpu().numpy().astype(np.float64)
    cost = ((Q[:, :, None] - Qtrue[:, None, :]) ** 2).sum(0)            # [est, true]
    r, c = linear_sum_assignment(cost)
    rmse_q = float(np.sqrt(((Q[:, r] - Qtrue[:, c]) ** 2).mean()))
    rmse_p = float(np.sqrt(((Ps[0].astype(np.float64)[:, r] - F.T[:, c]) ** 2).mean()))     # P = allele frequency of the coded allele / ... see note
    print(json.dumps({"N": N, "M": M, "K": K, "epochs": epochs, "seed": seed, "seconds": dt, "rmse_Q": rmse_q, "rmse_P_vs_F": rmse_p,
                      "Q_row_sums_min_max": [float(Q.sum(1).min()), float(Q.sum(1).max())]}))


if __name__ == "__main__":
    main()
