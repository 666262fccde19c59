#!/usr/bin/env python3
"""Wall-clock of full default runs (250 epochs, batch 800) on the 1000-Genomes-scale configs of BASELINE.json:
c2 = 2504 samples x 600k SNPs, K=7 single head; c3 = same matrix, multi-head K=2..10; c4 = 100k x 500k, K=8 (the
bench workload, here as a complete run on ONE GPU); c5 = 500k x 1M, K=16 (configs[4], 125 GB packed, on ONE GPU).  Synthetic admixture-model
genotypes generated on the device; RSVD (GPU, from packed) + GMM init + training + final Q + log-likelihood + writing
.Q/.P, i.e. everything `neural-admixture train` does after reading the file.  Usage: full_run.py [c2|c3|c4|c5] [epochs]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    if os.environ.get("NADM_LOG"):                     # timestamps of the reference-style progress messages on stderr
        import logging
        logging.basicConfig(format="%(asctime)s.%(msecs)03d %(message)s", datefmt="%H:%M:%S", level=logging.INFO, stream=sys.stderr)
    import neural_admixture_amd as na
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.io import PackedGenotypes, write_outputs, save_model
    from neural_admixture_amd.layout import ModelLayout
    from neural_admixture_amd.svd import RSVD
    dev = torch.device("cuda:0")
    N, M, Ktrue = {"c4": (100_000, 500_000, 8), "c5": (500_000, 1_000_000, 16)}.get(which, (2504, 600_000, 7))
    ld = ModelLayout.row_stride(M)
    torch.manual_seed(1234)
    Fq = (0.5 * torch.distributions.Beta(torch.tensor(0.5), torch.tensor(0.5)).sample((Ktrue, M))).clamp(0.005, 0.5).float().to(dev)
    Qt = torch.distributions.Dirichlet(torch.full((Ktrue,), 0.2)).sample((N,)).float().to(dev)
    xp = torch.empty((N, ld), dtype=torch.uint8, device=dev)
    for r0 in range(0, N, 50_000):                                      # row blocks: the generator takes Qt per block
        n = min(50_000, N - r0)
        check(lib.nadm_synth_packed(ptr(xp[r0:]), n, r0, M, ld, ptr(Qt[r0:]), ptr(Fq), Ktrue, 0.01, 1234, None))
    torch.cuda.synchronize()
    big = which in ("c4", "c5")
    data = PackedGenotypes(xp if big else xp.cpu(), N, M)     # c4 / c5: 12.5 / 125 GB, left resident in HBM (as after io.read_bed_packed(keep_on_device=True))
    del xp
    out = {"config": which, "N": N, "M": M, "epochs": epochs}
    t0 = time.time()
    rsvd_phases = {} if os.environ.get("NADM_RSVD_PHASES") else None     # (synchronises at every stage boundary: not for the headline total)
    V = RSVD(data, N, M, 8, 42, phases=rsvd_phases)
    out["rsvd_s"] = time.time() - t0
    if rsvd_phases is not None:
        out["rsvd_phases"] = {k_: round(v_, 4) for k_, v_ in rsvd_phases.items()}
    K, mn, mx = {"c2": (7, None, None), "c4": (8, None, None), "c5": (16, None, None)}.get(which, (None, 2, 10))
    # phase timers inside train(): wrap the module-level helpers it calls
    import importlib
    tr_mod = importlib.import_module("neural_admixture_amd.train")
    eng_mod = importlib.import_module("neural_admixture_amd.engine")
    phases = {}

    def timed(name, fn):
        def w(*a, **k):
            torch.cuda.synchronize()
            t = time.time()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            phases[name] = phases.get(name, 0.0) + time.time() - t
            return r
        return w
    tr_mod.gmm_p_init = timed("gmm_init_s", tr_mod.gmm_p_init)
    tr_mod.loglikelihood_packed = timed("loglik_s", tr_mod.loglikelihood_packed)
    eng_mod.Engine.pack_from_host = timed("pack_h2d_s", eng_mod.Engine.pack_from_host)
    eng_mod.Engine.load_params = timed("load_params_s", eng_mod.Engine.load_params)
    t1 = time.time()
    Ps, Qs, model = na.train(epochs, 800, 2e-3, K, 42, data, dev, 1, 1024, True, V, None, mn, mx, 8)
    torch.cuda.synchronize()
    out["train_call_s"] = time.time() - t1          # GMM init + pack/H2D + epochs + final Q + log-likelihood
    out["train_phases"] = phases
    with tempfile.TemporaryDirectory() as td:
        t2 = time.time()
        save_model(model, "run", td)
        write_outputs(Qs, "run", K, mn, mx, td, Ps)
        out["write_s"] = time.time() - t2
    out["total_s"] = time.time() - t0
    # epoch-loop-only timing: re-run the loop alone on the resident engine state
    eng = model.engine
    idx = torch.randperm(N, device=dev).to(torch.int32)
    torch.cuda.synchronize()
    t3 = time.time()
    for _ in range(10 if not big else (2 if which == "c4" else 1)):
        for s in range(0, N, 800):
            bb = min(800, N - s)
            eng.train_step(idx[s:s + bb], bb, 2e-3, False)
    torch.cuda.synchronize()
    ep = (time.time() - t3) / (10 if not big else (2 if which == "c4" else 1))
    out["epoch_s"] = ep
    out["genotypes_per_s_epoch_loop"] = N * M / ep
    print(json.dumps(out))


if __name__ == "__main__":
    main()
