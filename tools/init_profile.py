#!/usr/bin/env python3
"""Where does a default run spend its wall-clock OUTSIDE the epoch loop?  cProfile (cumulative host time; device work shows up where
the host waits for it) of RSVD and of train() on the c2 / c3 shapes of tools/full_run.py.  Usage: init_profile.py [c2|c3] [epochs]
-> gpurun_out/r06_init_profile_<cfg>.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    import neural_admixture_amd as na
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.io import PackedGenotypes, write_outputs, save_model
    from neural_admixture_amd.layout import ModelLayout
    from neural_admixture_amd.svd import RSVD
    dev = torch.device("cuda:0")
    N, M, Ktrue = 2504, 600_000, 7
    ld = ModelLayout.row_stride(M)
    torch.manual_seed(1234)
    Fq = (0.5 * torch.distributions.Beta(torch.tensor(0.5), torch.tensor(0.5)).sample((Ktrue, M))).clamp(0.005, 0.5).float().to(dev)
    Qt = torch.distributions.Dirichlet(torch.full((Ktrue,), 0.2)).sample((N,)).float().to(dev)
    xp = torch.empty((N, ld), dtype=torch.uint8, device=dev)
    check(lib.nadm_synth_packed(ptr(xp), N, 0, M, ld, ptr(Qt), ptr(Fq), Ktrue, 0.01, 1234, None))
    torch.cuda.synchronize()
    data = PackedGenotypes(xp.cpu(), N, M)
    del xp
    K, mn, mx = (7, None, None) if which == "c2" else (None, 2, 10)
    lines = []

    def prof(name, fn):
        pr = cProfile.Profile()
        t = time.time()
        pr.enable()
        r = fn()
        torch.cuda.synchronize()
        pr.disable()
        dt = time.time() - t
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(48)
        lines.append(f"==== {name}: {dt:.3f} s wall\n" + "\n".join(l for l in s.getvalue().splitlines()[4:] if l.strip()))
        return r
    V = prof("RSVD", lambda: RSVD(data, N, M, 8, 42))
    Ps, Qs, model = prof("train", lambda: na.train(epochs, 800, 2e-3, K, 42, data, dev, 1, 1024, True, V, None, mn, mx, 8))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        prof("write", lambda: (save_model(model, "run", td), write_outputs(Qs, "run", K, mn, mx, td, Ps)))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/r06_init_profile_{which}.txt", "w") as f:
        f.write("\n\n".join(lines) + "\n")
    print("\n\n".join(lines))


if __name__ == "__main__":
    main()
