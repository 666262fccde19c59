#!/usr/bin/env python3
"""A long version of tests/test_soak_handoffs.py, run once per round on the GPU box: N production steps (nadm_step: Q images, dZ image
built by the MLP backward's last blocks, small update riding in the next pass 1) against the same steps as the unfused launch
sequence, bit for bit; batch sizes cycle 800 / 790 / 37 / 800 / 128 / 1.  -> gpurun_out/r06_soak.txt

    python tools/soak.py [steps=100000] [dp [buckets=1]]          NADM_SOAK_KS=16 or 3,5,9: other heads than the default K = 8

dp: the sample-sharded launch sequence (NADM_MODE_DP on a 1-rank RCCL communicator: side stream + events every step, Adam as launches of
its own) against the single-GPU step instead -> gpurun_out/r06_soak_dp.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_soak_handoffs import _engines, _same_state          # noqa: E402
from unfused_step import unfused_step                          # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    dp = len(sys.argv) > 2 and sys.argv[2] == "dp"
    buckets = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # message B: 1 = on the compute stream (the default), > 1 = SNP-range buckets
    M, N = 60_000, 4000
    ks = [int(v) for v in os.environ.get("NADM_SOAK_KS", "8").split(",")]
    sizes = tuple(int(v) for v in os.environ.get("NADM_SOAK_SIZES", "800,790,37,800,128,1").split(","))   # 4200,4100,37,4200: pass 3 in sample slices too
    N = max(N, max(sizes) + 100)
    prod, ref = _engines(M, ks, 1024, N, seed=101, bmax=max(sizes))
    dev = prod.device
    if dp:
        import neural_admixture_amd as na
        from neural_admixture_amd.comm import rccl_comm
        comm = rccl_comm(0, 1)
        e = na.Engine(M, 8, 1024, ks, dev, max(sizes), mode="dp", comm=comm, n_buckets=buckets)
        assert e.lay.n_buckets == buckets
        e.pflat.copy_(prod.pflat)
        e.set_packed(prod.xp)
        prod = e
    step_ref = (lambda idx, b, wl: ref.train_step(idx, b, 2e-3, wl)) if dp else (lambda idx, b, wl: unfused_step(ref, idx, b, 2e-3, wl))
    gen = torch.Generator().manual_seed(11)
    t0 = time.time()
    bad = None
    for s in range(steps):
        b = sizes[s % len(sizes)]
        idx = torch.randint(0, N, (b,), generator=gen, dtype=torch.int32).to(dev)
        wl = (s % 7) != 3
        prod.train_step(idx, b, 2e-3, wl)
        step_ref(idx, b, wl)
        if s % 5000 == 4999:
            torch.cuda.synchronize()
            ok = _same_state(prod, ref) and int(prod._dzcnt.abs().sum().item()) == 0
            print(f"step {s + 1}: {'identical' if ok else 'DIFFERENT'}  ({time.time() - t0:.0f} s)", flush=True)
            if not ok:
                bad = s + 1
                break
    torch.cuda.synchronize()
    ok = bad is None and _same_state(prod, ref) and prod.read_loss() == ref.read_loss()
    what = (f"sample-sharded steps (NADM_MODE_DP, 1-rank RCCL communicator, message B in {buckets} bucket(s)) vs the single-GPU step" if dp else "production steps vs the unfused launch sequence")
    line = (f"r06 soak: {steps if bad is None else bad} consecutive {what} (M = {M}, K = {'/'.join(map(str, ks))}, Hd = 1024, batch sizes cycling {sizes}): parameters, moments "
            f"and loss sums {'BIT-IDENTICAL' if ok else 'DIFFER'}; group counters zero; {time.time() - t0:.0f} s")
    print(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", (f"r06_soak_dp_b{buckets}" if dp else "r06_soak") + ("" if ks == [8] else "_k" + "_".join(map(str, ks))) + ("" if max(sizes) == 800 else f"_b{max(sizes)}") + ".txt"), "w") as f:
        f.write(line + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
