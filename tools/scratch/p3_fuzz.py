#!/usr/bin/env python3
"""Random shapes for pass 3 (FP4 x FP6 instruction): dV from (a) the resident matrix + a row permutation and (b) pass 2's tiled copy of the
batch (a whole backward step) against a float64 product of the same dZ.  Prints the worst relative error (of max |dV|) over the cases."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nadm_oracle as O
import neural_admixture_amd as na
from neural_admixture_amd._lib import lib, check, ptr
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
worst = 0.0
for case in range(40):
    N = int(rng.integers(1, 900)); M = int(rng.integers(70, 30000)); C = int(rng.choice([2, 3, 4, 5, 8])); K = int(rng.choice([2, 5, 8, 11]))
    b = int(rng.integers(1, N + 1))
    Gm = O.synth_genotypes(N, M, 3, seed=case, missing=float(rng.choice([0.0, 0.03, 0.5])))
    V0 = (rng.standard_normal((M, C)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, size=(K, M)).astype(np.float32)
    p = O.make_params(case, V0, P0, 32, [K])
    e = na.Engine(M, C, 32, [K], dev, N)
    e.load_params(p.V, np.concatenate([P.T for P in p.P], axis=0), np.zeros(e.lay.n_small, dtype=np.float32) + 0.01)
    e.pack_from_host(torch.from_numpy(np.ascontiguousarray(Gm)))
    perm = rng.permutation(N)[:b].astype(np.int32)
    idx = torch.from_numpy(perm).to(dev)
    CP = e.lay.CP
    X = np.where(Gm[perm] == 3, 0, Gm[perm]).astype(np.float64) / 2
    # (a) resident matrix, own dZ
    dZ = np.zeros((b, CP), dtype=np.float32); dZ[:, :C] = rng.standard_normal((b, C)) * np.exp2(rng.uniform(-8, 8, size=(b, 1)))
    e.dZ[: b * CP] = torch.from_numpy(dZ.reshape(-1)).to(dev); e.invalidate_dz()
    check(lib.nadm_encode_bwd(ptr(e.xp), e.ld, ptr(idx), b, M, ptr(e.dZ), e._dz_image(b), CP, ptr(e.gbig), 0, None))
    torch.cuda.synchronize()
    got = e.gV().cpu().numpy().astype(np.float64)[:, :C]
    ref = X.T @ dZ[:, :C].astype(np.float64)
    ea = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300)
    # (b) a whole step: pass 2 leaves the tiled copy, the MLP backward the image
    e.forward(idx, b); e.backward(idx, b, True); torch.cuda.synchronize()
    dZ2 = e.dZ.cpu().numpy()[: b * CP].reshape(b, CP)[:, :C].astype(np.float64)
    got2 = e.gV().cpu().numpy().astype(np.float64)[:, :C]
    ref2 = X.T @ dZ2
    eb = np.abs(got2 - ref2).max() / (np.abs(ref2).max() + 1e-300)
    worst = max(worst, ea, eb)
    print(f"case {case}: N={N} b={b} M={M} C={C} K={K}  resident {ea:.2e}  copy {eb:.2e}", flush=True)
    assert ea < 2e-6 and eb < 2e-6
print("worst", worst)
