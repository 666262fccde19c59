#!/usr/bin/env python3
"""Default-length run (250 epochs) on the bundled demo matrix: HIP path vs the numpy fp32 oracle from identical init
(the reference's own RSVD V and GMM P_init from tests/golden/demo_k3.npz).  Prints max/mean |dQ|, |dP| at several horizons."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nadm_oracle as O
import neural_admixture_amd as na
from neural_admixture_amd.model import NeuralAdmixture

d = np.load(os.path.join(ROOT, "tests", "golden", "demo_k3.npz"))
N, M, K, Hd, seed, lr = int(d["N"]), int(d["M"]), int(d["K"]), int(d["Hd"]), int(d["seed"]), float(d["lr"])
G = O.unpack2bit(d["G_packed"], M)
V = np.ascontiguousarray(d["Vt"].T.astype(np.float32))
out = {}
for ep in (5, 25, 100, 250):
    p = O.make_params(seed, V, d["P_init"], Hd, [K])
    p, Qo, _ = O.train_run(G, p, ep, 800, lr, seed)
    tr = NeuralAdmixture(K, ep, 800, lr, torch.device("cuda:0"), seed, 1, True, None, None, None)
    Qs, Ps, _ = tr.launch_training(torch.from_numpy(d["P_init"].copy()), torch.from_numpy(G), Hd, 8, torch.from_numpy(V.copy()), M, N, None)
    dq, dp = np.abs(Qs[0] - Qo[0]), np.abs(Ps[0] - p.P[0])
    out[ep] = {"max_dQ": float(dq.max()), "mean_dQ": float(dq.mean()), "max_dP": float(dp.max()), "mean_dP": float(dp.mean())}
    if f"hi_e{ep}_Q" in d.files:
        out[ep]["max_dQ_vs_reference_fp32"] = float(np.abs(Qs[0] - d[f"hi_e{ep}_Q"]).max())
        out[ep]["reference_bf16_vs_fp32_max_dQ"] = float(np.abs(d[f"med_e{ep}_Q"] - d[f"hi_e{ep}_Q"]).max())
print(json.dumps(out))
