import os, sys, numpy as np, torch, logging
logging.disable(logging.CRITICAL)
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import seeded_inputs as SI
import neural_admixture_amd as na
from neural_admixture_amd.svd import RSVD
d = np.load("/root/repo/tests/golden/c2_end_to_end.npz")
N, M, K, C = int(d["N"]), int(d["M"]), int(d["K"]), int(d["C"])
G = torch.from_numpy(SI.genotypes(N, M, K, int(d["seed"]), threads=32))
dev = torch.device("cuda:0")
Vt = RSVD(G, N, M, C, int(d["run_seed"]), device=dev)
rows = SI.sample_rows(M, int(d["nrows"]), int(d["seed"]))
mx = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
print("Vt diff per comp vs ref rows", np.abs(Vt[:, rows] - d["Vt_rows"]).max(1))
P0 = np.clip(d["gmm_means"] @ Vt, 5e-6, 1 - 5e-6).astype(np.float32)
def run(Vt_):
    tr = na.NeuralAdmixture(K, int(d["epochs"]), int(d["b"]), float(d["lr"]), dev, int(d["run_seed"]), 1, True, None, None, None)
    Qs, Ps, model = tr.launch_training(torch.from_numpy(P0), G, int(d["Hd"]), C, torch.from_numpy(np.ascontiguousarray(Vt_.T)), M, N, None)
    return Qs[0], Ps[0]
Q0, Pq0 = run(Vt)
print("base vs ref: dQ", mx(Q0, d["hi_Q"]))
rng = np.random.default_rng(1)
for comp, amp in ((7, 2.7e-6), (7, 2.7e-7), (0, 1e-8), (3, 1e-8)):
    V2 = Vt.copy()
    z = rng.standard_normal(M).astype(np.float32)
    V2[comp] += amp * z / np.abs(z).max()
    Q1, P1 = run(V2)
    print(f"perturb comp {comp} by max {amp:g}: HIP-vs-HIP dQ {mx(Q1, Q0):.3e} mean {np.abs(Q1-Q0).mean():.3e} dP {mx(P1[rows], Pq0[rows]):.3e}")
# replace the sampled rows of OUR Vt by the reference's rows: how much of the distance do 4096 of 600k rows explain?
