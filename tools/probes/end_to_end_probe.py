import os, sys, logging, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import seeded_inputs as SI
import neural_admixture_amd as na
from neural_admixture_amd.svd import RSVD
T = sys.modules[na.train.__module__]
d = np.load("/root/repo/tests/golden/c2_end_to_end.npz")
N, M, K, C = int(d["N"]), int(d["M"]), int(d["K"]), int(d["C"])
G = torch.from_numpy(SI.genotypes(N, M, K, int(d["seed"]), threads=32))
dev = torch.device("cuda:0")
Vt = RSVD(G, N, M, C, int(d["run_seed"]), device=dev)
rows = SI.sample_rows(M, int(d["nrows"]), int(d["seed"]))
mx = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
print("log name", T.log.name)
Xp = T.pca_project_gpu(G.numpy(), Vt, dev)
print("X_pca rows vs ref", mx(Xp[:256], d["X_pca_rows"]), "scale", np.abs(d["X_pca_rows"]).max())
from neural_admixture_amd import gmm as _g
from neural_admixture_amd._gmm_fit import fit_means as sk_fit
X64 = Xp.astype("float64")
m_nat = _g.fit_means(X64, K, int(d["run_seed"]))
m_sk = sk_fit(X64, K, int(d["run_seed"]))
print("means native vs ref", mx(m_nat, d["gmm_means"]), " sklearn(here) vs ref", mx(m_sk, d["gmm_means"]), " native vs sklearn(here)", mx(m_nat, m_sk))
print("ref n_iter", int(d["gmm_n_iter"]), "lower bound", float(d["gmm_lower_bound"]))
for name, means in (("ref_means", d["gmm_means"]), ("native", m_nat), ("sklearn_here", m_sk)):
    P0 = np.clip(means @ Vt, 5e-6, 1 - 5e-6).astype(np.float32)
    tr = na.NeuralAdmixture(K, int(d["epochs"]), int(d["b"]), float(d["lr"]), dev, int(d["run_seed"]), 1, True, None, None, None)
    Qs, Ps, model = tr.launch_training(torch.from_numpy(P0), G, int(d["Hd"]), C, torch.from_numpy(np.ascontiguousarray(Vt.T)), M, N, None)
    print(name, "dQ", mx(Qs[0], d["hi_Q"]), "mean|dQ|", float(np.abs(Qs[0] - d["hi_Q"]).mean()), "dP", mx(Ps[0][rows], d["hi_P_rows"]),
          "| ref hi-med dQ", mx(d["med_Q"], d["hi_Q"]), "mean", float(np.abs(d["med_Q"] - d["hi_Q"]).mean()))
