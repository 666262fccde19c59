"""Accuracy of pass 2's gradients in the REALISTIC regime (P from the mixture init: many entries at the 5e-6 clip, heavy cancellation in the sums)
against float64, beside the same chain in torch fp32: is the bf16 hi+lo split of dR (16-17 bits) visible there?"""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import seeded_inputs as SI
import neural_admixture_amd as na
from neural_admixture_amd.svd import RSVD
d = np.load("/root/repo/tests/golden/c2_end_to_end.npz")
N, M, K, C = int(d["N"]), int(d["M"]), int(d["K"]), int(d["C"])
Gn = SI.genotypes(N, M, K, int(d["seed"]), threads=32)
dev = torch.device("cuda:0")
Vt = RSVD(torch.from_numpy(Gn), N, M, C, int(d["run_seed"]), device=dev)
for tag in ("gmm_init", "random_init"):
    if tag == "gmm_init":
        P0 = np.clip(d["gmm_means"] @ Vt, 5e-6, 1 - 5e-6).astype(np.float32); V0 = np.ascontiguousarray(Vt.T)
    else:
        V0, P0 = SI.init_v_p(M, C, K, 2026)
    b = 800
    e = na.Engine(M, C, 1024, [K], dev, b)
    e.load_params(V0, P0, na.model.init_encoder_weights(42, C, 1024, [K]))
    e.pack_from_host(torch.from_numpy(Gn[:b]))
    idx = torch.arange(b, dtype=torch.int32, device=dev)
    for it in range(2):                      # step 0 from the init, then after one Adam step + restrict_P
        e.forward(idx, b); e.backward(idx, b, True); torch.cuda.synchronize()
        L = e.lay
        Q = e.Q[: b * L.SP].view(b, L.SP)[:, :K].contiguous()
        Pd = e.P(0).contiguous()
        X = torch.from_numpy(np.where(Gn[:b] == 3, 0, Gn[:b]).astype(np.float32) / 2).to(dev)
        def chain(dt):
            Qd, Pp, Xd = Q.to(dt), Pd.to(dt), X.to(dt)
            Rraw = Qd @ Pp.T
            R = Rraw.clamp(0, 1)
            dR = (R - Xd) / ((1 - R) * R).clamp_min(1e-12) * ((Rraw >= 0) & (Rraw <= 1))
            return (dR.T @ Qd).double(), (dR @ Pp).double()
        dP64, dQ64 = chain(torch.float64)
        dP32, dQ32 = chain(torch.float32)
        kp = L.kp[0]
        chunks = int(L.dec_chunks[0])
        dq_h = e.dqpart[: chunks * b * kp].view(chunks, b, kp).double().sum(0)[:, :K]
        gP = e.gP(0).double()
        r = lambda a, ref: ((a - ref).abs().max() / ref.abs().max()).item()
        r2 = lambda a, ref: ((a - ref).pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()).item()
        print(f"{tag} step {it}: dP  hip max-rel {r(gP, dP64):.2e} l2-rel {r2(gP, dP64):.2e} | torch fp32 {r(dP32, dP64):.2e} {r2(dP32, dP64):.2e}")
        print(f"{tag} step {it}: dQ  hip max-rel {r(dq_h, dQ64):.2e} l2-rel {r2(dq_h, dQ64):.2e} | torch fp32 {r(dQ32, dQ64):.2e} {r2(dQ32, dQ64):.2e}   (|dQ|max {dQ64.abs().max().item():.3e}, frac P at clip {(Pd <= 5.1e-6).float().mean().item():.3f})")
        del dP64, dQ64, dP32, dQ32, X
        e.adam(2e-3)
    del e
