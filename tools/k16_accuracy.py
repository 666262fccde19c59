#!/usr/bin/env python3
"""dP / dQ-slab-sum / loss error of pass 2's two-k-slot variant (K = 9..16) against float64 on the device, and its time: for A/B builds
(NADM_LIB=tools/abl/<name>.so).  Usage: k16_accuracy.py [K=16] [b=800] [M=200000]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_admixture_amd as na
from neural_admixture_amd._lib import lib, check, ptr
K, b, M = (int(a) for a in (sys.argv[1:4] + ["16", "800", "200000"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
e = na.Engine(M, 8, 1024, [K], dev, b)
g = torch.Generator(device="cpu").manual_seed(0)
Qt = torch.distributions.Dirichlet(torch.full((K,), 0.2)).sample((b,)).float().to(dev)
Fq = (0.5 * torch.rand(K, M, generator=g)).clamp(0.005, 0.5).to(dev)
xp = torch.empty((b, e.ld), dtype=torch.uint8, device=dev)
check(lib.nadm_synth_packed(ptr(xp), b, 0, M, e.ld, ptr(Qt), ptr(Fq), K, 0.01, 1234, None))
e.set_packed(xp)
Gd = torch.empty((b, M), dtype=torch.uint8, device=dev)
check(lib.nadm_unpack2bit(ptr(xp), ptr(Gd), b, M, e.ld, None))
X = torch.where(Gd == 3, torch.zeros((), device=dev), Gd.float() / 2).double()
V = (torch.randn(M, 8, generator=g) / M ** 0.5).numpy()
P = torch.rand(K, M, generator=g).mul(0.9).add(0.05).numpy()
e.load_params(V, P, na.model.init_encoder_weights(42, 8, 1024, [K]))
idx = torch.arange(b, dtype=torch.int32, device=dev)
e.forward(idx, b); e.backward(idx, b, True); torch.cuda.synchronize()
Pd, Q = e.P(0).double(), e.Q[: b * e.lay.SP].view(b, e.lay.SP)[:, :K].double()
R = Q @ Pd.T
dR = (R - X) / ((1 - R) * R).clamp_min(1e-12) * ((R >= 0) & (R <= 1))
dP_ref, dQ_ref = dR.T @ Q, dR @ Pd
L = e.lay
dq = e.dqpart[: L.dec_chunks[0] * b * L.kp[0]].view(L.dec_chunks[0], b, L.kp[0]).double().sum(0)[:, :K]
# (mlp_backward folded tall slabs in place: recompute the slab with a fresh pass 2)
e.decode_all(idx, b, True); torch.cuda.synchronize()
dq = e.dqpart[: L.dec_chunks[0] * b * L.kp[0]].view(L.dec_chunks[0], b, L.kp[0]).double().sum(0)[:, :K]
err_p = float((e.gP(0).double() - dP_ref).abs().max() / dP_ref.abs().max())
err_q = float((dq - dQ_ref).abs().max() / dQ_ref.abs().max())
rms_p = float(((e.gP(0).double() - dP_ref) ** 2).mean().sqrt() / dP_ref.abs().max())
e.time_kernels(("decode_bce",))
for _ in range(60):
    e.train_step(idx, b, 2e-3, True)
ms = e.kernel_ms()["decode_bce"]
print(f"lib {os.environ.get('NADM_LIB', 'default')}: K={K} b={b} M={M}: max|ddP|/max {err_p:.3e} rms {rms_p:.3e}  max|ddQ|/max {err_q:.3e}  pass 2 {ms * 1e3:.1f} us")
