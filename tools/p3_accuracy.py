#!/usr/bin/env python3
"""Accuracy of pass 3 (dV = X^T dZ on the FP4 x FP6 matrix instruction) against a float64 product, next to what a plain fp32 matmul of
the same operands loses: max and rms error relative to max |dV|, for dZ whose rows span several orders of magnitude."""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nadm_oracle as O
import neural_admixture_amd as na
from neural_admixture_amd._lib import lib, check, ptr

dev = torch.device("cuda:0")
out = {}
for spread in (0.0, 2.0, 6.0):
    rng = np.random.default_rng(int(spread) + 1)
    N, M, Cc = 800, 65536, 8
    Gm = O.synth_genotypes(N, M, 4, seed=3, missing=0.02)
    e = na.Engine(M, Cc, 64, [5], dev, N)
    e.pack_from_host(torch.from_numpy(np.ascontiguousarray(Gm)))
    dZ = (rng.standard_normal((N, Cc)) * np.exp2(rng.uniform(-spread, spread, size=(N, 1)))).astype(np.float32)
    e.dZ[: N * Cc] = torch.from_numpy(dZ.reshape(-1)).to(dev)
    idx = torch.arange(N, dtype=torch.int32, device=dev)
    e.encode_backward(idx, N)
    torch.cuda.synchronize()
    got = e.gV().cpu().numpy().astype(np.float64)
    X = np.where(Gm == 3, 0, Gm).astype(np.float64) / 2
    ref = X.T @ dZ.astype(np.float64)
    f32 = (torch.from_numpy(X.astype(np.float32)).to(dev).T @ torch.from_numpy(dZ).to(dev)).cpu().numpy().astype(np.float64)
    sc = np.abs(ref).max()
    out[f"row magnitudes 2^+-{spread:g}"] = {"max_err_pass3": float(np.abs(got - ref).max() / sc), "rms_err_pass3": float(np.sqrt(((got - ref) ** 2).mean()) / sc),
                                               "max_err_fp32_matmul": float(np.abs(f32 - ref).max() / sc), "rms_err_fp32_matmul": float(np.sqrt(((f32 - ref) ** 2).mean()) / sc)}
print(json.dumps(out))
