#!/usr/bin/env python3
"""One GPU: cost of the data-parallel step's launch plan without any communication partner.
(a) plain train_step; (b) same kernels with pass 2 cut at the round boundary (no all-reduce); (c) train_step_ddp on a
1-rank RCCL group (sub-range launches + all-reduce calls + stream waits)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import neural_admixture_amd as na
from neural_admixture_amd.model import init_encoder_weights
from neural_admixture_amd._lib import lib, check, ptr

dev = torch.device("cuda:0")
M, K, b, rows = 500_000, 8, 800, 8000
eng = na.Engine(M, 8, 1024, [K], dev, b)
xp = torch.empty((rows, eng.ld), dtype=torch.uint8, device=dev)
Qt = torch.distributions.Dirichlet(torch.full((K,), 0.2)).sample((rows,)).float().to(dev)
Fq = torch.rand((K, M), device=dev) * 0.5
check(lib.nadm_synth_packed(ptr(xp), rows, 0, M, eng.ld, ptr(Qt), ptr(Fq), K, 0.01, 1, None))
eng.set_packed(xp)
rng = np.random.default_rng(0)
eng.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(5e-6, 1 - 5e-6, size=(K, M)).astype(np.float32),
                init_encoder_weights(42, 8, 1024, [K]))
perm = torch.randperm(rows).to(torch.int32).to(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)

def run(name, fn, steps=100, warm=10):
    for s in range(warm):
        fn(perm[(s % 10) * b:(s % 10 + 1) * b])
    torch.cuda.synchronize(); t = time.perf_counter()
    for s in range(steps):
        fn(perm[(s % 10) * b:(s % 10 + 1) * b])
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t) / steps * 1e3:.4f} ms/step", flush=True)

def plain(idx): eng.train_step(idx, b, 2e-3, True)
def split(idx):
    eng.forward(idx, b); eng.backward(idx, b, True, p_parts="rounds"); eng.adam(2e-3)
def cb_only(idx):
    eng.forward(idx, b); eng.backward(idx, b, True, on_grad_ready=lambda lo, hi: None, p_parts="rounds"); eng.adam(2e-3)
def ddp(idx): eng.train_step_ddp(idx, b, 2e-3, 1, True)
def ddp_defer(idx): eng.train_step_ddp(idx, b, 2e-3, 1, True, defer_tail=True)
def ddp_nosplit(idx):
    works = []
    eng.forward(idx, b)
    eng.backward(idx, b, True, on_grad_ready=lambda lo, hi: works.append(dist.all_reduce(eng.gflat[lo:hi], async_op=True)), p_parts=1)
    for w in works: w.wait()
    eng.adam(2e-3)
for _ in range(2):
    run("(a) plain", plain); run("(b) pass 2 cut at round boundary", split); run("(c) ddp step, 1-rank group", ddp); run("(c') ddp step, deferred last P piece", ddp_defer); eng.finish_ddp(); run("(d) ddp, pass 2 uncut", ddp_nosplit)
dist.destroy_process_group()
