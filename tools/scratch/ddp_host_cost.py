"""r03: is the data-parallel step on a 1-rank RCCL group bound by the host?  Host time to QUEUE one step (no GPU wait) against the
time the GPU needs for it, for the plain step, the data-parallel step, and the data-parallel step with fewer / no collectives."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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

def run(name, fn, n=40, warm=30):
    for s in range(warm):
        fn(perm[(s % 10) * b:(s % 10 + 1) * b])
    torch.cuda.synchronize(); t = time.perf_counter()
    for s in range(n):
        fn(perm[(s % 10) * b:(s % 10 + 1) * b])
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    tg = time.perf_counter() - t
    print(f"{name:44s} host {th / n * 1e3:.3f} ms/step to queue, {tg / n * 1e3:.3f} ms/step done", flush=True)

def plain(idx): eng.train_step(idx, b, 2e-3, True)
def ddp(idx): eng.train_step_ddp(idx, b, 2e-3, 1, True, defer_tail=True)
for _ in range(2):
    run("plain", plain); run("ddp (1-rank group), deferred updates", ddp); eng.finish_ddp()
# the same step with dist.all_reduce replaced by a no-op: what the three collective calls cost
real = dist.all_reduce
class _W:
    def wait(self): pass
dist.all_reduce = lambda *a, **k: (_W() if k.get("async_op") else None)
run("ddp, all_reduce stubbed out", ddp); eng.finish_ddp()
dist.all_reduce = real
pr = cProfile.Profile(); pr.enable()
for i in range(40): ddp(perm[(i % 10) * b:(i % 10 + 1) * b])
pr.disable(); torch.cuda.synchronize(); eng.finish_ddp()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
