"""Host cost of queueing one train_step (no GPU wait): how far ahead of the kernels can the Python loop run?"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cProfile, pstats
from neural_admixture_amd.engine import Engine
from neural_admixture_amd.layout import ModelLayout
dev = torch.device("cuda:0")
N, M, K = 20000, 500000, 8
ld = ModelLayout.row_stride(M)
xp = torch.randint(0, 85, (N, ld), dtype=torch.uint8, device=dev)   # codes 0..2 only: 85 = 0b01010101
eng = Engine(M, 8, 1024, [K], dev, max_batch=800)
eng.set_packed(xp)
rng = np.random.default_rng(0)
L = eng.lay
eng.load_params((rng.standard_normal((M, 8)) * 0.01).astype(np.float32), rng.uniform(0.1, 0.9, (K, M)).astype(np.float32),
                (rng.standard_normal(L.n_small) * 0.01).astype(np.float32))
idx = torch.randperm(N, device=dev).to(torch.int32)
for s in range(0, 8000, 800):
    eng.train_step(idx[s:s + 800], 800, 2e-3, False)
torch.cuda.synchronize()
for with_loss in (False, True):
    n = 40                                    # few enough that the launch queue never fills
    t = time.perf_counter()
    for i in range(n):
        s = (i * 800) % (N - 800)
        eng.train_step(idx[s:s + 800], 800, 2e-3, with_loss)
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    tg = time.perf_counter() - t
    print(f"with_loss={with_loss}: host {th / n * 1e3:.3f} ms/step to queue, {tg / n * 1e3:.3f} ms/step done")
pr = cProfile.Profile(); pr.enable()
for i in range(40):
    s = (i * 800) % (N - 800)
    eng.train_step(idx[s:s + 800], 800, 2e-3, False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
