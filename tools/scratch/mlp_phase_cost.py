"""r03: where do the MLP kernels spend their 12 / 21 us at b = 800?  Times nadm_mlp_fwd_images / nadm_mlp_bwd back to back (200 launches
between two HIP events) in the bench shape and with single phases taken away through the arguments: the partial-slab walks (n_chunks = 1,
dq M = 1 chunk), fewer samples (b = 400 / 200 / 100 -> 100 / 50 / 25 blocks), a narrower hidden layer."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neural_admixture_amd as na
from neural_admixture_amd._lib import lib, check, ptr
from neural_admixture_amd.model import init_encoder_weights
dev = torch.device("cuda:0")

def bench(M, b, Hd, K=8, n=200):
    eng = na.Engine(M, 8, Hd, [K], dev, b)
    rows = 2 * b
    xp = torch.randint(0, 85, (rows, eng.ld), dtype=torch.uint8, device=dev)
    eng.set_packed(xp)
    rng = np.random.default_rng(0)
    eng.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(0.05, 0.95, size=(K, M)).astype(np.float32),
                    init_encoder_weights(42, 8, Hd, [K]))
    idx = torch.arange(b, dtype=torch.int32, device=dev)
    eng.forward(idx, b); eng.backward(idx, b, True); torch.cuda.synchronize()
    L = eng.lay
    def t(fn):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    fwd = t(lambda: eng.mlp_forward(b))
    fwd1 = t(lambda: eng.mlp_forward(b, eng.zpart, 1))
    bwd = t(lambda: eng.mlp_backward(b, L.n_loss, weights=False))
    bwd0 = t(lambda: eng.mlp_backward(b, 0, weights=False))
    bwd1 = t(lambda: eng.mlp_backward(b, 0, dq_src=eng.dqpart, dq_M=1, weights=False))
    print(f"M={M:7d} b={b:4d} Hd={Hd:4d}: fwd {fwd:5.1f} us (1 chunk: {fwd1:5.1f})   bwd {bwd:5.1f} us (no loss block: {bwd0:5.1f}; 1 slab row: {bwd1:5.1f})", flush=True)

for args in ((500_000, 800, 1024), (500_000, 400, 1024), (500_000, 200, 1024), (500_000, 100, 1024), (500_000, 800, 256), (62_500, 800, 1024), (62_500, 6400, 1024)):
    bench(*args)
# an empty launch of the same grid as a floor
x = torch.zeros(200 * 256, device=dev)
def t_empty():
    for _ in range(20): x.add_(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): x.add_(0)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200
print(f"back-to-back trivial torch kernel on 51200 floats: {t_empty():.1f} us per launch")
