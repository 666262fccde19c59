"""Why do epochs inside train() take 45..75 ms at c4 when 125 back-to-back steps take 44.5?"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neural_admixture_amd.engine import Engine
from neural_admixture_amd.layout import ModelLayout
from neural_admixture_amd.model import epoch_order, _EpochOrders
dev = torch.device("cuda:0")
N, M, K = 100000, 500000, 8
ld = ModelLayout.row_stride(M)
xp = torch.randint(0, 85, (N, ld), dtype=torch.uint8, device=dev)
eng = Engine(M, 8, 1024, [K], dev, max_batch=800)
eng.set_packed(xp)
rng = np.random.default_rng(0)
L = eng.lay
eng.load_params((rng.standard_normal((M, 8)) * 0.01).astype(np.float32), rng.uniform(0.1, 0.9, (K, M)).astype(np.float32),
                (rng.standard_normal(L.n_small) * 0.01).astype(np.float32))
idx = torch.randperm(N, device=dev).to(torch.int32)
gen = torch.Generator().manual_seed(1)

def run(name, epochs, order_fn, sync_every, loss_every=0, sync_epoch=False):
    torch.cuda.synchronize()
    marks = []
    t0 = time.perf_counter()
    host = []
    for e in range(epochs):
        th = time.perf_counter()
        wl = loss_every > 0 and e % loss_every == 0
        order = order_fn(e)
        for s in range(0, N, 800):
            eng.train_step(order[s:s + 800], 800, 2e-3, wl)
        after(e)
        host.append((time.perf_counter() - th) * 1e3)
        if wl:
            eng.read_loss(reset=True)
        elif sync_epoch:
            torch.cuda.synchronize()
        if (e + 1) % sync_every == 0:
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / epochs * 1e3
    d = np.diff([t0] + marks) * 1e3 / sync_every
    print(f"{name}: {tot:.1f} ms/epoch; per-sync-interval {np.round(d, 1).tolist()}")
    print(f"    host ms/epoch to queue: {np.round(host, 1).tolist()}")

after = lambda e: None
run("fixed order, sync at end", 40, lambda e: idx, 40)
run("fixed order, loss every 5th", 40, lambda e: idx, 40, 5)
orders = _EpochOrders(gen, N, dev)
after = lambda e: orders.epoch_queued()
k = [0]
def nxt(e):
    k[0] += 1
    return orders.take(k[0] - 1, True)
run("prefetched orders, sync at end", 40, nxt, 40)
run("prefetched orders, loss every 5th (read_loss)", 40, nxt, 40, 5)
import time as _t
for name, fn in (("randperm(out=pinned)", lambda: torch.randperm(N, generator=gen, out=orders.pin[0])),
                 ("randperm(out=pageable)", lambda: torch.randperm(N, generator=gen, out=orders.second)),
                 ("randperm()", lambda: torch.randperm(N, generator=gen)),
                 ("epoch_order()", lambda: epoch_order(gen, N))):
    fn(); t = _t.perf_counter()
    for _ in range(20): fn()
    print(f"{name}: {(_t.perf_counter() - t) / 20 * 1e3:.2f} ms")
