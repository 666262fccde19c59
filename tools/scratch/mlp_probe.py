"""r03: phase stamps (s_memtime, shader cycles) of one block of the MLP forward inside real training steps.  NADM_LIB must point at a
build with -DNADM_MLP_PROBE (tools/build_variant.sh mlpprobe -DNADM_MLP_PROBE)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neural_admixture_amd as na
from neural_admixture_amd import _lib
from neural_admixture_amd.model import init_encoder_weights
raw = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
M, K, b, rows = 500_000, 8, 800, 8000
eng = na.Engine(M, 8, 1024, [K], dev, b)
xp = torch.randint(0, 85, (rows, eng.ld), dtype=torch.uint8, device=dev)
eng.set_packed(xp)
rng = np.random.default_rng(0)
eng.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(0.05, 0.95, size=(K, M)).astype(np.float32),
                init_encoder_weights(42, 8, 1024, [K]))
perm = torch.randperm(rows).to(torch.int32).to(dev)
names = {15: "start", 0: "zpart + weight loads landed, partials in LDS", 1: "Z combined", 2: "RMSNorm done", 3: "head-column products done (s_red free)",
         4: "wave sums in LDS", 5: "logits done", 14: "softmax + stores issued"}
acc = []
for it in range(30):
    eng.train_step(perm[(it % 10) * b:(it % 10 + 1) * b], b, 2e-3, True)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert raw.nadm_probe_read(buf) == 0
    acc.append([buf[i] for i in range(16)])
a = np.asarray(acc[10:], dtype=np.float64)
order = [15, 0, 1, 2, 3, 4, 5, 14]
t0 = a[:, 15]
prev = t0
print("MLP forward, block 40, thread 0: cumulative / phase cycles (median of 20 steps); 2400 cycles = 1 us at 2.4 GHz")
for i in order[1:]:
    print(f"  {names[i]:50s} {np.median(a[:, i] - t0):9.0f}   +{np.median(a[:, i] - prev):8.0f}")
    prev = a[:, i]
