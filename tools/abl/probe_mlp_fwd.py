import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["NADM_LIB"] = os.path.join(os.getcwd(), "tools/abl/probe.so")
import neural_admixture_amd as na
from neural_admixture_amd.model import init_encoder_weights
dev = torch.device("cuda:0")
M, K, b = 500_000, 8, 800
e = na.Engine(M, 8, 1024, [K], dev, b)
xp = torch.randint(0, 255, (2000, e.ld), dtype=torch.uint8, device=dev)
e.set_packed(xp)
rng = np.random.default_rng(0)
e.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(0.01, 0.99, (K, M)).astype(np.float32), init_encoder_weights(1, 8, 1024, [K]))
idx = torch.arange(b, dtype=torch.int32, device=dev)
e.fused_adam = False
for it in range(30):
    e.forward(idx, b); e.backward(idx, b, True)
e.forward(idx, b)
torch.cuda.synchronize()
t = e.H[400 * 1024: 400 * 1024 + 8].cpu().numpy().astype(np.int64)
print("fwd: weights+bias fetched(0), Z reduce(1), rmsnorm(2), hidden(3), logits incl Wk(4), -, -, end(7):", t)
