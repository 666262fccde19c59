import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["NADM_LIB"] = os.path.join(os.getcwd(), "tools/abl/probe2.so")
import neural_admixture_amd as na
from neural_admixture_amd.model import init_encoder_weights
dev = torch.device("cuda:0")
M, K, b = 500_000, 8, 800
e = na.Engine(M, 8, 1024, [K], dev, b)
xp = torch.randint(0, 255, (20000, e.ld), dtype=torch.uint8, device=dev)
e.set_packed(xp)
rng = np.random.default_rng(0)
e.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(0.01, 0.99, (K, M)).astype(np.float32), init_encoder_weights(1, 8, 1024, [K]))
idx = torch.randperm(20000)[:b].to(torch.int32).to(dev)
for fused in (None, (2e-3, 1.0)):
    for it in range(10):
        e.forward(idx, b)
        e.step_count += 1
        e.decode_all(idx, b, True, fused_adam=fused)
    torch.cuda.synchronize()
    L = e.lay
    for chunk in (300, 1100, 1800):
        base = chunk * b * L.kp[0]
        t = e.dqpart[base: base + 3].cpu().numpy().astype(np.int64)
        print(f"{'adam epilogue' if fused else 'dP write    '} chunk {chunk}: prologue {t[0]}  tile loop {t[1]}  epilogue {t[2]} cycles")
