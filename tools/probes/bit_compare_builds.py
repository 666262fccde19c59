"""Same steps under two builds of the library (NADM_LIB): prints a checksum of every parameter after 6 steps on a few shapes."""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, "/root/repo")
import neural_admixture_amd as na
from oracle import nadm_oracle as O
dev = torch.device("cuda:0")
for (N, M, ks, b) in ((900, 70_001, [8], 800), (300, 5003, [3], 100), (500, 20_000, [2, 5, 8], 333), (4500, 20_000, [7], 4301), (4400, 9_000, [2, 3, 9], 4099)):
    G = O.synth_genotypes(N, M, max(ks), seed=5, missing=0.02)
    rng = np.random.default_rng(1)
    V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, (sum(ks), M)).astype(np.float32)
    e = na.Engine(M, 8, 256, ks, dev, b)
    e.load_params(V0, P0, na.model.init_encoder_weights(3, 8, 256, ks))
    e.pack_from_host(torch.from_numpy(G))
    idx = torch.from_numpy(rng.permutation(N).astype(np.int32)).to(dev)
    for s in range(6):
        bb = b if s != 3 else b - 37
        e.train_step(idx[:bb], bb, 2e-3, True)
    e.sync(); torch.cuda.synchronize()
    print(N, M, ks, b, hashlib.sha256(e.pflat.cpu().numpy().tobytes()).hexdigest()[:16], e.read_loss()[1])
