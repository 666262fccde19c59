"""Pass 3 at an SNP-sharded rank's shape, forced slice counts (test build): kernel time of encode_bwd from the plan's HIP events."""
import os, sys, numpy as np, torch
os.environ.setdefault("NADM_LIB", "/root/repo/neural-admixture_amd/csrc/libnadm_testhooks.so")
sys.path.insert(0, "/root/repo")
import neural_admixture_amd as na
from neural_admixture_amd._lib import lib, check, ptr
dev = torch.device("cuda:0")
for (rows, M, b) in ((12800, 62_500, 6400), (12800, 125_000, 3200), (8000, 62_500, 800)):
    for force in (1, 2, 3, 4, 8, 0):
        lib.nadm_test_force_p3_slices(force)
        e = na.Engine(M, 8, 1024, [8], dev, b)
        K = 8
        Qt = torch.distributions.Dirichlet(torch.full((K,), 0.2)).sample((rows,)).float().to(dev)
        Fq = (0.5 * torch.rand(K, M)).clamp(0.005, 0.5).to(dev)
        xp = torch.empty((rows, e.ld), dtype=torch.uint8, device=dev)
        check(lib.nadm_synth_packed(ptr(xp), rows, 0, M, e.ld, ptr(Qt), ptr(Fq), K, 0.01, 1234, None))
        e.set_packed(xp)
        rng = np.random.default_rng(42)
        e.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(5e-6, 1 - 5e-6, size=(8, M)).astype(np.float32),
                      na.model.init_encoder_weights(42, 8, 1024, [8]))
        perm = torch.randperm(rows).to(torch.int32).to(dev)
        for s in range(30):
            e.train_step(perm[(s % 2) * b:(s % 2) * b + b], b, 2e-3, True)
        e.time_kernels(("encode_bwd", "decode_bce", "mlp_fwd", "mlp_bwd", "encode_fwd"))
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for s in range(60):
            e.train_step(perm[(s % 2) * b:(s % 2) * b + b], b, 2e-3, True)
        e.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 60
        k = e.kernel_ms()
        print(f"rows/step {b} M {M} force {force} -> slices {lib.nadm_encode_slices(b, M, 8)}: step {dt*1e3:.4f} ms  " + " ".join(f"{n}={v*1e3:.1f}" for n, v in k.items()), flush=True)
        del e
    lib.nadm_test_force_p3_slices(0)
