#!/usr/bin/env python3
"""Headline benchmark: genotypes/s (samples x SNPs per second of training) at K=8.

A "step" is one training step of the hot path on one batch: gather 800 rows of the 2-bit packed
matrix, encoder X.V, RMSNorm+MLP+softmax, decoder Q.P^T + clamp + BCE forward/backward (loss value
included), MLP backward, dV = X^T.dZ, gradient all-reduce (N>1), Adam on every parameter, P clamp.
Workload (BASELINE.json configs[3]): synthetic 100k samples x 500k SNPs, K=8, resident 2-bit packed
in HBM (12.5 GB; sharded by samples over ranks), batch 800 PER GPU (weak scaling: the reference's
global-batch-800 semantics would leave 100 rows per GPU at N=8).

    python bench.py --gpus 1 --steps 100 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak (same guide); the three GEMM-shaped products run as bf16 pieces


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)   # the first ~25 launches after start-up run ~7 % slow (clock ramp)
    ap.add_argument("--rows", type=int, default=100_000, help="total samples N (sharded over ranks)")
    ap.add_argument("--snps", type=int, default=500_000)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--batch", type=int, default=800, help="rows per step PER GPU")
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--no-loss", action="store_true", help="skip the loss value (gradients unchanged); default computes it every step like the reference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallelism", choices=("dp", "snp"), default="dp",
                    help="dp (default, the reference's scheme): samples sharded, gradient all-reduce; snp: SNPs sharded, every rank "
                         "processes the global batch of batch*N rows on its M/N SNPs, two small all-reduces per step")
    ap.add_argument("--force-ddp", action="store_true", help="single GPU: run the data-parallel step (sub-range launches + RCCL all-reduce on a 1-rank group)")
    ap.add_argument("--time-kernels", choices=("dominant", "all"), default="dominant",
                    help="HIP events inside the timed region around the dominant kernel only (default: two event records per step) "
                         "or around every kernel of the step (kernel_ms table; the records cost ~10 %% of the step)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional check of the N>1 flow on a ONE-GPU box: all ranks use cuda:0 and gloo carries the tensors "
                         "(RCCL refuses two ranks per device); not a measurement")
    ap.add_argument("--cpu-rows", type=int, default=2400, help="rows of the same workload used for the bounded CPU baseline")
    return ap.parse_args()


def make_dataset(eng, rows_local, row0, K, dev, seed=1234):
    """Admixture-model synthetic genotypes generated on the device straight into packed bytes."""
    from neural_admixture_amd._lib import lib, check, ptr
    g = torch.Generator(device="cpu").manual_seed(seed)
    beta = torch.distributions.Beta(torch.tensor(0.5), torch.tensor(0.5))
    torch.manual_seed(seed)
    Fq = (0.5 * beta.sample((K, eng.M))).clamp(0.005, 0.5).float().to(dev)           # same on every rank
    xp = torch.empty((rows_local, eng.ld), dtype=torch.uint8, device=dev)
    dirich = torch.distributions.Dirichlet(torch.full((K,), 0.2))
    chunk = 16384
    for s in range(0, rows_local, chunk):
        n = min(chunk, rows_local - s)
        torch.manual_seed(seed + 1 + (row0 + s) // chunk)
        Qt = dirich.sample((n,)).float().to(dev)
        check(lib.nadm_synth_packed(ptr(xp[s:]), n, row0 + s, eng.M, eng.ld, ptr(Qt), ptr(Fq), K, 0.01, seed, None), "synth")
    torch.cuda.synchronize()
    del g
    return xp


def cpu_baseline(eng, args, dev):
    """The C port of the step (oracle/nadm_oracle_c.c, OpenMP) on a bounded row sample of the same matrix."""
    from neural_admixture_amd._lib import lib, check, ptr
    from oracle.c_port import CPort
    L = eng.lay
    n = min(args.cpu_rows, eng.xp.shape[0])
    b = min(args.batch, n)
    Gd = torch.empty((n, L.M), dtype=torch.uint8, device=dev)
    check(lib.nadm_unpack2bit(ptr(eng.xp), ptr(Gd), n, L.M, eng.ld, None), "unpack")
    G = Gd.cpu().numpy()
    del Gd
    sm = eng.small.cpu().numpy()
    h = L.heads
    K = L.ks[0]
    cp = CPort(eng.V().cpu().numpy(), eng.P(0).cpu().numpy(), sm[h.g_off:h.g_off + L.C], sm[h.w1_off:h.w1_off + L.Hd * L.C].reshape(L.Hd, L.C),
               sm[h.b1_off:h.b1_off + L.Hd], sm[h.wk_off[0]:h.wk_off[0] + K * L.Hd].reshape(K, L.Hd), sm[h.bk_off[0]:h.bk_off[0] + K])
    idx = np.arange(n)
    cp.step(G, idx[:b], 2e-3)                      # warm-up step (page-in, thread pool)
    nsteps = max(1, n // b)
    t0 = time.perf_counter()
    for s in range(nsteps):
        cp.step(G, idx[s * b:(s + 1) * b], 2e-3)
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": nsteps * b * L.M / dt, "unit": "genotypes/s", "cores": cp.threads(), "kind": "port",
            "sample": f"{nsteps} steps of {b} rows x {L.M} SNPs (first {n} rows of the same matrix), fused C/OpenMP port of the step, cpu={model}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_ddp or args.parallelism == "snp":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner through C stdio on stdout; keep stdout for the one JSON line: send fd 1 to stderr
        # while the communicator is created (eager with device_id=) and flush the C buffers before switching back
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.share_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), rank=rank, world_size=world)
            t_ = torch.zeros(1, device=f"cuda:{local_rank}")
            dist.all_reduce(t_)
            torch.cuda.synchronize()
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    import neural_admixture_amd as na
    from neural_admixture_amd.model import init_encoder_weights

    M, K, b = args.snps, args.k, args.batch
    snp = args.parallelism == "snp"
    rng = np.random.default_rng(42)                                     # identical parameters on every rank
    V0 = (0.01 * rng.standard_normal((M, 8))).astype(np.float32)
    P0 = rng.uniform(5e-6, 1 - 5e-6, size=(K, M)).astype(np.float32)
    if snp:
        # every rank: ALL rows of its SNP slice (same bytes in HBM per rank as the sample-sharded layout), global batch b*world
        from neural_admixture_amd.snp_parallel import SnpShardedEngine
        rows_local, gb = args.rows, b * world
        eng = SnpShardedEngine(M, 8, args.hidden, [K], dev, gb, rank, world)
        eng.set_packed(make_dataset(eng, rows_local, 0, K, dev, seed=1234 + 7 * rank))
        gperm = torch.Generator(device="cpu").manual_seed(1000)         # the same global batches on every rank
    else:
        rows_local, gb = args.rows // world, b
        eng = na.Engine(M, 8, args.hidden, [K], dev, b)
        eng.set_packed(make_dataset(eng, rows_local, rank * rows_local, K, dev))
        gperm = torch.Generator(device="cpu").manual_seed(1000 + rank)
    eng.load_params(V0, P0, init_encoder_weights(42, 8, args.hidden, [K]))
    del V0, P0
    perm = torch.randperm(rows_local, generator=gperm).to(torch.int32).to(dev)
    nb = rows_local // gb
    with_loss = not args.no_loss
    lr = 2e-3

    def step(s):
        o = (s % nb) * gb
        if snp:
            eng.train_step(perm[o:o + gb], gb, lr, with_loss)
        elif world > 1 or args.force_ddp:
            eng.train_step_ddp(perm[o:o + b], b, lr, world, with_loss, defer_tail=True)
        else:
            eng.train_step(perm[o:o + b], b, lr, with_loss)

    for s in range(args.warmup):
        step(s)
    eng.timers = {}
    eng.timed_names = None if args.time_kernels == "all" else {"decode_bce"}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s)
    if not snp and (world > 1 or args.force_ddp):
        eng.finish_ddp()                                   # the last step's deferred P piece belongs to the timed work
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timers, eng.timers = eng.timers, None
    loss_sum, loss_last = eng.read_loss()
    assert np.isfinite(loss_last) or not with_loss

    kms = {name: float(np.mean([a.elapsed_time(c) for a, c in evs])) for name, evs in timers.items()}
    if args.time_kernels == "dominant":
        # the other kernels of the step, for the kernel_ms table only: a short pass AFTER the timed region with events around
        # every kernel (those records cost ~4 % of the step, which is why the timed region carries only the dominant kernel's)
        eng.timers, eng.timed_names = {}, None
        for s in range(min(args.steps, 20)):
            step(args.warmup + args.steps + s)
        if not snp and (world > 1 or args.force_ddp):
            eng.finish_ddp()
        torch.cuda.synchronize()
        extra, eng.timers = eng.timers, None
        for name, evs in extra.items():
            if name != "decode_bce":
                kms[name] = float(np.mean([a.elapsed_time(c) for a, c in evs]))
    # dominant kernel = decode_bce.  Algorithmic bytes per launch: one 2-bit pass over the
    # batch (b*M/4) + read P and write dP once (2 * 4*M*K).
    dom = "decode_bce"
    # Algorithmic bytes of one pass-2 launch, in the accounting of SURVEY.md 8d (whole step = 0.75 B/genotype of packed X +
    # 36 B per parameter of parameter/optimizer traffic): one 2-bit pass over the batch + the pass's share of the
    # per-parameter traffic.  Single-GPU and SNP-sharded steps run Adam on the P rows in the kernel's epilogue, so the launch
    # carries ALL 36 B of a P parameter (read p; write/read g; read/write p, m, v); in the data-parallel step Adam is a
    # launch of its own behind the all-reduce and pass 2 keeps only read p + write g = 8 B.
    fused = (snp or not (world > 1 or args.force_ddp)) and getattr(eng, "fused_adam", False)
    rows_b, m_loc = (gb, eng.M) if snp else (b, M)                       # snp: global batch x own SNP slice
    alg_bytes = rows_b * m_loc / 4 + (36 if fused else 8) * m_loc * K
    traffic = None                                                      # HBM bytes/launch from the committed PMC passes (same workload)
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")) as f:
            pm = json.load(f)
        if abs(pm["algorithmic_bytes_per_launch"] - alg_bytes) < 1:
            traffic = pm["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    achieved = alg_bytes / (kms[dom] * 1e-3) / 1e9
    step_bytes = b * M * (0.75 + 36.0 * (8 + K) / b)                     # SURVEY.md 8d whole-step figure
    cfg_name = {(100_000, 500_000, 8): "configs[3]", (500_000, 1_000_000, 16): "configs[4]"}.get((args.rows, M, K), "configs[3] shape, resized")
    out = {
        "metric": "genotypes/sec (samples x SNPs / epoch-sec) at K=%d" % K,
        "value": world * b * M * args.steps / dt, "unit": "genotypes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: synthetic {args.rows} samples x {M} SNPs, K={K}, 2-bit packed resident in HBM, "
                               f"{'SNP' if snp else 'sample'}-sharded over {world} GPU(s), batch {b}/GPU, hidden {args.hidden}, n_components 8, "
                               f"loss value {'every step' if with_loss else 'skipped'}",
                   "global_batch": b * world, "parallelism": f"{args.parallelism}{world}"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel_ms": kms, "alg_bytes_per_launch": alg_bytes,
                     "alg_bytes_note": "2-bit pass over the batch + %d B per P parameter (%s)" % (
                         36 if fused else 8, "Adam on P fused into the launch" if fused else "read P, write dP; Adam is a separate launch"),
                     "whole_step": {"alg_bytes": step_bytes, "achieved": step_bytes / (dt / args.steps) / 1e9,
                                    "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
                     # algorithmic flops per genotype = 4C + 6K (X.V, Q.P^T, dP, dQ, dV); the pass-2 kernel
                     # is bound by vector-ALU issue (BCE algebra per genotype), not by either roof -- DESIGN.md section 4
                     "mfma": {"alg_flops_per_genotype": 4 * 8 + 6 * K,
                              "achieved_tflops": world * b * M * args.steps / dt * (4 * 8 + 6 * K) / 1e12 / world,
                              "peak_tflops": MFMA_BF16_PEAK_TFLOPS,
                              "frac": b * M * args.steps / dt * (4 * 8 + 6 * K) / 1e12 / MFMA_BF16_PEAK_TFLOPS}},
        "loss_last_step": loss_last,
    }
    if args.share_gpu:
        out["config"]["share_gpu"] = "all ranks on cuda:0 over gloo: functional check, not a measurement"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eng, args, dev)
        print(json.dumps(out))
    if world > 1 or args.force_ddp or args.parallelism == "snp":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
