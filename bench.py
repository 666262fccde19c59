#!/usr/bin/env python3
"""Headline benchmark: genotypes/s (samples x SNPs per second of training) at K=8.

A "step" is one training step of the hot path on one batch: gather the batch's rows of the 2-bit packed
matrix, encoder X.V, RMSNorm+MLP+softmax, decoder Q.P^T + clamp + BCE forward/backward (loss value
included), MLP backward, dV = X^T.dZ, gradient all-reduce (N>1), Adam on every parameter, P clamp.
Workload (BASELINE.json configs[3]): synthetic 100k samples x 500k SNPs, K=8, resident 2-bit packed
in HBM (12.5 GB; sharded by samples over ranks).

Two multi-GPU modes, both sample-sharded: gradients reduce-scattered over RCCL, Adam on each rank's 1/N slice of the parameters,
all-gather of the result (csrc/nadm_step.hip, NADM_MODE_DP) -- every step is ONE C call:
  default            batch 800 PER GPU ("weak": the work per GPU is fixed, --batch_size 800*N in reference terms)
  --global-batch B   the reference's own semantics (neural_admixture.py:287: batch_size // num_gpus rows per GPU, i.e.
                     100 rows/GPU at N=8 for the default B=800; "strong": the work per step is fixed)

    python bench.py --gpus 1 --steps 100 --warmup 30
    python bench.py --gpus 8                       # spawns 8 ranks itself (re-exec under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --min-k 2 --max-k 10 --rows 2504 --snps 600000      # configs[2]: nine heads over one pass of X

With N > 1 ranks the same processes go on, behind the headline, to time every open multi-GPU alternative (`"alt"` in the JSON line:
the reference's global batch 800, SNP sharding at both batch semantics, message B in two buckets, a second communicator for message
A) and the step's own collectives at their real sizes (`"collectives"`), so that ONE run on an 8-GPU node decides DESIGN.md section 5's
open questions.  `--alt` runs those legs on one GPU too; `--share-gpu` proves the whole flow on a one-GPU box.
"""
import argparse
import hashlib
import json
import os
import socket
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak (same guide); the three GEMM-shaped products run as bf16 pieces
N_SIMD = 256 * 4          # 256 CUs x 4 SIMDs
PROFILE_ROUND = "r06"     # profiles/<round>_pmc_*.json hold the counter passes of the kernels of THIS build (tools/pmc_profile.py)

import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner through C stdio on stdout whenever a communicator is created; stdout carries the ONE JSON line:
    send fd 1 to stderr for the duration and flush the C buffers before switching back."""
    import ctypes
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved_fd, 1)
        os.close(saved_fd)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--ramp-ms", type=float, default=400.0,
                    help="untimed clock-ramp phase BEFORE the --warmup steps: the same steps for this many milliseconds.  After "
                         "start-up the part needs ~25 launches (~15 ms) to reach its sustained clock, and the first steps run ~15 %% slow")
    ap.add_argument("--rows", type=int, default=100_000, help="total samples N (sharded over ranks)")
    ap.add_argument("--snps", type=int, default=500_000)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--min-k", type=int, default=None, help="with --max-k: one decoder head per K in [min_k, max_k] (configs[2])")
    ap.add_argument("--max-k", type=int, default=None)
    ap.add_argument("--batch", type=int, default=800, help="rows per step PER GPU (weak scaling, the default mode)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="reference semantics: rows per step over ALL GPUs, batch_size // num_gpus per GPU (neural_admixture.py:287)")
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--no-loss", action="store_true", help="skip the loss value (gradients unchanged); default computes it every step like the reference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallelism", choices=("dp", "snp"), default="dp",
                    help="dp (default, the reference's scheme): samples sharded, gradient all-reduce; snp: SNPs sharded, every rank "
                         "processes the global batch of batch*N rows on its M/N SNPs, two small all-reduces per step")
    ap.add_argument("--buckets", type=int, default=None, help="sample-sharded step: SNP ranges message B = [small | V] travels in (default: NeuralAdmixture.dp_buckets)")
    ap.add_argument("--p3-whole", action="store_true", help="sample-sharded step with buckets: keep pass 3 one launch (only the next pass 1 is pipelined)")
    ap.add_argument("--comm-a", action="store_true", help="sample-sharded step: a second communicator for message A = [all P]")
    ap.add_argument("--force-ddp", action="store_true", help="single GPU: run the sample-sharded step (NADM_MODE_DP) on a 1-rank RCCL communicator")
    ap.add_argument("--emulate-world", type=int, default=None, metavar="W",
                    help="single GPU, with --force-ddp: the step of rank 0 of W ranks with no-op collectives (nadm_comm_emulated) -- Adam on 1/W of "
                         "the parameters, the rest treated as gathered.  The per-rank GPU and host cost of a W-rank step; NOT a scaling measurement")
    ap.add_argument("--time-kernels", choices=("dominant", "all"), default="dominant",
                    help="HIP events inside the timed region around the dominant kernel only (default: two event records per step) "
                         "or around every kernel of the step (kernel_ms table; the records cost ~4 %% of the step)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional check of the N>1 flow on a ONE-GPU box: all ranks use cuda:0 and gloo carries the tensors "
                         "(RCCL refuses two ranks per device); not a measurement")
    ap.add_argument("--alt", action="store_true", help="run the alternative multi-GPU legs + the collective micro-benchmark behind the headline (default at N > 1)")
    ap.add_argument("--no-alt", action="store_true", help="N > 1: only the headline")
    ap.add_argument("--alt-steps", type=int, default=50, help="timed steps of every alternative leg")
    ap.add_argument("--alt-warmup", type=int, default=10, help="untimed steps in front of every alternative leg")
    ap.add_argument("--alt-deadline", type=float, default=420.0,
                    help="seconds the alternative legs may take in all; past it rank 0 prints the headline line without them (alt_error says so) "
                         "and every rank leaves -- a stuck leg must not cost the run its headline")
    ap.add_argument("--coll-reps", type=int, default=20, help="timed repetitions of every collective of the micro-benchmark")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the 20 steps behind the timed region in which pass 2 clocks itself (profiling passes: "
                                                                  "only the kernels of the timed steps)")
    ap.add_argument("--no-epoch-loop", action="store_true", help="skip the production epoch loop behind the timed region (profiling passes: only the K-step kernels)")
    ap.add_argument("--cpu-rows", type=int, default=2400, help="rows of the same workload used for the bounded CPU baseline")
    ap.add_argument("--cpu-steps", type=int, default=12, help="timed steps of the CPU baseline (min / median / max reported)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one process per GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def source_hash():
    """Identity of the kernels a profile was taken from: sha256 over the KERNEL sources of libnadm.so (the host-only files --
    nadm_step.hip, nadm_host.h -- queue launches and do not change what a launch does)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "neural-admixture_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name in ("nadm_genotype_passes.hip", "nadm_small_kernels.hip", "nadm_common.h"):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def load_profile(name, workload_key):
    """profiles/<round>_<name>.json if it was taken from THESE kernel sources on THIS workload; else None, loudly."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{name}.json")
    try:
        with open(path) as f:
            pm = json.load(f)
    except (OSError, ValueError):
        print(f"[bench] no {path}: run tools/pmc_profile.py on the GPU box and commit the summary", file=sys.stderr)
        return None
    if pm.get("src_hash") != source_hash():
        print(f"[bench] {path} was taken from other kernel sources ({pm.get('src_hash')} != {source_hash()}): counters not reported; "
              "re-run tools/pmc_profile.py", file=sys.stderr)
        return None
    if pm.get("workload_key") != workload_key:
        return None                                                     # another shape than the profiled one: nothing to say
    return pm


def make_dataset(eng, rows_local, row0, K, dev, seed=1234):
    """Admixture-model synthetic genotypes generated on the device straight into packed bytes."""
    from neural_admixture_amd._lib import lib, check, ptr
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
    nb = max(1, n // b)
    nsteps = max(args.cpu_steps, 1)
    times = []
    for s in range(nsteps):                        # the sample's batches, cycled: every step does the full work of one batch
        t0 = time.perf_counter()
        cp.step(G, idx[(s % nb) * b:(s % nb + 1) * b], 2e-3)
        times.append(time.perf_counter() - t0)
    dt = float(np.sum(times))
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
            "value_best_step": b * L.M / float(np.min(times)), "value_median_step": b * L.M / float(np.median(times)),
            "step_s": {"min": float(np.min(times)), "median": float(np.median(times)), "max": float(np.max(times)), "n": nsteps},
            "sample": f"{nsteps} timed steps (after 1 warm-up) of {b} rows x {L.M} SNPs cycling over the first {n} rows of the same matrix, "
                      f"fused C/OpenMP port of the step (oracle/nadm_oracle_c.c), cpu={model}"}


def cpu_baseline_reference_shaped(eng, args, dev):
    """The step in the reference's own shape -- its sequence of torch CPU operators, true fp32 (oracle/torch_shape.py, pinned against
    the reference's fixtures) -- on every host core: what SURVEY 8d / BASELINE.md section 3 describe, beside the fused C port."""
    from neural_admixture_amd._lib import lib, check, ptr
    from oracle.torch_shape import TorchShapedModel
    L = eng.lay
    b = min(args.batch, eng.xp.shape[0])
    n = min(eng.xp.shape[0], 2 * b)
    Gd = torch.empty((n, L.M), dtype=torch.uint8, device=dev)
    check(lib.nadm_unpack2bit(ptr(eng.xp), ptr(Gd), n, L.M, eng.ld, None), "unpack")
    G = Gd.cpu()
    del Gd
    sm = eng.small.cpu().numpy()
    h, K = L.heads, L.ks[0]
    cores = max(1, (os.cpu_count() or 2) // 2)      # one thread per physical core (the elementwise chain is memory-bound: SMT siblings add nothing)
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    torch.set_float32_matmul_precision("highest")
    try:
        m = TorchShapedModel(eng.V().cpu().numpy(), [eng.P(0).cpu().numpy()], sm[h.g_off:h.g_off + L.C], sm[h.w1_off:h.w1_off + L.Hd * L.C].reshape(L.Hd, L.C),
                             sm[h.b1_off:h.b1_off + L.Hd], [sm[h.wk_off[0]:h.wk_off[0] + K * L.Hd].reshape(K, L.Hd)], [sm[h.bk_off[0]:h.bk_off[0] + K]], 2e-3)
        m.step(G[:b])                                  # warm-up (thread pool, allocator)
        times = []
        for s_ in range(2):
            t0 = time.perf_counter()
            m.step(G[(s_ % (n // b)) * b:(s_ % (n // b) + 1) * b])
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(prev)
    return {"value": b * L.M / float(np.median(times)), "unit": "genotypes/s", "cores": cores, "kind": "port",
            "shape": "the reference's operator sequence on torch CPU ops, fp32 (oracle/torch_shape.py)",
            "step_s": {"min": float(np.min(times)), "median": float(np.median(times)), "max": float(np.max(times)), "n": len(times)},
            "sample": f"2 timed steps (after 1 warm-up) of {b} rows x {L.M} SNPs, torch {torch.__version__} CPU, {cores} threads",
            "reference_itself_in_the_survey_container": {"value": 9.2e7, "cores": 8, "source": "BASELINE.md section 2: the unmodified reference, "
                                                         "8 Xeon cores (AMX bf16 matmuls), 8000 x 50000, K=8 -- another machine, quoted for scale"}}

def box_fingerprint(dev):
    """What THIS box sustains (r06): (1) csrc/nadm_calib.hip -- a fixed stream of packed-f32 VALU chains + bf16 matrix instructions at pass 2's
    occupancy on every CU, with s_memtime (shader cycles) and s_memrealtime (constant rate) read around it in every block: the
    effective shader clock under an issue-bound load, and the duration of the fixed work; (2) a 1 GiB device copy.  A slower
    headline on a box whose calibration launch is slower by the same factor is the box, not the code."""
    from neural_admixture_amd._lib import lib, ptr
    cap, iters = 1024, 8192
    out = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.nadm_calib_clock(iters, ptr(out), cap, ptr(sink), st)
    if nb <= 0:
        return None
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b_ in evs:
        a.record()
        lib.nadm_calib_clock(iters, ptr(out), cap, ptr(sink), st)
        b_.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b_) for a, b_ in evs)
    o = out[:2 * nb].view(nb, 2).cpu().numpy().astype(np.float64)
    khz = float(lib.nadm_wall_clock_khz())
    ok = o[:, 1] > 0
    ghz = float(np.median(o[ok, 0] / o[ok, 1])) * khz * 1e3 / 1e9 if khz > 0 and ok.any() else None
    n = 1 << 30
    src, dst = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    dst.copy_(src)
    cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b_ in cev:
        a.record()
        dst.copy_(src)
        b_.record()
    torch.cuda.synchronize()
    cms = sorted(a.elapsed_time(b_) for a, b_ in cev)
    del src, dst
    return {"effective_sclk_ghz": ghz, "calib_ms": ms[len(ms) // 2], "calib_ms_min": ms[0], "calib_blocks": int(nb), "calib_iters": iters,
            "shader_cycles_per_iter": float(np.median(o[:, 0])) / iters, "wall_clock_khz": khz,
            "copy_gbs": 2.0 * n / (cms[len(cms) // 2] * 1e-3) / 1e9,
            "note": "calib: 8 v_pk_fma_f32 chains + 1 v_mfma_f32_16x16x32_bf16 per round, 3 waves/SIMD on every CU (csrc/nadm_calib.hip); "
                    "effective_sclk_ghz = s_memtime cycles / s_memrealtime ticks x rate, median over blocks; copy: 1 GiB read + 1 GiB written"}


ALT_LEGS = ("dp_weak", "dp_global_batch", "snp_weak", "snp_global_batch", "dp_2buckets", "dp_comm_a")
ALT_COLLECTIVES = ("reduce_scatter_msg_a", "all_gather_msg_a", "reduce_scatter_msg_b", "all_gather_msg_b", "all_reduce_small")


def timed_leg(step, finish, warmup, steps, world, dev):
    """`warmup` untimed + `steps` timed calls of step(i) between barrier + synchronize on both sides; the max over ranks, seconds."""
    import torch.distributed as dist
    for i in range(warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def collectives_bench(comm, lay, dev, world, reps=20):
    """The step's own communicator on the step's own message sizes: reduce-scatter / all-gather of message A = [all P] and message
    B = [small | V] (slice x world floats each, in place, nadm_comm_t's contract) and the 200 KB all-reduce of the SNP-sharded step
    (b x sum K floats).  us per call (max over ranks of the mean over `reps` back-to-back calls) and bus GB/s = bytes x (W-1)/W / time
    (all-reduce: 2 x)."""
    import torch.distributed as dist
    c = comm.handle.contents
    st = torch.cuda.current_stream().cuda_stream
    n_a, n_b = int(lay.slice_a) * world, int(lay.slice_b) * world
    n_small = 50_000                                              # 200 KB: Z / dQ partials of a 6400-row global batch
    buf = torch.zeros(max(n_a, n_b, n_small), dtype=torch.float32, device=dev)
    tr = getattr(comm, "transport", None)
    if tr is not None:
        tr.buffers.append(buf)
    out = {}
    try:
        for name, fn, n, per in (("reduce_scatter_msg_a", c.reduce_scatter, n_a, n_a // world), ("all_gather_msg_a", c.all_gather, n_a, n_a // world),
                                 ("reduce_scatter_msg_b", c.reduce_scatter, n_b, n_b // world), ("all_gather_msg_b", c.all_gather, n_b, n_b // world),
                                 ("all_reduce_small", c.all_reduce, n_small, n_small)):
            def call(_i, fn=fn, per=per):
                if fn(c.ctx, buf.data_ptr(), per, st):
                    raise RuntimeError(f"collective {name} failed")
            dt = timed_leg(call, lambda: None, 3, reps, world, dev) / reps
            frac = (world - 1) / world if world > 1 else 1.0
            bus = 4.0 * n * frac * (2.0 if name == "all_reduce_small" else 1.0) / dt / 1e9
            out[name] = {"bytes": 4 * n, "us": dt * 1e6, "bus_gbs": bus}
            buf.zero_()
    finally:
        if tr is not None:
            tr.buffers[:] = [t_ for t_ in tr.buffers if t_ is not buf]
    return out


def validate_line(d, world):
    """Problems (strings) with a bench JSON line as the driver / a reader needs it; empty = fine.  With several ranks (or --alt) every
    alternative leg and every collective of the micro-benchmark must be there with a positive time."""
    bad = []
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "box"):
        if k not in d:
            bad.append(f"missing {k}")
    if d.get("n_gpus") != world:
        bad.append(f"n_gpus {d.get('n_gpus')} != {world}")
    if not (isinstance(d.get("value"), (int, float)) and d["value"] > 0 and d.get("ms_per_step", 0) > 0):
        bad.append("value / ms_per_step not positive")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        if k not in (d.get("roofline") or {}):
            bad.append(f"roofline.{k} missing")
    if world > 1 or "alt" in d:
        alt, coll = d.get("alt") or {}, d.get("collectives") or {}
        for leg in ALT_LEGS:
            v = alt.get(leg)
            if not v or not (v.get("ms_per_step", 0) > 0 and v.get("value", 0) > 0 and v.get("global_batch", 0) > 0):
                bad.append(f"alt.{leg} missing or empty")
        for c in ALT_COLLECTIVES:
            v = coll.get(c)
            if not v or not (v.get("us", 0) > 0 and v.get("bytes", 0) > 0 and "bus_gbs" in v):
                bad.append(f"collectives.{c} missing or empty")
    return bad


def bench_params(M, S):
    rng = np.random.default_rng(42)                                     # identical parameters on every rank
    V0 = (0.01 * rng.standard_normal((M, 8))).astype(np.float32)
    P0 = rng.uniform(5e-6, 1 - 5e-6, size=(S, M)).astype(np.float32)
    return V0, P0


def alt_legs(args, eng, comm, dev, world, rank, ks, headline):
    """Behind the headline, in the same processes: every alternative DESIGN.md section 5 lists for "the first 8-GPU node", `--alt-steps`
    timed steps each (barrier + synchronize on both sides, max over ranks), no CPU legs.  value = the genotypes ALL ranks processed
    per second of that leg.  Returns (legs, collectives)."""
    import neural_admixture_amd as na
    from neural_admixture_amd import comm as nacomm
    from neural_admixture_amd.model import init_encoder_weights
    from neural_admixture_amd.snp_parallel import SnpShardedEngine
    M, S, K, lr, with_loss = args.snps, sum(ks), max(ks), 2e-3, not args.no_loss
    steps, warm = max(1, args.alt_steps), max(0, args.alt_warmup)
    if comm is None:                                                    # one GPU, plain step in the headline: a 1-rank communicator for the legs
        comm = nacomm.torch_comm(rank, world) if args.share_gpu else nacomm.make_comm(dev, rank, world)
    mk_comm = (lambda: nacomm.torch_comm(rank, world)) if args.share_gpu else (lambda: nacomm.make_comm(dev, rank, world))
    legs = {"dp_weak": dict(headline, note="the headline above: samples sharded, --batch rows per GPU and step")}
    small = init_encoder_weights(42, 8, args.hidden, ks)
    xp = eng.xp
    rows_local = xp.shape[0]
    seq = torch.arange(rows_local, dtype=torch.int32, device=dev)       # stored shard order, contiguous batches (model.py launch_training)

    def run(name, e, rows_step, units_step, order, note):
        nb = max(1, int(order.numel()) // rows_step)

        def step(i):
            o = (i % nb) * rows_step
            e.train_step(order[o:o + rows_step], rows_step, lr, with_loss)
        dt = timed_leg(step, e.sync, warm, steps, world, dev)
        legs[name] = {"ms_per_step": dt / steps * 1e3, "value": units_step * steps / dt, "unit": "genotypes/s", "steps": steps,
                      "rows_per_rank_per_step": rows_step, "global_batch": units_step // M, "note": note}

    def dp_engine(**kw):
        e = na.Engine(M, 8, args.hidden, ks, dev, args.batch, mode="dp", comm=comm, **kw)
        e.set_packed(xp)
        e.load_params(*bench_params(M, S), small)
        return e

    dp = eng if eng.mode == "dp" else dp_engine()
    if eng.mode != "dp":
        run("dp_weak_1rank_comm", dp, args.batch, args.batch * world * M, seq, "the sample-sharded step on this rank count's communicator (the headline ran the plain step)")
    b_ref = max(1, args.batch // world)                                 # neural_admixture.py:287
    run("dp_global_batch", dp, b_ref, b_ref * world * M, seq,
        f"the reference's own semantics: --batch_size {args.batch} over all GPUs = {b_ref} rows per GPU and step (strong scaling)")
    if dp is not eng:
        del dp
    e2 = dp_engine(n_buckets=2)
    run("dp_2buckets", e2, args.batch, args.batch * world * M, seq, f"message B = [small | V] in {e2.lay.n_buckets} SNP-range buckets, pipelined against pass 3 / the next pass 1")
    del e2
    ca = mk_comm()
    e3 = dp_engine(comm_a=ca)
    run("dp_comm_a", e3, args.batch, args.batch * world * M, seq, "a second communicator for message A = [all P]: A and B share the links instead of queueing")
    coll = collectives_bench(comm, eng.lay if eng.mode == "dp" else e3.lay, dev, world, reps=max(1, args.coll_reps))
    del e3
    ca.close()
    torch.cuda.empty_cache()
    # SNP sharding: every rank holds ALL rows of its M / world SNPs (the same bytes per rank) and processes the global batch
    gb_w, gb_r = args.batch * world, args.batch
    es = SnpShardedEngine(M, 8, args.hidden, ks, dev, gb_w, comm=comm)
    es.set_packed(make_dataset(es, args.rows, 0, K, dev, seed=1234 + 7 * rank))
    es.load_params(*bench_params(M, S), small)
    gperm = torch.Generator(device="cpu").manual_seed(1000)             # the same global batches on every rank
    order = torch.randperm(args.rows, generator=gperm).to(torch.int32).to(dev)
    run("snp_weak", es, gb_w, gb_w * M, order, f"SNPs sharded, global batch {gb_w} (= --batch x GPUs): two ~{4 * gb_w * (8 + S) // 2 // 1000} KB all-reduces per step, no big message")
    run("snp_global_batch", es, gb_r, gb_r * M, order, f"SNPs sharded at the reference's global batch {gb_r}")
    del es
    torch.cuda.empty_cache()
    return legs, coll


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                               # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1
    if args.emulate_world is not None and (world > 1 or not args.force_ddp):
        raise SystemExit("--emulate-world needs --force-ddp on ONE GPU")
    rccl_ranks = 0
    if use_dist:
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
            t_ = torch.ones(1, device=f"cuda:{local_rank}")
            dist.all_reduce(t_)                                         # every rank contributes 1: the sum counts the ranks RCCL connected
            torch.cuda.synchronize()
            rccl_ranks = 0 if args.share_gpu else int(round(float(t_.item())))
            if int(round(float(t_.item()))) != world:
                raise RuntimeError(f"all-reduce over {world} ranks summed to {float(t_.item())}")
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    hang_guard = None
    if use_dist:
        # several ranks: a collective that never completes (a peer gone, a link down) would keep this process until the driver's own limit;
        # say so and leave after 15 minutes instead (cancelled once the line is printed)
        import threading

        def _give_up():
            sys.stderr.write(f"[bench] rank {rank}: no result within the deadline -- a collective or a peer is stuck; aborting\n")
            sys.stderr.flush()
            os._exit(3)
        hang_guard = threading.Timer(1800.0 if args.share_gpu else 900.0, _give_up)
        hang_guard.daemon = True
        hang_guard.start()
    import neural_admixture_amd as na
    from neural_admixture_amd.model import init_encoder_weights

    M = args.snps
    ks = [args.k] if args.min_k is None else list(range(args.min_k, args.max_k + 1))
    S, K = sum(ks), max(ks)
    if args.global_batch is not None:
        b = args.global_batch // world                                  # neural_admixture.py:287
        if b < 1:
            raise SystemExit("--global-batch smaller than the number of GPUs")
    else:
        b = args.batch
    snp = args.parallelism == "snp"
    ddp = not snp and (world > 1 or args.force_ddp)
    from neural_admixture_amd import comm as nacomm
    comm = None
    if args.emulate_world is not None:
        comm = nacomm.emulated_comm(args.emulate_world)
    elif snp or ddp:
        # a communicator of the library's own (ncclCommInitRank; torch.distributed only carries the 128-byte id): RCCL prints a
        # banner through C stdio on stdout -- keep stdout for the one JSON line
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = nacomm.torch_comm(rank, world) if args.share_gpu else nacomm.make_comm(dev, rank, world)
            if comm.kind == "rccl":
                rccl_ranks = comm.count_ranks(dev)                      # every rank contributes 1: the ranks the library's communicator connected
                if rccl_ranks != world:
                    raise RuntimeError(f"all-reduce over {world} ranks summed to {rccl_ranks}")
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    V0, P0 = bench_params(M, S)
    if snp:
        # every rank: ALL rows of its SNP slice (same bytes in HBM per rank as the sample-sharded layout), global batch b*world
        from neural_admixture_amd.snp_parallel import SnpShardedEngine
        rows_local, gb = args.rows, b * world
        eng = SnpShardedEngine(M, 8, args.hidden, ks, dev, gb, comm=comm)
        eng.set_packed(make_dataset(eng, rows_local, 0, K, dev, seed=1234 + 7 * rank))
        gperm = torch.Generator(device="cpu").manual_seed(1000)         # the same global batches on every rank
    else:
        rows_local, gb = args.rows // world, b
        comm_a = None
        if ddp and args.comm_a:
            comm_a = nacomm.emulated_comm(args.emulate_world) if args.emulate_world is not None else nacomm.make_comm(dev, rank, world)
        n_buckets = args.buckets if args.buckets is not None else na.NeuralAdmixture.dp_buckets
        eng = na.Engine(M, 8, args.hidden, ks, dev, b, mode="dp" if ddp else "single", comm=comm if ddp else None,
                        n_buckets=n_buckets if ddp else 1, comm_a=comm_a, p3_whole=args.p3_whole)
        eng.set_packed(make_dataset(eng, rows_local, rank * rows_local, K, dev))
        gperm = torch.Generator(device="cpu").manual_seed(1000 + rank)
    eng.load_params(V0, P0, init_encoder_weights(42, 8, args.hidden, ks))
    del V0, P0
    if world > 1 and not snp:
        # the production trainer stores a rank's rows in shard order and steps through them contiguously (model.py launch_training:
        # `order = seq`; DistributedSampler.set_epoch is never called, src/loaders.py:27): the bench's pass 1 gathers what the trainer's does
        perm = torch.arange(rows_local, dtype=torch.int32, device=dev)
    else:
        perm = torch.randperm(rows_local, generator=gperm).to(torch.int32).to(dev)
    nb = max(1, rows_local // gb)
    with_loss = not args.no_loss
    lr = 2e-3

    def step(s):                                                        # ONE C call: launches and collectives of the step
        o = (s % nb) * gb
        eng.train_step(perm[o:o + gb], gb, lr, with_loss)

    # untimed clock-ramp phase: the same steps until --ramp-ms of wall-clock have passed (the sustained clock is reached
    # after ~25 launches; `--warmup 5 --steps 20` straight after start-up would time the ramp, not the step)
    n_ramp = 0
    torch.cuda.synchronize()
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
        for _ in range(10):
            step(n_ramp)
            n_ramp += 1
        torch.cuda.synchronize()
        if world > 1:                                                   # every rank leaves the phase after the same number of steps
            t = torch.tensor([1.0 if (time.perf_counter() - t_r) * 1e3 < args.ramp_ms else 0.0], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if float(t.item()) == 0.0:
                break
    for s in range(args.warmup):
        step(n_ramp + s)
    from neural_admixture_amd._lib import T_NAMES
    eng.time_kernels(T_NAMES if args.time_kernels == "all" else ("decode_bce",))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(n_ramp + args.warmup + s)
    eng.sync()                                             # what the last step left to "the next one" belongs to the timed work
    t_queued = time.perf_counter() - t0                    # the host has queued everything (the GPU is still running)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_rank = dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kms = eng.kernel_ms()
    loss_sum, loss_last = eng.read_loss()
    assert np.isfinite(loss_last) or not with_loss
    if args.time_kernels == "dominant":
        # the other kernels of the step, for the kernel_ms table only: a short pass AFTER the timed region with events around
        # every kernel (those records cost ~4 % of the step, which is why the timed region carries only the dominant kernel's)
        eng.time_kernels(T_NAMES)
        for s in range(min(args.steps, 20)):
            step(n_ramp + args.warmup + args.steps + s)
        eng.sync()
        torch.cuda.synchronize()
        for name, v in eng.kernel_ms().items():
            if name != "decode_bce":
                kms[name] = v
    eng.time_kernels(None)
    # the dominant kernel clocks itself (r06), straight behind the timed region and NOT inside it: while a probe pointer is set pass 2 runs its
    # measurement build (the block in the middle of the grid reads s_memtime / s_memrealtime around its work) -- the timed steps run the kernel
    # without it
    from neural_admixture_amd._lib import lib as _nlib, ptr as _ptr
    clk_probe = torch.zeros(2, dtype=torch.int64, device=dev)
    if not args.no_clock_probe:
        _nlib.nadm_clock_probe(_ptr(clk_probe))
        for s in range(20):
            step(n_ramp + args.warmup + args.steps + 20 + s)
        eng.sync()
        torch.cuda.synchronize()
        _nlib.nadm_clock_probe(None)
    probe_cycles, probe_ticks = (int(v) for v in clk_probe.cpu())
    # the host's cost of queueing a step, measured where the GPU cannot hide it: steps queued back to back onto an idle device
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    for s in range(20):
        step(s)
    host_queue_ms = (time.perf_counter() - t_h) / 20 * 1e3
    torch.cuda.synchronize()
    box = box_fingerprint(dev)                              # the chip is warm from the timed region: what this box sustains (r06)
    if box is not None:
        # ... and what it sustained under the dominant kernel itself, measured INSIDE that kernel during the timed region (pass 2's S = 1
        # form; None for the other forms): denser than the calibration stream, pass 2 clocks lower, and by how much differs box to box
        box["dominant_kernel_sclk_ghz"] = probe_cycles / probe_ticks * box["wall_clock_khz"] * 1e3 / 1e9 if probe_ticks > 0 else None
        box["dominant_kernel_probe_block_us"] = probe_ticks / (box["wall_clock_khz"] * 1e3) * 1e6 if probe_ticks > 0 else None
    if box is not None and world > 1:
        clocks = [None] * world
        dist.all_gather_object(clocks, (box["effective_sclk_ghz"], box["calib_ms"]))
        box["by_rank"] = [{"effective_sclk_ghz": c_[0], "calib_ms": c_[1]} for c_ in clocks]
    # ---- the literal "epoch-seconds" of the metric: the production trainer's epoch loop (model.NeuralAdmixture.launch_training: the
    # sampler's order per epoch drawn and copied underneath the steps, every batch of the epoch incl. a ragged last one, the loss value
    # only on the epochs that log it -- every 5th, neural_admixture.py:416) over whole epochs of the resident matrix.  Reported beside
    # the K-step figure, which stays the headline (`value`): that one computes the loss value on EVERY step like the reference does.
    full_run = None
    if world == 1 and not snp and args.emulate_world is None and not args.force_ddp and not args.no_epoch_loop:
        from neural_admixture_amd.model import _EpochOrders
        n_ep = 5                                                            # epochs 0..4: one logged epoch in five, like a default run
        host_threads = torch.get_num_threads()
        torch.set_num_threads(1)                                            # like the trainer: the loop is launches only (model.launch_training)
        orders = _EpochOrders(torch.Generator().manual_seed(42), rows_local, dev)
        torch.cuda.synchronize()
        t_e = time.perf_counter()
        for epoch in range(n_ep):
            logged = epoch % 5 == 0
            order = orders.take(epoch, prefetch=epoch + 1 < n_ep)
            for s0 in range(0, rows_local, b):
                bb = min(b, rows_local - s0)
                eng.train_step(order[s0:s0 + bb], bb, lr, logged)
            orders.epoch_queued()
            if logged:
                eng.read_loss(reset=True)
        eng.sync()
        torch.cuda.synchronize()
        ep_s = (time.perf_counter() - t_e) / n_ep
        torch.set_num_threads(host_threads)
        full_run = {"epoch_ms_full_run": ep_s * 1e3, "genotypes_per_s": rows_local * M / ep_s, "epochs_timed": n_ep,
                    "steps_per_epoch": (rows_local + b - 1) // b,
                    "note": "production epoch loop, loss value on logged epochs only (1 in 5); `value` above computes it on every step"}
    # ---- roofline of the dominant kernel = pass 2 (decode_bce: all heads of the step) ----
    # Algorithmic bytes in the accounting of SURVEY.md 8d (whole step = 0.75 B/genotype of packed X + 36 B per parameter of
    # parameter/optimizer traffic): one 2-bit pass over the batch + the pass's share of the per-parameter traffic.
    #   alg_8d : SURVEY 8d's full 36 B per P parameter when Adam on P runs in the kernel's epilogue (single-GPU and SNP-sharded
    #            steps) -- that rule prices the dP write + re-read and a second P read, which the fused epilogue does NOT perform;
    #            8 B (read P, write dP) when Adam is a launch of its own behind the all-reduce (data-parallel step)
    #   alg_min: the bytes the launch must really move: fused = read P, m, v + write P, m, v = 24 B per parameter; unfused = 8 B
    dom = "decode_bce"
    fused = snp or not ddp                                                # Adam + restrict_P in the pass's epilogue
    rows_b, m_loc = (gb, eng.M) if snp else (b, M)                       # snp: global batch x own SNP slice
    # X is priced ONCE whatever the number of heads (0.25 B per genotype and pass); the launches of a multi-head model
    # walk X once per head, which is real traffic but not algorithmic -- reported separately as x_walks / bytes_incl_x_rewalks
    x_walks = len(ks)
    alg_8d = rows_b * m_loc / 4 + (36 if fused else 8) * m_loc * S
    alg_min = rows_b * m_loc / 4 + (24 if fused else 8) * m_loc * S
    kp_sum = sum(int(k_) for k_ in eng.lay.kp)
    composition = {                                                      # known bytes of the pass-2 launch(es) per access pattern (tools/pmc_profile.py)
        "x_pieces_read": x_walks * rows_b * m_loc / 4.0,
        "param_stream_read": (12 if fused else 4) * m_loc * kp_sum * 1.0 + (0 if fused else 0),
        "batch_copy_write": rows_b * m_loc / 4.0 if eng._xg is not None else 0.0,
        "dq_slab_write": sum(int(c_) * rows_b * int(k_) * 4.0 for c_, k_ in zip(eng.lay.dec_chunks, eng.lay.kp)),
        "param_stream_write": (12 if fused else 4) * m_loc * kp_sum * 1.0}
    t_dom = kms[dom] * 1e-3
    wk = f"{rows_b}x{m_loc}x{'-'.join(map(str, ks))}:{'fused' if fused else 'unfused'}:{'loss' if with_loss else 'noloss'}"
    hbm = load_profile("pmc_hbm", wk) if rank == 0 else None
    sq = load_profile("pmc_sq", wk) if rank == 0 else None
    genotypes_launch = rows_b * m_loc * len(ks)                          # per-genotype BCE algebra runs once per head
    issue = None
    if sq is not None:
        # SQ_INSTS_VALU / SQ_INSTS_MFMA are summed over the shader engines by rocprofv3's per-dispatch record
        valu, mfma = sq["valu_insts_per_launch"], sq["mfma_insts_per_launch"]
        # the shader clock is MEASURED in this run (s_memtime against the constant-rate clock inside pass 2 itself), not assumed; no "peak" instruction rate is claimed (an instruction costs 2.9-8.3 cycles by its class): the figure is the
        # VALU wave-instructions each SIMD issued per shader cycle of THIS run's launch -- times the mix's mean cost it is valu_busy_frac
        clk = (box.get("dominant_kernel_sclk_ghz") or box.get("effective_sclk_ghz")) if box is not None else None
        simd_cycles = sq.get("busy_cycles_sum_over_se", 0.0) * 32.0        # SQ_BUSY_CYCLES is per shader engine (32 SIMDs each)
        issue = {"valu_insts_per_genotype": valu * 64.0 / genotypes_launch, "mfma_insts_per_64_genotypes": mfma * 64.0 / genotypes_launch,
                 "valu_wave_insts_per_launch": valu, "sclk_ghz": clk, "sclk_source": "measured in this run: inside the dominant kernel (box.dominant_kernel_sclk_ghz), else the calibration stream's",
                 "valu_insts_per_simd_cycle": valu / t_dom / (N_SIMD * clk * 1e9) if clk else None,
                 "shader_cycles_per_launch_this_run": t_dom * clk * 1e9 if clk else None,
                 "shader_cycles_per_launch_profiled": simd_cycles / N_SIMD if simd_cycles else None,
                 # the clock the DOMINANT kernel itself ran at in this run, if it spent the profiled launch's cycles: denser than the
                 # calibration stream, it clocks a few per cent lower on the same box (MI355X_MICROARCH.md, DVFS give-back)
                 "implied_sclk_ghz_dominant_kernel": simd_cycles / N_SIMD / t_dom / 1e9 if simd_cycles else None,
                 # counters of the same launch: cycles a SIMD spent issuing VALU work / matrix work over all SIMD-cycles of the launch.
                 # The two do not overlap on a SIMD (tools/ubench_issue.hip), so their sum is the issue utilisation; the rest is the
                 # partly filled last round of blocks, barriers and dependency stalls
                 "simd_valu_busy_frac": sq["active_inst_valu_quad_cycles"] * 4.0 / simd_cycles if simd_cycles else None,
                 "simd_mfma_busy_frac": sq["valu_mfma_busy_cycles"] / simd_cycles if simd_cycles else None,
                 "note": "instruction counts and busy fractions: replayed counter passes (issue_source); clock and launch duration: this run"}
    achieved = alg_8d / t_dom / 1e9
    step_bytes = 0.75 * b * M + 36.0 * (8 + S) * M                        # SURVEY.md 8d whole-step figure: 0.75 + 36 (C+S)/b per genotype
    per_step_units = (gb if snp else b * world) * M
    cfg_name = {(100_000, 500_000, (8,)): "configs[3]", (500_000, 1_000_000, (16,)): "configs[4]", (2504, 600_000, (7,)): "configs[1]",
                (2504, 600_000, tuple(range(2, 11))): "configs[2]"}.get((args.rows, M, tuple(ks)), "configs[3] shape, resized")
    mode = "snp" if snp else ("dp-global-batch" if args.global_batch is not None else "dp")
    out = {
        "metric": "genotypes/sec (samples x SNPs / epoch-sec) at K=%s" % (K if len(ks) == 1 else f"{ks[0]}..{ks[-1]}"),
        "value": per_step_units * args.steps / dt, "unit": "genotypes/s",
        "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.global_batch is not None else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: synthetic {args.rows} samples x {M} SNPs, K={'/'.join(map(str, ks))}, 2-bit packed resident in HBM, "
                               f"{'SNP' if snp else 'sample'}-sharded over {world} GPU(s), batch {b}/GPU, hidden {args.hidden}, n_components 8, "
                               f"loss value {'every step' if with_loss else 'skipped'}",
                   "global_batch": gb if snp else b * world, "parallelism": f"{args.parallelism}{world}", "mode": mode,
                   "clock_ramp_steps_untimed": n_ramp, "transport": comm.kind if comm is not None else None,
                   "emulated_world": args.emulate_world,
                   "message_b_buckets": eng.lay.n_buckets if ddp else None, "pass3_in_ranges": (not args.p3_whole) if ddp else None,
                   "second_communicator": (eng.comm_a is not None) if ddp else None},
        # one C call per step (nadm_step): host time to queue a step onto an idle device; the timed region's own loop took
        # t_queued to queue (GPU-bound: the launch queue stays ahead)
        "host_queue_ms_per_step": host_queue_ms, "host_loop_ms_per_step_in_timed_region": t_queued / args.steps * 1e3,
        # bound: what binds the kernel (VALU + matrix instruction issue: issue_frac of all SIMD-cycles); roof_8d: the roof SURVEY 8d prices
        # it against (achieved / peak / frac are in that accounting)
        "roofline": {"bound": "valu_issue", "roof_8d": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_8d": achieved / HBM_PEAK_GBS, "frac_min": alg_min / t_dom / 1e9 / HBM_PEAK_GBS,
                     # what the counters say the launch is bound by: the fraction of all SIMD-cycles of the launch spent issuing VALU and
                     # matrix instructions (they do not overlap on a SIMD)
                     "issue_frac": (issue["simd_valu_busy_frac"] + issue["simd_mfma_busy_frac"]) if issue and issue["simd_valu_busy_frac"] is not None else None,
                     "valu_busy_frac": issue["simd_valu_busy_frac"] if issue else None, "mfma_busy_frac": issue["simd_mfma_busy_frac"] if issue else None,
                     "x_walks": x_walks, "bytes_incl_x_rewalks": alg_8d + (x_walks - 1) * rows_b * m_loc / 4,
                     "traffic": hbm["traffic_bytes_per_launch"] if hbm else None,
                     "traffic_source": (f"profiles/{PROFILE_ROUND}_pmc_hbm.json (replayed, not measured in this run: separate --pmc passes of FETCH_SIZE and WRITE_SIZE; {hbm['correction']}; src_hash {hbm['src_hash']})"
                                        if hbm else None),
                     "issue_source": (f"profiles/{PROFILE_ROUND}_pmc_sq.json (replayed, not measured in this run: separate --pmc passes of the SQ counters on "
                                      f"the same kernel sources and workload; src_hash {sq['src_hash']})" if sq else None),
                     "traffic_composition": composition,
                     "kernel_ms": kms, "alg_bytes_per_launch": alg_8d, "alg_bytes_min_per_launch": alg_min,
                     "alg_bytes_note": "frac/frac_8d: 2-bit pass over the batch + %d B per P parameter (SURVEY 8d rule; %s); frac_min: the same pass "
                                       "+ %d B per P parameter, the bytes the launch has to move" % (
                                           36 if fused else 8, "Adam on P fused into the launch, no dP round trip" if fused else "read P, write dP; Adam is a separate launch",
                                           24 if fused else 8),
                     "issue": issue, "workload_key": wk,
                     "whole_step": {"alg_bytes": step_bytes, "achieved": step_bytes / (dt / args.steps) / 1e9,
                                    "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
                     # algorithmic flops per genotype = 4C + 6S (X.V, Q.P^T, dP, dQ, dV)
                     "mfma": {"alg_flops_per_genotype": 4 * 8 + 6 * S,
                              "achieved_tflops": b * M * args.steps / dt * (4 * 8 + 6 * S) / 1e12,
                              "peak_tflops": MFMA_BF16_PEAK_TFLOPS,
                              "frac": b * M * args.steps / dt * (4 * 8 + 6 * S) / 1e12 / MFMA_BF16_PEAK_TFLOPS}},
        "loss_last_step": loss_last,
        "epoch_loop": full_run,
        "box": box,
    }
    if world > 1:
        out["ms_per_step_this_rank"] = dt_rank / args.steps * 1e3          # rank 0's own clock; ms_per_step is the max over ranks
    if args.share_gpu:
        out["config"]["share_gpu"] = "all ranks on cuda:0 over gloo: functional check, not a measurement"
    if ((world > 1 and not args.no_alt) or args.alt) and not snp and args.emulate_world is None:
        headline = {"ms_per_step": out["ms_per_step"], "value": out["value"], "unit": "genotypes/s", "steps": args.steps,
                    "rows_per_rank_per_step": b, "global_batch": b * world}
        # The legs come behind a measured headline and must never cost it: a deadline of their own (a leg that hangs -- a collective that
        # never completes, a second communicator that does not come up) and a catch-all (a leg that raises) both end in rank 0 printing the
        # line it already has, with alt_error saying what happened, and every rank leaving.
        import threading

        def _headline_only(why):
            if rank == 0:
                line = dict(out, alt=None, collectives=None, alt_error=why)
                os.write(saved_stdout_fd, (json.dumps(line) + "\n").encode())
            sys.stderr.write(f"[bench] rank {rank}: {why}\n")
            sys.stderr.flush()
            os._exit(0)
        saved_stdout_fd = os.dup(1)                                     # (stdout itself points at stderr while the legs run)
        alt_guard = threading.Timer(max(1.0, args.alt_deadline), _headline_only,
                                    args=(f"the alternative legs did not complete within {args.alt_deadline:.0f} s: headline only",))
        alt_guard.daemon = True
        alt_guard.start()
        try:
            with stdout_to_stderr():                                    # (the legs create communicators)
                out["alt"], out["collectives"] = alt_legs(args, eng, comm if ddp else None, dev, world, rank, ks, headline)
        except BaseException as e:                                      # noqa: BLE001 -- whatever it was, the headline goes out
            if rank == 0 or world == 1:
                _headline_only(f"the alternative legs failed on rank {rank} ({type(e).__name__}: {e}): headline only")
            sys.stderr.write(f"[bench] rank {rank}: alternative legs failed ({type(e).__name__}: {e}); waiting for the deadline\n")
            sys.stderr.flush()
            time.sleep(max(1.0, args.alt_deadline) + 5.0)               # the other ranks are inside a leg: all leave at the deadline
            os._exit(0)
        alt_guard.cancel()
        os.close(saved_stdout_fd)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and len(ks) == 1 and args.emulate_world is None:
            out["cpu_baseline"] = cpu_baseline(eng, args, dev)
            out["cpu_baseline_reference_shaped"] = cpu_baseline_reference_shaped(eng, args, dev)
        problems = validate_line(out, world)
        if problems:
            print(f"[bench] incomplete line: {problems}", file=sys.stderr)
        print(json.dumps(out))
        sys.stdout.flush()
    if hang_guard is not None:
        hang_guard.cancel()
    comm_a2 = eng.comm_a
    if comm is not None:                                                # every rank gets here: the communicator goes down together
        del eng
        comm.close()
        if comm_a2 is not None:
            comm_a2.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
