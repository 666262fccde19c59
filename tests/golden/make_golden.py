#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE on CPU.

Runs only in the build container (the reference is not present on the GPU box); the .npz files it
writes are committed and are the only thing that travels.  Nothing here is reference source: the
script imports the reference package from a scratch build and records inputs/outputs.

Recipe for the scratch build:
    mkdir -p /tmp/refbuild && cd /tmp/refbuild
    cp -r /root/reference/neural_admixture /root/reference/setup.py .
    printf '__version__ = "0.0.0+scratch"\n__version_tuple__ = (0, 0, 0)\n' > neural_admixture/_version.py
    python3 setup.py build_ext --inplace          # Cython utils/rsvd (.so), ~1 min
    NADM_REF=/tmp/refbuild python3 /root/repo/tests/golden/make_golden.py

Precision: the reference calls torch.set_float32_matmul_precision('medium') inside
launch_training (model/neural_admixture.py:349), which selects bf16 matmuls on this AMX CPU.  Every
trajectory is captured twice: "hi" (that call neutralised -> true fp32, the oracle target) and
"med" (as-is, the reference's own noise floor; used to state tolerances, not as a target).
"""
import os
import sys
import math
import contextlib

import numpy as np
import torch

REF = os.environ.get("NADM_REF", "/tmp/refbuild")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.dirname(os.path.abspath(__file__))

from neural_admixture.model.neural_admixture import Q_P, NeuralAdmixture  # noqa: E402
from neural_admixture.src.utils_c import utils as ref_cy                  # noqa: E402
from neural_admixture.src import utils as ref_utils                       # noqa: E402
from neural_admixture.src.svd import RSVD                                 # noqa: E402

_real_set_prec = torch.set_float32_matmul_precision


@contextlib.contextmanager
def precision(mode):
    """mode 'hi': neutralise the reference's 'medium' request; 'med': leave it."""
    if mode == "hi":
        torch.set_float32_matmul_precision = lambda *_a, **_k: None
        _real_set_prec("highest")
    try:
        yield
    finally:
        torch.set_float32_matmul_precision = _real_set_prec
        _real_set_prec("highest")


def synth(N, M, K, seed=1234, missing=0.01):
    rng = np.random.default_rng(seed)
    Fq = np.clip(0.5 * rng.beta(0.5, 0.5, size=(K, M)), 0.005, 0.5)
    Qt = rng.dirichlet(0.2 * np.ones(K), size=N)
    G = rng.binomial(2, Qt @ Fq).astype(np.uint8)
    G[rng.random((N, M)) < missing] = 3
    return G


def pack_rule(G):
    """pack2bit.cu:26-31 restated with explicit loops (independent of oracle.pack2bit)."""
    N, M = G.shape
    Mp = (M + 3) // 4
    out = np.zeros((N, Mp), dtype=np.uint8)
    for r in range(N):
        for c in range(Mp):
            b = 0
            for i in range(4):
                j = 4 * c + i
                if j < M:
                    b |= (int(G[r, j]) & 3) << (2 * i)
            out[r, c] = b
    return out


def state_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def flat_state(prefix, sd):
    return {f"{prefix}{k.replace('.', '_')}": v for k, v in sd.items()}


def grads_np(model):
    return {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}


# ------------------------------------------------------------------------------------------
def case_pack():
    rng = np.random.default_rng(7)
    G = rng.integers(0, 4, size=(5, 11), dtype=np.uint8)
    G[0, :4] = [0, 1, 2, 3]
    G2 = rng.integers(0, 256, size=(3, 9), dtype=np.uint8)   # high bits must be masked (&3)
    np.savez_compressed(os.path.join(OUT, "pack_layout.npz"), G=G, packed=pack_rule(G),
                        G_hibits=G2, packed_hibits=pack_rule(G2))


def case_bce_elementwise():
    """clamp_(r,0,1) -> BCELoss(sum) forward/backward on hand-picked pre-clamp values
    (neural_admixture.py:97,288): pins the -100 log clamp, the 1e-12 denominator, and the
    inclusive gradient mask on the PRE-clamp value without any GEMM rounding in the way."""
    one = np.float32(1.0)
    specials = np.asarray([0.0, 1.0, -0.0, -1e-3, 1.0 + 1e-3, np.nextafter(one, np.float32(2)), np.nextafter(one, np.float32(0)),
                           1e-30, 1e-13, 1e-12, 1e-7, 0.5, 0.25, 0.999999, 2.0, -5.0, 1e-38], dtype=np.float32)
    rng = np.random.default_rng(12)
    r = np.concatenate([specials, rng.uniform(-0.1, 1.1, 200).astype(np.float32)])
    rr = np.repeat(r, 3)
    xx = np.tile(np.asarray([0.0, 0.5, 1.0], dtype=np.float32), len(r))
    rt = torch.tensor(rr, requires_grad=True)
    out = torch.clamp_(rt * 1.0, 0, 1)
    per = torch.nn.functional.binary_cross_entropy(out, torch.tensor(xx), reduction="none")
    per.sum().backward()
    np.savez_compressed(os.path.join(OUT, "bce_elementwise.npz"), r_raw=rr, x=xx, loss=per.detach().numpy(), grad=rt.grad.numpy())


def one_step_case(name, N, M, ks, Hd, C, seed, edge=False, steps=3, lr=2e-3, sup=False):
    G = synth(N, M, max(ks), seed=seed + 1, missing=0.03)
    rng = np.random.default_rng(seed + 2)
    V0 = (rng.standard_normal((M, C)) / math.sqrt(M)).astype(np.float32)
    S = sum(ks)
    P0 = rng.uniform(0.02, 0.98, size=(S, M)).astype(np.float32)
    if edge:
        # exact zeros: r is exactly 0 whatever the summation order (robust); near-one values keep
        # r < 1.  All-ones rows are NOT used: there r = sum_k q_k = 1 +- 1ulp depends on the GEMM's
        # summation order, so the reference's own result is not reproducible to rounding; the
        # saturated elementwise semantics are pinned separately by case_bce_elementwise().
        P0[:, :40] = 0.0
        P0[:, 40:60] = rng.uniform(0.999, 1.0, size=(S, 20)).astype(np.float32)
    out = dict(G=G, V0=V0, P0=P0, ks=np.asarray(ks), Hd=Hd, seed=seed, lr=lr)
    if sup:   # supervised term of neural_admixture.py:470-473 on head 0
        labels = rng.integers(0, ks[0], size=N)
        out["labels"] = labels
        ce = torch.nn.CrossEntropyLoss(reduction="sum")
        yt = torch.tensor(labels, dtype=torch.int64)
    with precision("hi"):
        torch.manual_seed(seed)
        model = Q_P(Hd, C, V=torch.tensor(V0), P=torch.tensor(P0), ks_list=list(ks))
        out.update(flat_state("init_", state_np(model)))
        opt = model.create_custom_adam(device=torch.device("cpu"), lr=lr)
        loss_fn = torch.nn.BCELoss(reduction="sum")
        Gt = torch.tensor(G)
        for s in range(steps):
            opt.zero_grad(set_to_none=True)
            (recs, probs), X = model(Gt)
            loss = sum(loss_fn(rec, X) for rec in recs)
            if sup:
                loss = loss + 100 * ce(probs[0], yt)
            loss.backward()
            out[f"loss{s}"] = np.float64(loss.item())
            if s == 0:
                with torch.no_grad():
                    Xf = Gt.float() / 2
                    Xf = torch.where(Xf == 1.5, 0.0, Xf)
                    out["Z0"] = (Xf @ model.V).numpy().copy()
                for h, q in enumerate(probs):
                    out[f"Q0_{h}"] = q.detach().numpy().copy()
                out.update({f"grad0_{k.replace('.', '_')}": v for k, v in grads_np(model).items()})
            opt.step()
            model.restrict_P()
            out.update(flat_state(f"after{s}_", state_np(model)))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, "losses", [out[f"loss{s}"] for s in range(steps)])


def run_reference_training(G, V_MC, P_SM, K, min_k, max_k, Hd, epochs, batch, lr, seed, mode):
    """launch_training on CPU (num_gpus=0), recording each step's loss."""
    N, M = G.shape
    with precision(mode):
        torch.manual_seed(seed)
        na = NeuralAdmixture(K, epochs, batch, lr, torch.device("cpu"), seed, 0, True, None, min_k, max_k)
        step_losses = []
        orig = na._run_step

        def wrapped(x):
            loss = orig(x)
            step_losses.append(float(loss.item()))
            return loss
        na._run_step = wrapped
        Qs, Ps, model = na.launch_training(torch.tensor(P_SM.copy()), torch.tensor(G), Hd, V_MC.shape[1],
                                           torch.tensor(V_MC.copy()), M, N, None)
        sd = state_np(model)
    return Qs, Ps, sd, np.asarray(step_losses)


def case_demo():
    """c1: bundled demo BED, K=3, batch 800 (> N: one step per epoch), 5 and 25 epochs."""
    seed, K, Hd, C, lr = 42, 3, 1024, 8, 2e-3
    ref_utils.set_seed(seed)
    bed = "/root/reference/demo/data/demo_data.bed"
    data, _, N, M = ref_utils.read_data(bed)
    Vt = RSVD(data, N, M, C, seed)                                   # [C,M]
    # GMM init exactly as train.py:49-63 does it
    from sklearn.mixture import GaussianMixture
    X_pca = (data.astype(np.float32) / 2 @ Vt.T).astype("float64")
    gmm = GaussianMixture(n_components=K, n_init=5, init_params="k-means++", tol=1e-4,
                          covariance_type="full", max_iter=100, random_state=seed).fit(X_pca)
    P_SM = np.clip(gmm.means_ @ Vt, 5e-6, 1 - 5e-6).astype(np.float32)
    V_MC = np.ascontiguousarray(Vt.T.astype(np.float32))
    raw_bed = np.fromfile(bed, dtype=np.uint8)
    out = dict(bed_bytes=raw_bed, N=N, M=M, G_packed=pack_rule(data), Vt=Vt.astype(np.float32), P_init=P_SM,
               gmm_means=gmm.means_, seed=seed, K=K, Hd=Hd, lr=lr)
    for mode in ("hi", "med"):
        for ep in (5, 25):
            Qs, Ps, sd, sl = run_reference_training(data, V_MC, P_SM, K, None, None, Hd, ep, 800, lr, seed, mode)
            out[f"{mode}_e{ep}_Q"] = Qs[0]
            out[f"{mode}_e{ep}_P"] = Ps[0]
            out[f"{mode}_e{ep}_losses"] = sl
            if ep == 5:
                out[f"{mode}_e{ep}_V"] = sd["V"]
                out.update(flat_state(f"{mode}_e5_sd_", {k: v for k, v in sd.items()
                                                          if not k.startswith("decoders") and k != "V"}))
                P64 = np.ascontiguousarray(Ps[0].astype(np.float64))
                Q64 = np.ascontiguousarray(Qs[0].astype(np.float64))
                out[f"{mode}_e5_loglik"] = np.float64(ref_cy.loglikelihood(data, P64, Q64, K))
                out[f"{mode}_e5_fst"] = np.asarray([[NeuralAdmixture._hudsons_fst(torch.tensor(Ps[0][:, a]), torch.tensor(Ps[0][:, b_]))
                                                     if b_ < a else 0.0 for b_ in range(K)] for a in range(K)])
            print("demo", mode, ep, "loss0", sl[0], "lossN", sl[-1])
    np.savez_compressed(os.path.join(OUT, "demo_k3.npz"), **out)


def case_multibatch():
    """c4-shaped miniature: N=1000, M=2048, K=8, b=400 (steps 400,400,200), 3 epochs."""
    N, M, K, Hd, C, b, ep, seed, lr = 1000, 2048, 8, 1024, 8, 400, 3, 42, 2e-3
    G = synth(N, M, K, seed=1234)
    rng = np.random.default_rng(5)
    V_MC = (rng.standard_normal((M, C)) / math.sqrt(M)).astype(np.float32)
    P_SM = rng.uniform(0.05, 0.95, size=(K, M)).astype(np.float32)
    out = dict(G_packed=pack_rule(G), N=N, M=M, K=K, Hd=Hd, b=b, epochs=ep, seed=seed, lr=lr, V0=V_MC, P0=P_SM)
    g = torch.Generator().manual_seed(seed)
    sampler = torch.utils.data.RandomSampler(range(N), generator=g)
    out["orders"] = np.asarray([list(iter(sampler)) for _ in range(ep)], dtype=np.int64)
    for mode in ("hi", "med"):
        Qs, Ps, sd, sl = run_reference_training(G, V_MC, P_SM, K, None, None, Hd, ep, b, lr, seed, mode)
        out[f"{mode}_Q"], out[f"{mode}_P"], out[f"{mode}_V"], out[f"{mode}_losses"] = Qs[0], Ps[0], sd["V"], sl
        print("multibatch", mode, sl[:3], sl[-1])
    np.savez_compressed(os.path.join(OUT, "multibatch_k8.npz"), **out)


def case_multihead_run():
    """c3-shaped miniature: ks=2..5, N=300, M=1531 (not a multiple of 4), b=128, 2 epochs."""
    N, M, Hd, C, b, ep, seed, lr = 300, 1531, 256, 8, 128, 2, 7, 2e-3
    mn, mx = 2, 5
    ks = list(range(mn, mx + 1))
    G = synth(N, M, 4, seed=99, missing=0.02)
    rng = np.random.default_rng(6)
    V_MC = (rng.standard_normal((M, C)) / math.sqrt(M)).astype(np.float32)
    P_SM = rng.uniform(0.05, 0.95, size=(sum(ks), M)).astype(np.float32)
    out = dict(G_packed=pack_rule(G), N=N, M=M, ks=np.asarray(ks), Hd=Hd, b=b, epochs=ep, seed=seed, lr=lr, V0=V_MC, P0=P_SM)
    Qs, Ps, sd, sl = run_reference_training(G, V_MC, P_SM, None, mn, mx, Hd, ep, b, lr, seed, "hi")
    for h in range(len(ks)):
        out[f"hi_Q{h}"], out[f"hi_P{h}"] = Qs[h], Ps[h]
    out["hi_V"], out["hi_losses"] = sd["V"], sl
    print("multihead", sl[:2], sl[-1])
    np.savez_compressed(os.path.join(OUT, "multihead_run.npz"), **out)


def case_ddp(world=2):
    """DDP emulation on CPU (NCCL cannot run here): one shared model, per-rank batches from
    torch's DistributedSampler(shuffle=True, seed) with set_epoch never called (loaders.py:27),
    per-rank batch = batch_size//world (neural_admixture.py:287), gradients averaged (DDP mean)."""
    from torch.utils.data.distributed import DistributedSampler
    N, M, K, Hd, C, batch, ep, seed, lr = 203, 1024, 4, 128, 8, 64, 2, 11, 2e-3
    G = synth(N, M, K, seed=321)
    rng = np.random.default_rng(8)
    V_MC = (rng.standard_normal((M, C)) / math.sqrt(M)).astype(np.float32)
    P_SM = rng.uniform(0.05, 0.95, size=(K, M)).astype(np.float32)
    b_local = batch // world
    with precision("hi"):
        torch.manual_seed(seed)
        model = Q_P(Hd, C, V=torch.tensor(V_MC.copy()), P=torch.tensor(P_SM.copy()), ks_list=[K])
        opt = model.create_custom_adam(device=torch.device("cpu"), lr=lr)
        loss_fn = torch.nn.BCELoss(reduction="sum")
        Gt = torch.tensor(G)
        samplers = [DistributedSampler(range(N), num_replicas=world, rank=r, shuffle=True, seed=seed) for r in range(world)]
        rank_orders, losses0 = [], []
        for e in range(ep):
            idx = [list(iter(s)) for s in samplers]
            if e == 0:
                rank_orders = np.asarray(idx, dtype=np.int64)
            nsteps = math.ceil(len(idx[0]) / b_local)
            for s in range(nsteps):
                acc = None
                for r in range(world):
                    bi = idx[r][s * b_local:(s + 1) * b_local]
                    opt.zero_grad(set_to_none=True)
                    (recs, _), X = model(Gt[bi])
                    loss = sum(loss_fn(rec, X) for rec in recs)
                    loss.backward()
                    if r == 0:
                        losses0.append(float(loss.item()))
                    gr = [p.grad.detach().clone() for p in model.parameters()]
                    acc = gr if acc is None else [a + g_ for a, g_ in zip(acc, gr)]
                for p, a in zip(model.parameters(), acc):
                    p.grad = a / world
                opt.step()
                model.restrict_P()
        with torch.no_grad():
            model.return_func = model._return_infer
            probs, _ = model(Gt)
        sd = state_np(model)
    np.savez_compressed(os.path.join(OUT, f"ddp_w{world}.npz"), G_packed=pack_rule(G), N=N, M=M, K=K, Hd=Hd, batch=batch,
                        epochs=ep, seed=seed, lr=lr, world=world, V0=V_MC, P0=P_SM, rank_orders=rank_orders,
                        losses_rank0=np.asarray(losses0), Q=probs[0].numpy(), P=sd["decoders.decoders.0.weight"], V=sd["V"])
    print("ddp", world, losses0[:2], losses0[-1])


def case_supervised():
    """Supervised mode through the reference's own train() (model/train.py:74-83 P init from class means of
    the raw codes; neural_admixture.py:460-474 loss = BCE + 100 * CrossEntropy(sum) on the softmax output)."""
    from neural_admixture.model.train import train as ref_train
    N, M, K, Hd, C, b, ep, seed, lr = 240, 1024, 4, 128, 8, 100, 3, 13, 2e-3
    G = synth(N, M, K, seed=77, missing=0.02)
    rng = np.random.default_rng(9)
    names = np.asarray(["POP_C", "POP_A", "POP_D", "POP_B"])
    pops = names[rng.integers(0, K, size=N)]
    Vt = (rng.standard_normal((C, M)) / math.sqrt(M)).astype(np.float32)
    out = dict(G_packed=pack_rule(G), N=N, M=M, K=K, Hd=Hd, b=b, epochs=ep, seed=seed, lr=lr, Vt=Vt, pops=pops)
    for mode in ("hi", "med"):
        step_losses = []
        orig = NeuralAdmixture._run_step_supervised

        def wrapped(self, x, y, _orig=orig, _sl=step_losses):
            loss = _orig(self, x, y)
            _sl.append(float(loss.item()))
            return loss
        NeuralAdmixture._run_step_supervised = wrapped
        try:
            with precision(mode):
                torch.manual_seed(seed)
                Ps, Qs, model = ref_train(ep, b, lr, K, seed, torch.tensor(G), torch.device("cpu"), 0, Hd, True,
                                          Vt.copy(), list(pops), None, None, C)
        finally:
            NeuralAdmixture._run_step_supervised = orig
        sd = state_np(model)
        out[f"{mode}_Q"], out[f"{mode}_P"], out[f"{mode}_V"] = Qs[0], Ps[0], sd["V"]
        out[f"{mode}_losses"] = np.asarray(step_losses)
        print("supervised", mode, step_losses[:2], step_losses[-1])
    np.savez_compressed(os.path.join(OUT, "supervised_k4.npz"), **out)


def case_long_horizon():
    """The horizon BASELINE's second metric is quoted on: a DEFAULT run is 250 epochs (entry.py:27, loop
    model/neural_admixture.py:365-366).  (1) the bundled demo, K=3, 250 epochs, from the SAME V / P_init as demo_k3.npz (read
    back from that fixture, so both fixtures describe one run); (2) the multibatch miniature (N=1000, M=2048, K=8, b=400) for 60
    epochs = 180 steps with the sampler's own orders.  Both twice: "hi" (true fp32, the target) and "med" (the reference as it
    ships: bf16 matmuls) -- the distance between the two is the yardstick the end-of-run tolerances are stated against."""
    d = np.load(os.path.join(OUT, "demo_k3.npz"))
    seed, K, Hd, lr = int(d["seed"]), int(d["K"]), int(d["Hd"]), float(d["lr"])
    ref_utils.set_seed(seed)
    data, _, N, M = ref_utils.read_data("/root/reference/demo/data/demo_data.bed")
    V_MC = np.ascontiguousarray(d["Vt"].T.astype(np.float32))
    P_SM = d["P_init"].astype(np.float32)
    out = dict(epochs=250, seed=seed, K=K, Hd=Hd, lr=lr)
    for mode in ("hi", "med"):
        Qs, Ps, sd, sl = run_reference_training(data, V_MC, P_SM, K, None, None, Hd, 250, 800, lr, seed, mode)
        out[f"{mode}_Q"], out[f"{mode}_P"], out[f"{mode}_V"], out[f"{mode}_losses"] = Qs[0], Ps[0], sd["V"], sl
        out[f"{mode}_loglik"] = np.float64(ref_cy.loglikelihood(data, np.ascontiguousarray(Ps[0].astype(np.float64)),
                                                                np.ascontiguousarray(Qs[0].astype(np.float64)), K))
        print("demo e250", mode, sl[0], sl[-1], out[f"{mode}_loglik"])
    np.savez_compressed(os.path.join(OUT, "demo_k3_e250.npz"), **out)

    m = np.load(os.path.join(OUT, "multibatch_k8.npz"))
    N, M, K, Hd, b, seed, lr, ep = int(m["N"]), int(m["M"]), int(m["K"]), int(m["Hd"]), int(m["b"]), int(m["seed"]), float(m["lr"]), 60
    G = synth(N, M, K, seed=1234)
    assert np.array_equal(pack_rule(G), m["G_packed"])             # the same matrix as multibatch_k8.npz (inputs live there)
    out = dict(epochs=ep, N=N, M=M, K=K, Hd=Hd, b=b, seed=seed, lr=lr)
    for mode in ("hi", "med"):
        Qs, Ps, sd, sl = run_reference_training(G, m["V0"], m["P0"], K, None, None, Hd, ep, b, lr, seed, mode)
        out[f"{mode}_Q"], out[f"{mode}_P"], out[f"{mode}_V"], out[f"{mode}_losses"] = Qs[0], Ps[0], sd["V"], sl
        out[f"{mode}_loglik"] = np.float64(ref_cy.loglikelihood(G, np.ascontiguousarray(Ps[0].astype(np.float64)),
                                                                np.ascontiguousarray(Qs[0].astype(np.float64)), K))
        print("multibatch e60", mode, sl[:2], sl[-1], out[f"{mode}_loglik"])
    np.savez_compressed(os.path.join(OUT, "multibatch_k8_e60.npz"), **out)


# ------------------------------------------------------------------------------------------
# r06: BASELINE configs[1] at FULL WIDTH (2504 x 600k, K = 7, batch 800: steps of 800/800/800/104) against the reference
# itself.  Inputs are regenerated from a seed on both sides (tests/golden/seeded_inputs.py): the fixtures hold the
# reference's outputs and a sha256 of every regenerated input, nothing else.
C2 = dict(N=2504, M=600_000, K=7, seed=2026, Hd=1024, C=8, b=800, lr=2e-3, epochs=5, run_seed=42, nrows=4096)


def _c2_matrix():
    import seeded_inputs as SI
    G = SI.genotypes(C2["N"], C2["M"], C2["K"], C2["seed"])
    return SI, G


def _sampled(SI, out, tag, P_Mk, V_MC):
    """Rows of P / V at seeded SNP indices + float64 column sums over all SNPs."""
    rows = SI.sample_rows(C2["M"], C2["nrows"], C2["seed"])
    out[f"{tag}_P_rows"] = P_Mk[rows]
    out[f"{tag}_P_colsum"] = P_Mk.astype(np.float64).sum(0)
    if V_MC is not None:
        out[f"{tag}_V_rows"] = V_MC[rows]
        out[f"{tag}_V_colsum"] = V_MC.astype(np.float64).sum(0)
        out[f"{tag}_V_abssum"] = np.abs(V_MC.astype(np.float64)).sum(0)


def case_c2_trajectory():
    """launch_training (model/neural_admixture.py:324-392) for 5 epochs = 20 steps from a seeded V0 / P0, 'hi' and 'med'."""
    SI, G = _c2_matrix()
    V0, P0 = SI.init_v_p(C2["M"], C2["C"], C2["K"], C2["seed"])
    out = dict(sha_G=SI.sha(G), sha_V0=SI.sha(V0), sha_P0=SI.sha(P0), **{k: v for k, v in C2.items()})
    for mode in ("hi", "med"):
        Qs, Ps, sd, sl = run_reference_training(G, V0, P0, C2["K"], None, None, C2["Hd"], C2["epochs"], C2["b"], C2["lr"],
                                                C2["run_seed"], mode)
        out[f"{mode}_Q"], out[f"{mode}_losses"] = Qs[0], sl
        _sampled(SI, out, mode, Ps[0], sd["V"])
        out[f"{mode}_loglik"] = np.float64(ref_cy.loglikelihood(G, np.ascontiguousarray(Ps[0].astype(np.float64)),
                                                                np.ascontiguousarray(Qs[0].astype(np.float64)), C2["K"]))
        print("c2 trajectory", mode, sl[:4], sl[-1], out[f"{mode}_loglik"], flush=True)
    np.savez_compressed(os.path.join(OUT, "c2_trajectory.npz"), **out)


def case_c2_trajectory_warm():
    """As case_c2_trajectory, from a start inside the regime a real run trains in (seeded_inputs.warm_v_p: P near the true
    frequencies, V spanning the signal directions), reproducible on both sides to the bit -- the engine alone in the mixture-init
    regime, where c2_end_to_end also carries the (irreproducible) noise component of the RSVD."""
    SI, G = _c2_matrix()
    V0, P0 = SI.warm_v_p(C2["N"], C2["M"], C2["K"], C2["C"], C2["seed"])
    out = dict(sha_G=SI.sha(G), sha_V0=SI.sha(V0), sha_P0=SI.sha(P0), **{k: v for k, v in C2.items()})
    for mode in ("hi", "med"):
        Qs, Ps, sd, sl = run_reference_training(G, V0, P0, C2["K"], None, None, C2["Hd"], C2["epochs"], C2["b"], C2["lr"],
                                                C2["run_seed"], mode)
        out[f"{mode}_Q"], out[f"{mode}_losses"] = Qs[0], sl
        _sampled(SI, out, mode, Ps[0], sd["V"])
        out[f"{mode}_loglik"] = np.float64(ref_cy.loglikelihood(G, np.ascontiguousarray(Ps[0].astype(np.float64)),
                                                                np.ascontiguousarray(Qs[0].astype(np.float64)), C2["K"]))
        print("c2 trajectory warm", mode, sl[:4], sl[-1], out[f"{mode}_loglik"], flush=True)
    np.savez_compressed(os.path.join(OUT, "c2_trajectory_warm.npz"), **out)


def case_c2_end_to_end():
    """The whole default pipeline on the same matrix: the reference's RSVD (src/svd.py:39-83), then its train()
    (model/train.py:19-149: PCA projection, GaussianMixture, launch_training, log-likelihood), 5 epochs.  Kept: a sample of Vt,
    the mixture's means, final Q, sampled P rows, the log-likelihood it printed.  'hi' and 'med'."""
    from neural_admixture.model import train as ref_train_mod
    from sklearn.mixture import GaussianMixture
    SI, G = _c2_matrix()
    ref_utils.set_seed(C2["run_seed"])
    Vt = RSVD(G, C2["N"], C2["M"], C2["C"], C2["run_seed"])                 # [C, M]
    rows = SI.sample_rows(C2["M"], C2["nrows"], C2["seed"])
    out = dict(sha_G=SI.sha(G), Vt_rows=Vt[:, rows].astype(np.float32), Vt_colsum=Vt.astype(np.float64).sum(1),
               Vt_abssum=np.abs(Vt.astype(np.float64)).sum(1), **{k: v for k, v in C2.items()})
    # the first 20 singular values of the sketch are not returned by RSVD; its Vt alone defines the init.  Gram matrix of the
    # projected samples (what the mixture sees) as a summary of the spectrum:
    captured = {}
    real_fit = GaussianMixture.fit

    def fit(self, X, y=None):
        r = real_fit(self, X, y)
        captured["means"], captured["lower_bound"], captured["n_iter"] = self.means_.copy(), self.lower_bound_, self.n_iter_
        captured["X_pca_rows"] = np.asarray(X[:256]).copy()
        return r

    class _Utils:                                                            # train.py:139: utils.loglikelihood(...)
        @staticmethod
        def loglikelihood(data, P, Q, K):
            captured["loglik"] = float(ref_cy.loglikelihood(data, P, Q, K))
            return captured["loglik"]
    real_utils = ref_train_mod.utils
    for mode in ("hi", "med"):
        GaussianMixture.fit = fit
        ref_train_mod.utils = _Utils
        try:
            with precision(mode):
                torch.manual_seed(C2["run_seed"])
                Ps, Qs, model = ref_train_mod.train(C2["epochs"], C2["b"], C2["lr"], C2["K"], C2["run_seed"], torch.tensor(G),
                                                    torch.device("cpu"), 0, C2["Hd"], True, Vt.copy(), None, None, None, C2["C"])
        finally:
            GaussianMixture.fit = real_fit
            ref_train_mod.utils = real_utils
        out[f"{mode}_Q"], out[f"{mode}_loglik"] = Qs[0], np.float64(captured["loglik"])
        _sampled(SI, out, mode, Ps[0], state_np(model)["V"])
        if mode == "hi":
            out["gmm_means"], out["gmm_lower_bound"], out["gmm_n_iter"] = captured["means"], captured["lower_bound"], captured["n_iter"]
            out["X_pca_rows"] = captured["X_pca_rows"]
        print("c2 end-to-end", mode, captured["loglik"], captured["lower_bound"], captured["n_iter"], flush=True)
    np.savez_compressed(os.path.join(OUT, "c2_end_to_end.npz"), **out)


def case_c2_end_to_end_t4():
    """The yardstick of c2_end_to_end: the SAME reference run ('hi', true fp32) with torch's intra-op pool at 4 threads instead of 8 --
    nothing changes but the summation order inside the reference's own fp32 operators (RSVD, projection and mixture fit are computed
    exactly as in case_c2_end_to_end: they do not run on torch's pool).  From the mixture init the run sits where Adam turns rounding-level
    differences of near-zero gradients into steps of +-lr, so two fp32 evaluations of the reference drift apart; the distance between
    them is what any other fp32 implementation can be held to."""
    from neural_admixture.model import train as ref_train_mod
    SI, G = _c2_matrix()
    ref_utils.set_seed(C2["run_seed"])
    Vt = RSVD(G, C2["N"], C2["M"], C2["C"], C2["run_seed"])
    e2e = np.load(os.path.join(OUT, "c2_end_to_end.npz"))
    rows = SI.sample_rows(C2["M"], C2["nrows"], C2["seed"])
    assert np.array_equal(Vt[:, rows].astype(np.float32), e2e["Vt_rows"])           # the same V as the 8-thread fixture
    out = dict(sha_G=SI.sha(G), threads=4)
    torch.set_num_threads(4)
    try:
        with precision("hi"):
            torch.manual_seed(C2["run_seed"])
            Ps, Qs, model = ref_train_mod.train(C2["epochs"], C2["b"], C2["lr"], C2["K"], C2["run_seed"], torch.tensor(G),
                                                torch.device("cpu"), 0, C2["Hd"], True, Vt.copy(), None, None, None, C2["C"])
    finally:
        torch.set_num_threads(8)
    out["hi_Q"] = Qs[0]
    _sampled(SI, out, "hi", Ps[0], state_np(model)["V"])
    print("c2 end-to-end, 4 threads vs 8: max |dQ|", np.abs(Qs[0] - e2e["hi_Q"]).max(), "mean", np.abs(Qs[0] - e2e["hi_Q"]).mean(),
          "max |dP|", np.abs(out["hi_P_rows"] - e2e["hi_P_rows"]).max(), "max |dV|", np.abs(out["hi_V_rows"] - e2e["hi_V_rows"]).max(), flush=True)
    np.savez_compressed(os.path.join(OUT, "c2_end_to_end_t4.npz"), **out)


def case_c4_trajectory(name="c4_trajectory", N=8000, M=500_000, K=8, seed=2027):
    """configs[3]'s MODEL at its width against the reference itself: K = 8, M = 500k, batch 800 -- the bench's step -- on 8000 seeded samples
    (the reference needs ~45 s per 10 steps here; the full 100k rows would take 9 minutes per epoch), one epoch = 10 steps from a seeded
    V0 / P0.  'hi' outputs stored; of 'med' (the reference as it ships) only its distances from 'hi' -- the yardstick.
    c5_trajectory: the same for configs[4]'s model -- K = 16 (pass 2's two-k-slot form), M = 1M -- on 2400 samples = 3 steps."""
    import seeded_inputs as SI
    C = 8
    G = SI.genotypes(N, M, K, seed)
    V0, P0 = SI.init_v_p(M, C, K, seed)
    out = dict(sha_G=SI.sha(G), sha_V0=SI.sha(V0), sha_P0=SI.sha(P0), N=N, M=M, K=K, C=C, seed=seed, Hd=1024, b=800, lr=2e-3, epochs=1, run_seed=42, nrows=4096)
    rows = SI.sample_rows(M, 4096, seed)
    res = {}
    for mode in ("hi", "med"):
        Qs, Ps, sd, sl = run_reference_training(G, V0, P0, K, None, None, 1024, 1, 800, 2e-3, 42, mode)
        res[mode] = (Qs[0], Ps[0][rows], sd["V"][rows], sl, Ps[0].astype(np.float64).sum(0))
        print(name, mode, sl[:3], sl[-1], flush=True)
    out["hi_Q"], out["hi_P_rows"], out["hi_V_rows"], out["hi_losses"], out["hi_P_colsum"] = res["hi"]
    out["med_dQ"] = np.abs(res["med"][0] - res["hi"][0]).max()
    out["med_dP"] = np.abs(res["med"][1] - res["hi"][1]).max()
    out["med_dV"] = np.abs(res["med"][2] - res["hi"][2]).max()
    out["med_losses"] = res["med"][3]
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)


def case_c2_multihead():
    """configs[2] at full width: one epoch (4 steps) of ks = 2..10 on the same matrix from a seeded V0 / P0 [54, M], 'hi'."""
    SI, G = _c2_matrix()
    ks = list(range(2, 11))
    V0, P0 = SI.init_v_p(C2["M"], C2["C"], sum(ks), C2["seed"] + 1)
    out = dict(sha_G=SI.sha(G), sha_V0=SI.sha(V0), sha_P0=SI.sha(P0), ks=np.asarray(ks), init_seed=C2["seed"] + 1,
               **{k: v for k, v in C2.items() if k not in ("K", "epochs")})
    Qs, Ps, sd, sl = run_reference_training(G, V0, P0, None, 2, 10, C2["Hd"], 1, C2["b"], C2["lr"], C2["run_seed"], "hi")
    rows = SI.sample_rows(C2["M"], 1024, C2["seed"])
    for h in range(len(ks)):
        out[f"hi_Q{h}"], out[f"hi_P{h}_rows"], out[f"hi_P{h}_colsum"] = Qs[h], Ps[h][rows], Ps[h].astype(np.float64).sum(0)
    out["hi_V_rows"], out["hi_V_colsum"], out["hi_losses"] = sd["V"][rows], sd["V"].astype(np.float64).sum(0), sl
    print("c2 multihead", sl, flush=True)
    np.savez_compressed(os.path.join(OUT, "c2_multihead.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    import resource                                         # the full-width cases hold tens of GB: fail with MemoryError, not the OOM killer
    resource.setrlimit(resource.RLIMIT_AS, (58 << 30, 58 << 30))
    cases = {
        "pack": case_pack,
        "bce": case_bce_elementwise,
        "one_step_k3": lambda: one_step_case("one_step_k3", 64, 509, [3], 64, 8, seed=3),
        "one_step_multihead": lambda: one_step_case("one_step_multihead", 64, 509, [2, 3, 4], 64, 8, seed=4),
        "one_step_k8_h1024": lambda: one_step_case("one_step_k8_h1024", 48, 777, [8], 1024, 8, seed=5),
        "one_step_edge": lambda: one_step_case("one_step_edge", 64, 509, [3], 64, 8, seed=6, edge=True),
        "multibatch": case_multibatch,
        "multihead_run": case_multihead_run,
        "ddp": lambda: case_ddp(2),
        "ddp4": lambda: case_ddp(4),                       # N = 203 over 4 ranks: one wrapped duplicate, 16-row batches, ragged last one (3 rows)
        "demo": case_demo,
        "supervised": case_supervised,
        "one_step_supervised": lambda: one_step_case("one_step_supervised", 64, 509, [5], 64, 8, seed=8, sup=True),
        # BASELINE configs[1] / configs[2] model shapes (single head K=7; heads K=2..10) at fixture size
        "one_step_k7_h1024": lambda: one_step_case("one_step_k7_h1024", 56, 1021, [7], 1024, 8, seed=9),
        "long_horizon": case_long_horizon,                 # r04: the default 250-epoch horizon (demo) and 60 epochs of the multibatch miniature
        "one_step_heads2to10": lambda: one_step_case("one_step_heads2to10", 40, 613, list(range(2, 11)), 256, 8, seed=10),
        # r05: configs[4]'s model shape (K = 16: the two-k-slot variant of pass 2, 7 MFMAs per tile) and the smallest K that uses it
        "one_step_k16_h1024": lambda: one_step_case("one_step_k16_h1024", 48, 1021, [16], 1024, 8, seed=11),
        "one_step_k9": lambda: one_step_case("one_step_k9", 64, 509, [9], 64, 8, seed=12),
        # r06: configs[1] / configs[2] at full width against the reference itself (minutes each; inputs regenerated from a seed)
        "c2_trajectory": case_c2_trajectory,
        "c2_trajectory_warm": case_c2_trajectory_warm,
        "c2_end_to_end": case_c2_end_to_end,
        "c2_end_to_end_t4": case_c2_end_to_end_t4,
        "c2_multihead": case_c2_multihead,
        "c4_trajectory": case_c4_trajectory,
        "c5_trajectory": lambda: case_c4_trajectory("c5_trajectory", N=2400, M=1_000_000, K=16, seed=2028),
    }
    for name in (sys.argv[1:] or list(cases)):            # no arguments: every fixture; else only the named cases
        cases[name]()
    print("done ->", OUT)
