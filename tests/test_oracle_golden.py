"""Pin the CPU oracle (oracle/nadm_oracle.py) against every golden fixture captured from the
reference (tests/golden/make_golden.py).  Tolerances are fp32-rounding class: the fixtures are the
reference forced to true-fp32 matmuls ("hi"); the "med" variants (the reference's own bf16 path)
are only used to show how far the reference is from itself."""
import os

import numpy as np
import pytest

from oracle import nadm_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def mx(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def rel(a, b):
    return mx(a, b) / (float(np.abs(b).max()) + 1e-30)


def test_pack_layout():
    d = np.load(f"{G}/pack_layout.npz")
    assert np.array_equal(O.pack2bit(d["G"]), d["packed"])
    assert np.array_equal(O.pack2bit(d["G_hibits"]), d["packed_hibits"])       # &3 masking
    assert np.array_equal(O.unpack2bit(d["packed"], d["G"].shape[1]), d["G"])
    # empty / ragged
    assert O.pack2bit(np.zeros((0, 5), np.uint8)).shape == (0, 2)
    assert O.pack2bit(np.zeros((3, 0), np.uint8)).shape == (3, 0)


def test_bce_elementwise_semantics():
    """-100 log clamp, 1e-12 denominator, inclusive mask on the pre-clamp value."""
    d = np.load(f"{G}/bce_elementwise.npz")
    r_raw, x = d["r_raw"], d["x"]
    R = np.clip(r_raw, np.float32(0), np.float32(1))
    with np.errstate(divide="ignore", invalid="ignore"):
        loss = -(x * np.maximum(np.log(R), O.LOG_CLAMP) + (1 - x) * np.maximum(np.log1p(-R), O.LOG_CLAMP))
        grad = (R - x) / np.maximum((1 - R) * R, O.BCE_EPS)
    grad[(r_raw < 0) | (r_raw > 1)] = 0
    assert np.allclose(loss, d["loss"], rtol=2e-6, atol=1e-30)
    assert np.allclose(grad, d["grad"], rtol=2e-6, atol=0)


@pytest.mark.parametrize("name", ["one_step_k3", "one_step_multihead", "one_step_k8_h1024", "one_step_edge",
                                  "one_step_supervised", "one_step_k7_h1024", "one_step_heads2to10", "one_step_k16_h1024", "one_step_k9"])
def test_one_step(name):
    d = np.load(f"{G}/{name}.npz")
    ks = [int(k) for k in d["ks"]]
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), ks)
    # initial weights: same torch RNG stream as the reference's module construction
    assert mx(p.W1, d["init_common_encoder_0_weight"]) == 0
    assert mx(p.b1, d["init_common_encoder_0_bias"]) == 0
    for h in range(len(ks)):
        assert mx(p.Wk[h], d[f"init_multihead_encoder_heads_{h}_weight"]) == 0
        assert mx(p.bk[h], d[f"init_multihead_encoder_heads_{h}_bias"]) == 0
    edge = name.endswith("edge")
    gtol = 3e-3 if edge else 1e-5      # edge: r within 1e-3 of 1 amplifies GEMM rounding ~1e4x
    opt = O.Adam(p, float(d["lr"]))
    for s in range(3):
        loss, grads, aux = O.step_grads(p, d["G"], d["labels"] if "labels" in d.files else None)
        assert abs(loss - float(d[f"loss{s}"])) / float(d[f"loss{s}"]) < 2e-6
        if s == 0:
            assert mx(aux["Z"], d["Z0"]) < 2e-6
            for h in range(len(ks)):
                assert mx(aux["Qs"][h], d[f"Q0_{h}"]) < 1e-6
                assert rel(grads[f"P{h}"], d[f"grad0_decoders_decoders_{h}_weight"]) < 1e-5
                assert rel(grads[f"Wk{h}"], d[f"grad0_multihead_encoder_heads_{h}_weight"]) < gtol
                assert rel(grads[f"bk{h}"], d[f"grad0_multihead_encoder_heads_{h}_bias"]) < gtol
            assert rel(grads["V"], d["grad0_V"]) < gtol
            assert rel(grads["g"], d["grad0_batch_norm_weight"]) < gtol
            assert rel(grads["W1"], d["grad0_common_encoder_0_weight"]) < gtol
            assert rel(grads["b1"], d["grad0_common_encoder_0_bias"]) < gtol
        opt.step(p, grads)
        if not edge:                   # Adam's first steps are sign-like: ill-conditioned on edge
            assert mx(p.V, d[f"after{s}_V"]) < 2e-6
            assert mx(p.W1, d[f"after{s}_common_encoder_0_weight"]) < 2e-6
            assert mx(p.g, d[f"after{s}_batch_norm_weight"]) < 2e-6
        for h in range(len(ks)):
            tol = 5e-5 if edge else 1e-6
            assert mx(p.P[h], d[f"after{s}_decoders_decoders_{h}_weight"]) < tol


@pytest.mark.parametrize("name", ["one_step_k3", "one_step_multihead", "one_step_k8_h1024", "one_step_k16_h1024"])
def test_reference_shaped_torch_restatement(name):
    """oracle/torch_shape.py -- the reference's own operator sequence on torch CPU ops, bench.py's second CPU baseline -- against the
    tensors captured from the reference: loss of three steps, parameters after each."""
    import torch
    from oracle.torch_shape import TorchShapedModel
    d = np.load(f"{G}/{name}.npz")
    ks = [int(k) for k in d["ks"]]
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), ks)
    torch.set_float32_matmul_precision("highest")
    m = TorchShapedModel(p.V, p.P, p.g, p.W1, p.b1, p.Wk, p.bk, float(d["lr"]))
    Gt = torch.from_numpy(np.ascontiguousarray(d["G"]))
    for s in range(3):
        loss = m.step(Gt)
        assert abs(loss - float(d[f"loss{s}"])) / float(d[f"loss{s}"]) < 2e-6
        assert mx(m.V.detach().numpy(), d[f"after{s}_V"]) < 2e-6
        for h in range(len(ks)):
            assert mx(m.P[h].detach().numpy(), d[f"after{s}_decoders_decoders_{h}_weight"]) < 1e-6


def test_multibatch_trajectory_and_batch_order():
    d = np.load(f"{G}/multibatch_k8.npz")
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), [int(d["K"])])
    orders = []
    p, Qs, losses = O.train_run(Gm, p, int(d["epochs"]), int(d["b"]), float(d["lr"]), int(d["seed"]), record_orders=orders)
    assert np.array_equal(np.asarray(orders), d["orders"])           # RandomSampler two-draw rule
    assert mx(Qs[0], d["hi_Q"]) < 5e-4
    assert mx(p.P[0], d["hi_P"]) < 5e-5
    assert mx(p.V, d["hi_V"]) < 5e-4
    ref = d["hi_losses"].reshape(int(d["epochs"]), -1).sum(1)
    assert np.allclose(losses, ref, rtol=1e-6)
    # the reference's own bf16 ('medium') run is much further from its fp32 run than the oracle is
    assert mx(d["med_Q"], d["hi_Q"]) > 10 * mx(Qs[0], d["hi_Q"])


def test_multihead_trajectory():
    d = np.load(f"{G}/multihead_run.npz")
    ks = [int(k) for k in d["ks"]]
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), ks)
    p, Qs, losses = O.train_run(Gm, p, int(d["epochs"]), int(d["b"]), float(d["lr"]), int(d["seed"]))
    for h in range(len(ks)):
        assert mx(Qs[h], d[f"hi_Q{h}"]) < 1e-4
        assert mx(p.P[h], d[f"hi_P{h}"]) < 1e-5
    assert mx(p.V, d["hi_V"]) < 1e-4
    assert np.allclose(losses, d["hi_losses"].reshape(int(d["epochs"]), -1).sum(1), rtol=1e-6)


@pytest.mark.parametrize("world", [2, 4])
def test_ddp_emulation(world):
    """world 2: 102 rows per rank, batches of 32 (last one 6); world 4: N = 203 is padded to 204 by wrapping one index
    (DistributedSampler, loaders.py:26-27), 51 rows per rank, batches of 64 // 4 = 16 (neural_admixture.py:287), last one 3."""
    d = np.load(f"{G}/ddp_w{world}.npz")
    assert int(d["world"]) == world
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), [int(d["K"])])
    orders = []
    p, Qs, losses = O.train_run(Gm, p, int(d["epochs"]), int(d["batch"]), float(d["lr"]), int(d["seed"]), world=world,
                                record_orders=orders)
    eo = O.EpochOrder(int(d["N"]), int(d["seed"]), world)
    for r in range(world):
        assert np.array_equal(eo.rank_indices(orders[0], r), d["rank_orders"][r])   # DistributedSampler shard
    if world == 4:
        assert len(np.unique(d["rank_orders"])) == int(d["N"]) and d["rank_orders"].size == int(d["N"]) + 1   # one wrapped duplicate
    assert mx(Qs[0], d["Q"]) < 1e-4 and mx(p.P[0], d["P"]) < 1e-5 and mx(p.V, d["V"]) < 1e-4
    assert np.allclose(losses, d["losses_rank0"].reshape(int(d["epochs"]), -1).sum(1), rtol=1e-6)


@pytest.mark.parametrize("ep", [5, 25])
def test_demo_c1(ep):
    d = np.load(f"{G}/demo_k3.npz")
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["Vt"].T, d["P_init"], int(d["Hd"]), [3])
    p, Qs, losses = O.train_run(Gm, p, ep, 800, float(d["lr"]), int(d["seed"]))
    assert mx(Qs[0], d[f"hi_e{ep}_Q"]) < 1e-4
    assert mx(p.P[0], d[f"hi_e{ep}_P"]) < 1e-5
    assert np.allclose(losses, d[f"hi_e{ep}_losses"], rtol=1e-6)
    if ep == 5:
        assert mx(p.V, d["hi_e5_V"]) < 1e-5
        ll = O.loglikelihood(Gm, p.P[0], Qs[0])
        assert abs(ll - float(d["hi_e5_loglik"])) / abs(float(d["hi_e5_loglik"])) < 1e-7
        assert abs(O.hudson_fst(p.P[0][:, 1], p.P[0][:, 0]) - d["hi_e5_fst"][1, 0]) < 1e-5


def test_supervised_run():
    """Supervised mode through the reference's own train(): label mapping, class-mean P init (raw codes, values
    up to 3, train.py:82), BCE + 100*CE.  The init puts most r at the clamp, where one rounding flips the
    gradient mask of an element whose gradient is ~1e3..1e12, so after the first step the reference differs from
    ITSELF by 2e-3 in loss between its fp32 and bf16 runs; only step 0 is a rounding-level pin."""
    d = np.load(f"{G}/supervised_k4.npz")
    N, M, K, Hd = int(d["N"]), int(d["M"]), int(d["K"]), int(d["Hd"])
    Gm = O.unpack2bit(d["G_packed"], M)
    y = O.labels_from_pops(d["pops"])
    assert sorted(set(y.tolist())) == list(range(K))
    P0 = O.supervised_p_init(Gm, y, K)
    assert P0.max() > 1.0                                  # raw codes, not allele frequencies
    p = O.make_params(int(d["seed"]), np.ascontiguousarray(d["Vt"].T), P0, Hd, [K])
    order = O.EpochOrder(N, int(d["seed"]))
    opt = O.Adam(p, float(d["lr"]))
    ls = []
    for _ in range(int(d["epochs"])):
        for idx in O.batches(order.next_epoch(), int(d["b"])):
            loss, g, _ = O.step_grads(p, Gm[idx], y[idx])
            opt.step(p, g)
            ls.append(loss)
    ls, ref, med = np.asarray(ls), d["hi_losses"], d["med_losses"]
    assert abs(ls[0] - ref[0]) / ref[0] < 2e-6
    self_noise = np.abs(med - ref) / ref
    assert np.all(np.abs(ls - ref) / ref < np.maximum(3 * self_noise, 5e-3))


# SURVEY 8c's end-of-run bounds at the horizon BASELINE's wall-clock metric is quoted on (a default run: 250 epochs, entry.py:27)
END_OF_RUN = dict(mean_dq=1e-2, max_dp=1e-2, loglik_rel=1e-4, loss_rel=1e-3)


def check_end_of_run(Q, P, losses, loglik, d, worst_sample_factor=1.0):
    """Q / P / per-epoch loss / log-likelihood of a run against the reference's fp32 ("hi") run of the same length, next to the
    distance of the reference's own bf16 ("med") run from it -- the yardstick: two fp32 summation orders separate over
    hundreds of steps like the reference separates from itself."""
    dq, dp = np.abs(Q - d["hi_Q"]), np.abs(P - d["hi_P"])
    ref_dq = np.abs(d["med_Q"] - d["hi_Q"])
    assert dq.mean() <= END_OF_RUN["mean_dq"], dq.mean()
    assert dp.max() <= END_OF_RUN["max_dp"], dp.max()
    # the yardstick: on average closer to the fp32 run than the reference's own bf16 run is, and so is the single worst sample -- for the
    # pinned CPU oracle (measured r04: demo e250 max 0.058 against the reference's 0.060, mean 5.6e-3 against 7.7e-3) and, since r06, for the
    # GPU production path alike (5.82e-2 / 5.6e-3 there; r04-r05 passed worst_sample_factor = 2 for it)
    assert dq.mean() <= ref_dq.mean() and dq.max() <= worst_sample_factor * ref_dq.max(), (dq.max(), ref_dq.max())
    ref_l = np.asarray(d["hi_losses"], dtype=np.float64).reshape(len(losses), -1).sum(1)
    assert np.max(np.abs(np.asarray(losses) - ref_l) / ref_l) <= END_OF_RUN["loss_rel"]
    assert abs(loglik - float(d["hi_loglik"])) / abs(float(d["hi_loglik"])) <= END_OF_RUN["loglik_rel"]


def test_default_horizon_demo_250_epochs():
    """The bundled demo for the DEFAULT 250 epochs (entry.py:27; neural_admixture.py:365-366) from the reference's own init."""
    dm, d = np.load(f"{G}/demo_k3.npz"), np.load(f"{G}/demo_k3_e250.npz")
    Gm = O.unpack2bit(dm["G_packed"], int(dm["M"]))
    p = O.make_params(int(dm["seed"]), dm["Vt"].T, dm["P_init"], int(dm["Hd"]), [3])
    p, Qs, losses = O.train_run(Gm, p, int(d["epochs"]), 800, float(dm["lr"]), int(dm["seed"]))
    check_end_of_run(Qs[0], p.P[0], losses, O.loglikelihood(Gm, p.P[0], Qs[0]), d)


def test_long_horizon_multibatch_60_epochs():
    """180 steps of the multibatch miniature (N=1000, M=2048, K=8, b=400) with the sampler's own epoch orders."""
    m, d = np.load(f"{G}/multibatch_k8.npz"), np.load(f"{G}/multibatch_k8_e60.npz")
    Gm = O.unpack2bit(m["G_packed"], int(m["M"]))
    p = O.make_params(int(m["seed"]), m["V0"], m["P0"], int(m["Hd"]), [int(m["K"])])
    p, Qs, losses = O.train_run(Gm, p, int(d["epochs"]), int(m["b"]), float(m["lr"]), int(m["seed"]))
    check_end_of_run(Qs[0], p.P[0], losses, O.loglikelihood(Gm, p.P[0], Qs[0]), d)


def test_oracle_first_step_at_configs1_full_width():
    """The oracle pinned against the reference itself at a BASELINE width (r06): the first step (800 rows) of configs[1]'s
    trajectory fixture (2504 x 600k, K = 7; inputs regenerated from the seed, tests/golden/seeded_inputs.py) -- per-step loss against the
    loss the reference's launch_training recorded.  (~1.5 minutes, ~12 GB: the whole 20-step trajectory is the GPU suite's,
    tests/test_c2_full_width.py.)"""
    import sys
    sys.path.insert(0, G)
    import seeded_inputs as SI
    d = np.load(f"{G}/c2_trajectory.npz")
    N, M, K, C = int(d["N"]), int(d["M"]), int(d["K"]), int(d["C"])
    Gm = SI.genotypes(N, M, K, int(d["seed"]), threads=min(16, os.cpu_count() or 8))
    assert SI.sha(Gm) == str(d["sha_G"])
    V0, P0 = SI.init_v_p(M, C, K, int(d["seed"]))
    assert SI.sha(V0) == str(d["sha_V0"]) and SI.sha(P0) == str(d["sha_P0"])
    p = O.make_params(int(d["run_seed"]), V0, P0, int(d["Hd"]), [K])
    perm = O.EpochOrder(N, int(d["run_seed"]), 1).next_epoch()
    loss, grads, _ = O.step_grads(p, Gm[perm[:800]])
    assert abs(loss - float(d["hi_losses"][0])) / float(d["hi_losses"][0]) < 2e-6, (loss, float(d["hi_losses"][0]))
