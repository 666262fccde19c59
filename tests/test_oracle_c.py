"""Pin the C port (oracle/nadm_oracle_c.c, bench.py's cpu_baseline) against the golden vectors."""
import os

import numpy as np
import pytest

from oracle import nadm_oracle as O
from oracle.c_port import CPort

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("name", ["one_step_k3", "one_step_k8_h1024", "one_step_k7_h1024", "one_step_k16_h1024"])
def test_c_port_one_step(name):
    d = np.load(f"{G}/{name}.npz")
    ks = [int(k) for k in d["ks"]]
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), ks)
    cp = CPort(p.V, p.P[0], p.g, p.W1, p.b1, p.Wk[0], p.bk[0])
    Gm = d["G"]
    idx = np.arange(Gm.shape[0])
    for s in range(3):
        out = cp.step(Gm, idx, float(d["lr"]), apply=True, want_grads=(s == 0), want_q=(s == 0))
        assert abs(out["loss"] - float(d[f"loss{s}"])) / float(d[f"loss{s}"]) < 2e-6
        if s == 0:
            assert np.abs(out["Q"] - d["Q0_0"]).max() < 1e-6
            assert rel(out["V"], d["grad0_V"]) < 2e-5
            assert rel(out["P0"], d["grad0_decoders_decoders_0_weight"]) < 1e-5
            assert rel(out["W1"], d["grad0_common_encoder_0_weight"]) < 2e-5
            assert rel(out["Wk0"], d["grad0_multihead_encoder_heads_0_weight"]) < 2e-5
            assert rel(out["g"], d["grad0_batch_norm_weight"]) < 2e-5
            assert rel(out["b1"], d["grad0_common_encoder_0_bias"]) < 2e-5
            assert rel(out["bk0"], d["grad0_multihead_encoder_heads_0_bias"]) < 2e-5
        assert np.abs(cp.a["V"] - d[f"after{s}_V"]).max() < 5e-6
        assert np.abs(cp.a["P"] - d[f"after{s}_decoders_decoders_0_weight"]).max() < 5e-6


def test_c_port_multibatch_trajectory():
    d = np.load(f"{G}/multibatch_k8.npz")
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), [int(d["K"])])
    cp = CPort(p.V, p.P[0], p.g, p.W1, p.b1, p.Wk[0], p.bk[0])
    order = O.EpochOrder(int(d["N"]), int(d["seed"]))
    losses = []
    for _ in range(int(d["epochs"])):
        acc = 0.0
        for idx in O.batches(order.next_epoch(), int(d["b"])):
            acc += cp.step(Gm, idx, float(d["lr"]))["loss"]
        losses.append(acc)
    assert np.allclose(losses, d["hi_losses"].reshape(int(d["epochs"]), -1).sum(1), rtol=2e-6)
    assert np.abs(cp.a["P"] - d["hi_P"]).max() < 1e-4
    assert np.abs(cp.a["V"] - d["hi_V"]).max() < 1e-3
