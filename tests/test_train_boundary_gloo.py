"""World-size-2 runs THROUGH the drop-in boundary ``train(...)`` on CPU (gloo): the master-only initialisation, the barrier,
``dist.broadcast(P_init)`` / ``dist.broadcast(V)`` / ``dist.broadcast(pops)`` of the reference (model/train.py:86-113), then
the sharded training and the master-only report.  The kernels are replaced by the oracle (tests/fake_engine.py); the
broadcast branch, the message plan, the sharding and the reports are the product code.  Rank 1 is handed a WRONG V (zeros)
and no usable P on purpose: only the broadcasts can make the two ranks agree."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import nadm_oracle as O  # noqa: E402


def _inputs():
    d = np.load(os.path.join(HERE, "golden", "ddp_w2.npz"))
    G = O.unpack2bit(d["G_packed"], int(d["M"]))
    return d, G


def _train_worker(rank, world, port, out_path, K, parallelism, supervised):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na
    from fake_engine import OracleEngine, OracleSnpEngine
    na.NeuralAdmixture.engine_cls = OracleEngine
    na.NeuralAdmixture.engine_snp_cls = OracleSnpEngine
    d, G = _inputs()
    N = int(d["N"])
    V_CM = np.ascontiguousarray(d["V0"].T)                               # the boundary takes V as [C, M] (svd.py:83)
    if rank != 0:
        V_CM = np.zeros_like(V_CM)                                       # must be overwritten by the broadcast from rank 0
    pops = [f"pop{(i * 7) % K}" for i in range(N)] if supervised else None
    Ps, Qs, model = na.train(int(d["epochs"]), int(d["batch"]), float(d["lr"]), K, int(d["seed"]), torch.from_numpy(G),
                             torch.device("cpu"), world, int(d["Hd"]), rank == 0, V_CM, pops, None, None, 8, parallelism=parallelism)
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].numpy())
    else:
        assert Ps == [] and Qs == []                                     # neural_admixture.py:525-529
    # every rank ends with the same parameters: they all started from the broadcast P_init / V
    sm = model.engine.small.clone()
    ref = sm.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(sm, ref)
    dist.barrier()
    dist.destroy_process_group()


def _expected(K, supervised):
    """The same run restated with the oracle: master's initialisation, then the DDP emulation (world 2)."""
    from neural_admixture_amd.train import gmm_p_init, supervised_init
    d, G = _inputs()
    V_CM = np.ascontiguousarray(d["V0"].T)
    labels = None
    if supervised:
        pops = [f"pop{(i * 7) % K}" for i in range(int(d["N"]))]
        labels, P_init = supervised_init(G, pops, K)
    else:
        P_init = gmm_p_init(G, V_CM, K, None, None, 8, int(d["seed"]), None)
    p = O.make_params(int(d["seed"]), d["V0"].copy(), P_init.astype(np.float32), int(d["Hd"]), [K])
    p, Qs, _ = O.train_run(G, p, int(d["epochs"]), int(d["batch"]), float(d["lr"]), int(d["seed"]), world=2, labels=labels)
    return p, Qs


@pytest.mark.parametrize("parallelism,supervised", [("dp", False), ("snp", False), ("dp", True)])
def test_world2_train_runs_the_init_broadcasts(tmp_path, parallelism, supervised):
    world, K = 2, 4
    port = 34500 + (os.getpid() % 2000) + 3 * (parallelism == "snp") + 5 * supervised
    out = str(tmp_path / "train_w2.npz")
    mp.spawn(_train_worker, args=(world, port, out, K, parallelism, supervised), nprocs=world, join=True)
    r = np.load(out)
    p, Qs = _expected(K, supervised)
    tol = 2e-3 if supervised else 1e-4                                   # supervised: the raw-code init saturates R (chaotic, see test_oracle_golden)
    assert np.abs(r["Q"] - Qs[0]).max() < tol
    assert np.abs(r["P"] - p.P[0]).max() < tol
    assert np.abs(r["V"] - p.V).max() < tol


def _k17_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    import neural_admixture_amd as na
    from fake_engine import OracleSnpEngine
    na.NeuralAdmixture.engine_snp_cls = OracleSnpEngine
    d, G = _inputs()
    msgs = []

    class Grab(logging.Handler):
        def emit(self, rec):
            msgs.append(rec.getMessage())
    logging.getLogger("neural_admixture_amd.train").addHandler(Grab())
    Ps, Qs, model = na.train(1, int(d["batch"]), float(d["lr"]), 17, int(d["seed"]), torch.from_numpy(G), torch.device("cpu"), world,
                             32, rank == 0, np.ascontiguousarray(d["V0"].T), None, None, None, 8, parallelism="snp")
    if rank == 0:
        ll = [float(m.split(":")[1].strip().rstrip(".")) for m in msgs if "Log-likelihood" in m]
        assert len(ll) == 1
        assert Ps[0].shape == (int(d["M"]), 17)                           # the gathered matrix, not the rank's slice
        np.savez(out_path, ll=ll[0], ref=O.loglikelihood(G, Ps[0], Qs[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_snp_sharded_run_with_a_head_wider_than_16_reports_its_loglikelihood(tmp_path):
    """K > 16 has no HIP log-likelihood kernel, so the report falls back to the row-chunked float64 reduction; on an
    SNP-sharded engine (all rows, M/world SNPs) that fallback has to read the host rows, not the rank's slice, because P is the
    gathered [M, k] matrix (round-1 advisor finding: the run crashed here after training and before anything was written)."""
    port = 36500 + (os.getpid() % 2000)
    out = str(tmp_path / "k17.npz")
    mp.spawn(_k17_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    assert abs(float(r["ll"]) - float(r["ref"])) <= 1e-6 * abs(float(r["ref"])) + 1.0   # the log line prints with 6 decimals
