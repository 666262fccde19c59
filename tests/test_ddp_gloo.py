"""World-size-2 data-parallel path on CPU (gloo): sharding by DistributedSampler semantics,
per-rank batch = batch//world, gradient all-reduce + 1/world (DDP mean), final-Q gather.
The kernels are replaced by the oracle (tests/fake_engine.py); everything else is product code.
Checked against the DDP emulation captured from the reference (tests/golden/ddp_w2.npz)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na
    from fake_engine import OracleEngine
    from oracle import nadm_oracle as O
    d = np.load(os.path.join(HERE, "golden", "ddp_w2.npz"))
    G = O.unpack2bit(d["G_packed"], int(d["M"]))
    na.NeuralAdmixture.engine_cls = OracleEngine
    tr = na.NeuralAdmixture(int(d["K"]), int(d["epochs"]), int(d["batch"]), float(d["lr"]), torch.device("cpu"), int(d["seed"]),
                            world, rank == 0, None, None, None, loss_mode="always")
    Qs, Ps, model = tr.launch_training(torch.from_numpy(d["P0"]), torch.from_numpy(G), int(d["Hd"]), 8, torch.from_numpy(d["V0"]),
                                       int(d["M"]), int(d["N"]), None)
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].numpy(),
                 losses=np.asarray([tr.epoch_losses[e] for e in range(int(d["epochs"]))]))
    else:
        assert Qs == [] and Ps == []              # non-master returns empty lists (neural_admixture.py:525-529)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_matches_reference_ddp_emulation(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "ddp_out.npz")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = np.load(out)
    d = np.load(os.path.join(HERE, "golden", "ddp_w2.npz"))
    assert np.abs(r["Q"] - d["Q"]).max() < 1e-4
    assert np.abs(r["P"] - d["P"]).max() < 1e-5
    assert np.abs(r["V"] - d["V"]).max() < 1e-4
    assert np.allclose(r["losses"], d["losses_rank0"].reshape(int(d["epochs"]), -1).sum(1), rtol=1e-6)


def _snp_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na
    from fake_engine import OracleSnpEngine
    from oracle import nadm_oracle as O
    d = np.load(os.path.join(HERE, "golden", "ddp_w2.npz"))
    G = O.unpack2bit(d["G_packed"], int(d["M"]))
    na.NeuralAdmixture.engine_snp_cls = OracleSnpEngine
    tr = na.NeuralAdmixture(int(d["K"]), int(d["epochs"]), int(d["batch"]), float(d["lr"]), torch.device("cpu"), int(d["seed"]),
                            world, rank == 0, None, None, None, loss_mode="always", parallelism="snp")
    Qs, Ps, model = tr.launch_training(torch.from_numpy(d["P0"]), torch.from_numpy(G), int(d["Hd"]), 8, torch.from_numpy(d["V0"]),
                                       int(d["M"]), int(d["N"]), None)
    assert tr.engine.m1 - tr.engine.m0 < int(d["M"])       # really sharded
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].numpy(),
                 losses=np.asarray([tr.epoch_losses[e] for e in range(int(d["epochs"]))]))
    else:
        assert Qs == [] and Ps == []
    dist.barrier()
    dist.destroy_process_group()


def test_world2_snp_sharded_equals_the_sample_sharded_trajectory(tmp_path):
    """parallelism="snp": every rank owns half of the SNPs and processes the global batch (the union of the two ranks'
    DistributedSampler batches); two small all-reduces per step.  Up to summation order this is the reference's DDP
    trajectory, so it is checked against the same fixture (the DDP emulation captured from the reference)."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    out = str(tmp_path / "snp_out.npz")
    mp.spawn(_snp_worker, args=(world, port, out), nprocs=world, join=True)
    r = np.load(out)
    d = np.load(os.path.join(HERE, "golden", "ddp_w2.npz"))
    assert np.abs(r["Q"] - d["Q"]).max() < 1e-4
    assert np.abs(r["P"] - d["P"]).max() < 1e-5
    assert np.abs(r["V"] - d["V"]).max() < 1e-4
    assert r["P"].shape == d["P"].shape and r["V"].shape == d["V"].shape
