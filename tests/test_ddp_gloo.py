"""Multi-rank paths on CPU (gloo) at world 2, 4 and 8: sharding by DistributedSampler semantics (wrap-padding when N % world
!= 0), per-rank batch = batch // world (neural_admixture.py:287), gradient all-reduce + 1/world (DDP mean), the deferred
P / V / small-parameter updates of the data-parallel step, the SNP-sharded variant, final-Q gather (wrapped duplicates
included).  The kernels are replaced by the oracle (tests/fake_engine.py); everything else is product code.  Checked
against the DDP emulations captured from the reference (tests/golden/ddp_w2.npz, ddp_w4.npz) and, at the reference's
8-GPU shape (global batch 800 = 100 rows per rank), against the oracle's DDP emulation."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, out_path, fixture="ddp_w2.npz"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na
    from fake_engine import OracleEngine
    from oracle import nadm_oracle as O
    d = np.load(os.path.join(HERE, "golden", fixture))
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


def _snp_worker(rank, world, port, out_path, fixture="ddp_w2.npz"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na
    from fake_engine import OracleSnpEngine
    from oracle import nadm_oracle as O
    d = np.load(os.path.join(HERE, "golden", fixture))
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


@pytest.mark.parametrize("parallelism", ["dp", "snp"])
def test_world4_gloo_matches_reference_ddp_emulation(tmp_path, parallelism):
    """Four ranks, N = 203: DistributedSampler wraps one index to reach 204 (51 rows per rank, loaders.py:26-27), per-rank batch
    64 // 4 = 16 with a ragged last batch of 3, and the final-Q gather writes the duplicated row twice (same value).  Both
    shardings against the SAME emulation captured from the reference (make_golden.py case_ddp(4))."""
    world = 4
    port = (38500 if parallelism == "dp" else 39500) + (os.getpid() % 900)
    out = str(tmp_path / "w4.npz")
    mp.spawn(_worker if parallelism == "dp" else _snp_worker, args=(world, port, out, "ddp_w4.npz"), nprocs=world, join=True)
    r = np.load(out)
    d = np.load(os.path.join(HERE, "golden", "ddp_w4.npz"))
    assert r["Q"].shape == d["Q"].shape == (203, 4)
    assert np.abs(r["Q"] - d["Q"]).max() < 1e-4
    assert np.abs(r["P"] - d["P"]).max() < 1e-5
    assert np.abs(r["V"] - d["V"]).max() < 1e-4
    if parallelism == "dp":                                  # the master logs its own shard's loss (neural_admixture.py:414-417)
        assert np.allclose(r["losses"], d["losses_rank0"].reshape(int(d["epochs"]), -1).sum(1), rtol=1e-6)


def _wN_worker(rank, world, port, out_path, parallelism, N, M, K, Hd, batch, epochs, seed):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na
    from fake_engine import OracleEngine, OracleSnpEngine
    na.NeuralAdmixture.engine_cls = OracleEngine
    na.NeuralAdmixture.engine_snp_cls = OracleSnpEngine
    na.NeuralAdmixture.dp_buckets = 4
    G, V0, P0 = _wN_inputs(N, M, K)
    tr = na.NeuralAdmixture(K, epochs, batch, 2e-3, torch.device("cpu"), seed, world, rank == 0, None, None, None,
                            loss_mode="always", parallelism=parallelism)
    assert tr.batch_size == batch // world                   # neural_admixture.py:287
    Qs, Ps, model = tr.launch_training(torch.from_numpy(P0), torch.from_numpy(G), Hd, 8, torch.from_numpy(V0), M, N, None)
    if parallelism == "dp":                                  # message B in as many SNP ranges as M holds (4 asked for; 2048 SNPs each at least)
        assert tr.engine.lay.n_buckets == min(na.NeuralAdmixture.dp_buckets, (M + 2047) // 2048)
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].numpy(),
                 losses=np.asarray([tr.epoch_losses[e] for e in range(epochs)]))
    else:
        assert Qs == [] and Ps == []
    dist.barrier()
    dist.destroy_process_group()


def _wN_inputs(N, M, K):
    from oracle import nadm_oracle as O
    G = O.synth_genotypes(N, M, K, seed=4242, missing=0.02)
    rng = np.random.default_rng(17)
    V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.05, 0.95, size=(K, M)).astype(np.float32)
    return G, V0, P0


@pytest.mark.parametrize("world,parallelism,M", [(4, "dp", 512), (8, "dp", 512), (8, "snp", 512), (4, "dp", 6200)])
def test_global_batch_800_over_4_and_8_ranks(tmp_path, world, parallelism, M):
    """The reference's multi-GPU shape: --batch_size 800 over W GPUs = 800 // W rows per rank and step (200 / 100,
    neural_admixture.py:287), N = 1003 not a multiple of W (DistributedSampler wraps 1 / 5 indices, loaders.py:26-27), two
    steps per epoch with a ragged second one (51 / 26 rows per rank).  Product orchestration over gloo against the oracle's
    DDP emulation of the same run (oracle.train_run(world=W), itself pinned by ddp_w2 / ddp_w4 captured from the reference).
    M = 6200: message B = [small | V] travels in four SNP-range buckets (r05), each with slices and moments of its own."""
    from oracle import nadm_oracle as O
    N, K, Hd, batch, epochs, seed = 1003, 3, 32, 800, 2, 5
    port = 40500 + 1000 * (world == 8) + 500 * (parallelism == "snp") + 2000 * (M > 512) + (os.getpid() % 450)
    out = str(tmp_path / "wN.npz")
    mp.spawn(_wN_worker, args=(world, port, out, parallelism, N, M, K, Hd, batch, epochs, seed), nprocs=world, join=True)
    r = np.load(out)
    G, V0, P0 = _wN_inputs(N, M, K)
    p = O.make_params(seed, V0.copy(), P0.copy(), Hd, [K])
    p, Qs, losses = O.train_run(G, p, epochs, batch, 2e-3, seed, world=world)
    assert r["Q"].shape == (N, K)
    assert np.abs(r["Q"] - Qs[0]).max() < 1e-4
    assert np.abs(r["P"] - p.P[0]).max() < 1e-5
    assert np.abs(r["V"] - p.V).max() < 1e-4
    if parallelism == "dp":
        assert np.allclose(r["losses"], losses, rtol=1e-6)


def _transport_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_admixture_amd.comm import torch_comm
    c = torch_comm(rank, world)
    st = c.handle.contents
    sl = 12
    g = torch.arange(world * sl, dtype=torch.float32) * (rank + 1)          # "gradients": rank r holds (r + 1) * [0, 1, 2, ...]
    pbuf = torch.full((world * sl,), -1.0)
    pbuf[rank * sl:(rank + 1) * sl] = 100.0 * rank + torch.arange(sl, dtype=torch.float32)      # this rank's slice of the "parameters"
    small = torch.ones(8) * (rank + 1)
    c.transport.buffers += [g, pbuf, small]
    # the three collectives exactly as nadm_step calls them: raw pointers into registered buffers, in place
    assert st.reduce_scatter(None, g.data_ptr(), sl, None) == 0
    assert st.all_gather(None, pbuf.data_ptr(), sl, None) == 0
    assert st.all_reduce(None, small.data_ptr() + 8, 4, None) == 0             # a sub-range: elements 2..5
    tot = world * (world + 1) / 2
    assert torch.equal(g[rank * sl:(rank + 1) * sl], torch.arange(rank * sl, (rank + 1) * sl, dtype=torch.float32) * tot)     # own slice = the sum
    for r in range(world):
        assert torch.equal(pbuf[r * sl:(r + 1) * sl], 100.0 * r + torch.arange(sl, dtype=torch.float32))
    assert torch.equal(small, torch.tensor([rank + 1.0] * 2 + [tot] * 4 + [rank + 1.0] * 2))
    # a pointer outside every registered buffer is refused with a status, not an exception across the C frame
    assert st.all_reduce(None, torch.zeros(4).data_ptr(), 4, None) == 1 and c.transport.error is not None
    if rank == 0:
        open(out_path, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_torch_distributed_transport_of_the_step_collectives(tmp_path, world):
    """comm.torch_comm: the nadm_comm_t whose three function pointers call back into a torch.distributed group (the transport of the
    world-2 GPU tests that share one device, and the fallback when the library's RCCL communicator is unavailable) -- called here the
    way nadm_step calls it, on CPU tensors over gloo."""
    port = 43500 + (os.getpid() % 1500) + world
    out = str(tmp_path / "transport.txt")
    mp.spawn(_transport_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
