"""The sample- and SNP-sharded steps over REAL RCCL (one process per GPU, nadm_comm_rccl: ncclCommInitRank / ReduceScatter /
AllGather / AllReduce over xGMI) against the DDP emulation captured from the reference (tests/golden/ddp_w2.npz, ddp_w4.npz).

Needs a node with at least two GPUs; the build's test boxes have one, where these tests SKIP and the same logic is covered with
two processes sharing the GPU over gloo callbacks (tests/test_gpu_parity.py::test_world2_real_engine_on_one_gpu_...) and with a
1-rank RCCL communicator (tests/test_soak_handoffs.py).  On the first multi-GPU node these are the tests to run before bench.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


def _worker(rank, world, port, out_path, parallelism, fixture):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import neural_admixture_amd as na_
    from oracle import nadm_oracle as O_
    d = np.load(os.path.join(GOLD, fixture))
    G = O_.unpack2bit(d["G_packed"], int(d["M"]))
    tr = na_.NeuralAdmixture(int(d["K"]), int(d["epochs"]), int(d["batch"]), float(d["lr"]), dev, int(d["seed"]),
                              world, rank == 0, None, None, None, loss_mode="always", parallelism=parallelism)
    Qs, Ps, model = tr.launch_training(torch.from_numpy(d["P0"]), torch.from_numpy(G), int(d["Hd"]), 8, torch.from_numpy(d["V0"]),
                                       int(d["M"]), int(d["N"]), None)
    assert tr.engine.comm.kind == "rccl" and tr.engine.comm.world == world            # the library's own communicator, not callbacks
    if parallelism == "dp":
        assert tr.engine.moments_sharded and tr.engine.mflat.numel() == tr.engine.lay.slice_b + tr.engine.lay.slice_a
    # every rank ends with the same parameters (all-gathered / replicated)
    ref = tr.engine.small.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, tr.engine.small)
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].cpu().numpy(),
                 losses=np.asarray([tr.epoch_losses[e] for e in range(int(d["epochs"]))]))
    else:
        assert Qs == [] and Ps == []
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,parallelism", [(2, "dp"), (2, "snp"), (4, "dp"), (4, "snp")])
def test_sharded_training_over_rccl_matches_the_reference_ddp_emulation(tmp_path, world, parallelism):
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    import torch.multiprocessing as mp
    fixture = "ddp_w2.npz" if world == 2 else "ddp_w4.npz"
    port = 41500 + (os.getpid() % 2000) + 3 * world + (1 if parallelism == "snp" else 0)
    out = str(tmp_path / f"rccl_w{world}_{parallelism}.npz")
    mp.spawn(_worker, args=(world, port, out, parallelism, fixture), nprocs=world, join=True)
    r = np.load(out)
    d = np.load(os.path.join(GOLD, fixture))
    assert np.abs(r["Q"] - d["Q"]).max() < 1e-4
    assert np.abs(r["P"] - d["P"]).max() < 1e-5
    assert np.abs(r["V"] - d["V"]).max() < 1e-4
    if parallelism == "dp":
        assert np.allclose(r["losses"], d["losses_rank0"].reshape(int(d["epochs"]), -1).sum(1), rtol=1e-5)


def _bucket_worker(rank, world, port, out_path, second_comm, N, M, K, Hd, batch, epochs, seed):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import neural_admixture_amd as na_
    from test_ddp_gloo import _wN_inputs
    na_.NeuralAdmixture.dp_buckets = 4
    na_.NeuralAdmixture.dp_second_comm = bool(second_comm)
    G, V0, P0 = _wN_inputs(N, M, K)
    tr = na_.NeuralAdmixture(K, epochs, batch, 2e-3, dev, seed, world, rank == 0, None, None, None, loss_mode="always", parallelism="dp")
    Qs, Ps, model = tr.launch_training(torch.from_numpy(P0), torch.from_numpy(G), Hd, 8, torch.from_numpy(V0), M, N, None)
    e = tr.engine
    assert e.comm.kind == "rccl" and e.lay.n_buckets == 4 and (e.comm_a is not None) == bool(second_comm)
    for t in (e.small, e.big):                               # every rank ends with the same parameters (all-gathered bucket by bucket)
        ref = t.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, t)
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].cpu().numpy(),
                 losses=np.asarray([tr.epoch_losses[ep] for ep in range(epochs)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,second_comm", [(2, False), (2, True), (4, False), (4, True), (8, False)])
def test_bucketed_message_b_over_rccl_matches_the_ddp_emulation(tmp_path, world, second_comm):
    """r05: message B = [small | V] in four SNP-range buckets pipelined against pass 3 and the next pass 1, message A optionally on a
    communicator of its own, over REAL RCCL -- the issue order on the communicator(s) from two side streams, the per-bucket slices and
    moments -- against the oracle's DDP emulation of the same run (oracle.train_run(world=W), pinned by ddp_w2 / ddp_w4 from the
    reference).  Global batch 800 over W ranks (neural_admixture.py:287), ragged last step, N not a multiple of W."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    import sys
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from oracle import nadm_oracle as O
    from test_ddp_gloo import _wN_inputs
    N, M, K, Hd, batch, epochs, seed = 1003, 9000, 3, 32, 800, 2, 5
    port = 44500 + (os.getpid() % 2000) + 3 * world + int(second_comm)
    out = str(tmp_path / f"bkt_w{world}.npz")
    mp.spawn(_bucket_worker, args=(world, port, out, second_comm, N, M, K, Hd, batch, epochs, seed), nprocs=world, join=True)
    r = np.load(out)
    G, V0, P0 = _wN_inputs(N, M, K)
    p = O.make_params(seed, V0.copy(), P0.copy(), Hd, [K])
    p, Qs, losses = O.train_run(G, p, epochs, batch, 2e-3, seed, world=world)
    assert np.abs(r["Q"] - Qs[0]).max() < 1e-4
    assert np.abs(r["P"] - p.P[0]).max() < 1e-5
    assert np.abs(r["V"] - p.V).max() < 1e-4
    assert np.allclose(r["losses"], losses, rtol=1e-5)
