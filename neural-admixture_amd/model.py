"""Host-side mirror of the reference's model classes for the hot path.

``Q_P``            -- parameter container with the reference's state_dict keys / config JSON
                     (neural_admixture.py:100-230) and an inference ``__call__``.
``NeuralAdmixture`` -- trainer with the reference's constructor and ``launch_training`` signature
                     (neural_admixture.py:232-392), driving :class:`Engine` instead of autograd.
"""
from __future__ import annotations

import json
import logging
import sys
from collections import OrderedDict
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from .engine import Engine

logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(message)s")
log = logging.getLogger(__name__)


def init_encoder_weights(seed: int, C: int, Hd: int, ks):
    """Same RNG stream as the reference's module construction under ``torch.manual_seed(seed)``
    (src/utils.py:107): Linear(C,Hd) then Linear(Hd,k) for ascending k (neural_admixture.py:138-141,29),
    default nn.Linear init; RMSNorm weight = 1.  Returns the flat `small` vector (nadm.h order)."""
    torch.manual_seed(seed)
    lin1 = torch.nn.Linear(C, Hd, bias=True)
    heads = [torch.nn.Linear(Hd, k, bias=True) for k in sorted(ks)]
    parts = [torch.ones(C), lin1.weight.detach().reshape(-1), lin1.bias.detach()]
    for h in heads:
        parts += [h.weight.detach().reshape(-1), h.bias.detach()]
    return torch.cat(parts).to(torch.float32).numpy().copy()


class Q_P:
    """Parameter container / inference facade.  Not an nn.Module: the math lives in the HIP kernels."""

    def __init__(self, hidden_size: int, num_features: int, V: Optional[torch.Tensor] = None, P: Optional[torch.Tensor] = None,
                 ks_list: List[int] = [], is_train: bool = True, engine: Optional[Engine] = None, full_params: Optional[dict] = None):
        self.hidden_size, self.num_features = int(hidden_size), int(num_features)
        self.ks_list = [int(k) for k in ks_list]
        self.engine = engine
        self.full_params = full_params       # SNP-sharded runs: {"V": [M,C], "P": [[M,k] ...]} gathered on the master
        self._pending = None
        if engine is None and V is not None:
            self._pending = {"V": V}

    # ---- reference-compatible export (keys measured in SURVEY.md section 5) ----
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        e, L = self.engine, self.engine.lay
        sm = e.small.detach().cpu()
        h = L.heads
        sd = OrderedDict()
        fp = self.full_params
        sd["V"] = (fp["V"] if fp else e.V()).detach().cpu().contiguous()
        sd["batch_norm.weight"] = sm[h.g_off: h.g_off + L.C].clone()
        sd["common_encoder.0.weight"] = sm[h.w1_off: h.w1_off + L.Hd * L.C].view(L.Hd, L.C).clone()
        sd["common_encoder.0.bias"] = sm[h.b1_off: h.b1_off + L.Hd].clone()
        for i, k in enumerate(L.ks):
            sd[f"multihead_encoder.heads.{i}.weight"] = sm[h.wk_off[i]: h.wk_off[i] + k * L.Hd].view(k, L.Hd).clone()
            sd[f"multihead_encoder.heads.{i}.bias"] = sm[h.bk_off[i]: h.bk_off[i] + k].clone()
        for i in range(len(L.ks)):
            sd[f"decoders.decoders.{i}.weight"] = (fp["P"][i] if fp else e.P(i)).detach().cpu().contiguous()
        return sd

    def load_state_dict(self, sd, device: Optional[torch.device] = None, max_batch: int = 1024):
        """Build an inference engine from a reference-format state dict (no decoders needed)."""
        V = sd["V"].float()
        M, C = V.shape
        dev = device or torch.device("cuda:0")
        self.engine = Engine(M, C, self.hidden_size, self.ks_list, dev, max_batch)
        parts = [sd["batch_norm.weight"], sd["common_encoder.0.weight"].reshape(-1), sd["common_encoder.0.bias"]]
        for i in range(len(self.ks_list)):
            parts += [sd[f"multihead_encoder.heads.{i}.weight"].reshape(-1), sd[f"multihead_encoder.heads.{i}.bias"]]
        small = torch.cat([p.float().cpu() for p in parts]).numpy()
        P = np.zeros((sum(self.ks_list), M), dtype=np.float32)
        for i, k in enumerate(self.ks_list):
            key = f"decoders.decoders.{i}.weight"
            if key in sd:
                o = sum(self.ks_list[:i])
                P[o:o + k] = sd[key].float().cpu().numpy().T
        self.engine.load_params(V.cpu().numpy(), P, small)
        return self

    def to(self, device):
        return self

    def eval(self):
        return self

    def __call__(self, x: torch.Tensor):
        """x uint8 [b,M] (CPU or GPU) -> (list of Q_k [b,k], None).  Mirrors inference-mode forward
        (neural_admixture.py:157-177 with _return_infer)."""
        e = self.engine
        b = x.shape[0]
        e.pack_from_host(x.cpu().contiguous())
        idx = torch.arange(b, dtype=torch.int32, device=e.device)
        return e.infer_q(idx, b), None

    def save_config(self, name: str, save_dir: str) -> None:
        cfg = {"ks": self.ks_list, "num_features": self.num_features, "hidden_size": self.hidden_size, "activation": "relu"}
        with open(Path(save_dir) / f"{name}_config.json", "w") as fb:
            json.dump(cfg, fb)
        log.info("    Configuration file saved.")


def hudsons_fst(p1: torch.Tensor, p2: torch.Tensor) -> float:
    """neural_admixture.py:532-553."""
    num = torch.mean((p1 - p2) ** 2)
    den = torch.mean(p1 * (1 - p2) + p2 * (1 - p1)) + 1e-7
    return (num / den).item()


def fst_table(P: torch.Tensor) -> torch.Tensor:
    """Hudson's Fst of every pair of columns of P [M, k] at once: with G = P^T P and s = the column sums,
    mean((p_l - p_j)^2) = (G_ll + G_jj - 2 G_lj) / M and mean(p_l (1 - p_j) + p_j (1 - p_l)) = (s_l + s_j - 2 G_lj) / M -- one
    float64 product instead of k (k - 1) / 2 passes with a host read each (165 passes over 600k SNPs for heads K = 2..10)."""
    from .svd import gram64                                  # (elementwise: no first BLAS call of the process for a k x k product)
    Pd = P.detach()
    M = Pd.shape[0]
    G = gram64(Pd)
    sq, s1 = torch.diagonal(G), Pd.to(torch.float64).sum(dim=0)
    num = (sq[:, None] + sq[None, :] - 2 * G) / M
    den = (s1[:, None] + s1[None, :] - 2 * G) / M + 1e-7
    return (num / den).cpu()


def epoch_order(generator: torch.Generator, n: int) -> torch.Tensor:
    """The sample order ``iter(RandomSampler(range(n), generator=generator))`` yields for one epoch, as an int32 tensor,
    leaving ``generator`` in the state the sampler leaves it in: torch's sampler draws ``randperm(n)`` for the epoch and
    then a SECOND ``randperm(n)`` for the empty ``[:num_samples % n]`` tail (utils/data/sampler.py, RandomSampler.__iter__),
    so every epoch advances the stream by two draws.  Iterating the sampler object itself costs 60-120 ms per epoch at
    n = 100k (one Python ``yield`` per sample) -- more than the 57 ms the GPU needs for that epoch."""
    perm = torch.randperm(n, generator=generator)
    torch.randperm(n, generator=generator)
    return perm.to(torch.int32)


class _EpochOrders:
    """Sample orders of successive epochs on the device, one epoch ahead of the kernels.  Two things cost time at every epoch
    boundary when the order is made with ``epoch_order(...).to(device)`` (measured at N = 100k, 43 ms of kernels per epoch):
    the copy from pageable memory makes the host wait for every step queued before it, so the launch queue runs dry while
    the next order is drawn (two ``randperm`` = 10 ms of host time); and the int64 -> int32 conversion of 100k elements on
    the host is a parallel region of torch's intra-op pool -- its (here 128) threads spin at the region's barrier for
    milliseconds and the HIP runtime's own threads stall behind them: +5 ms per epoch even when the result is not used.
    Here epoch e+1's order is drawn as soon as epoch e's steps are queued (before the epoch's loss is read back), straight
    into pinned memory (``randperm(out=...)`` is serial), and the copy and the narrowing to int32 are queued on the compute
    stream behind those steps -- stream order is all the synchronisation the device buffers need (a copy on a side stream
    with events in both directions measured 3-4 ms per epoch slower: the launch queue then drains in bursts).  The
    generator is consumed in the same sequence as before, so the orders are the same.  Keep parallel CPU ops out of the
    epoch loop."""
    RING = 4                                             # pinned staging buffers; the host runs at most ~2.5 epochs ahead

    def __init__(self, generator: torch.Generator, n: int, device: torch.device):
        self.gen, self.n, self.dev = generator, n, device
        self.on_gpu = device.type == "cuda"
        if self.on_gpu:
            self.pin = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(self.RING)]
            self.copied = [None] * self.RING             # event: pin[i] has been read by its copy
            self.second = torch.empty(n, dtype=torch.int64)          # the sampler's second, unused draw
            self.wide = torch.empty(n, dtype=torch.int64, device=device)
            self.buf = [torch.empty(n, dtype=torch.int32, device=device) for _ in range(2)]
        self.ready = None
        self._after = None
        self._draw(0)

    def _draw(self, epoch: int) -> None:
        if not self.on_gpu:
            self.ready = epoch_order(self.gen, self.n).to(self.dev)
            return
        i = epoch % self.RING
        if self.copied[i] is not None:
            self.copied[i].synchronize()                 # RING epochs later: long since complete
        torch.randperm(self.n, generator=self.gen, out=self.pin[i])          # same two draws as epoch_order()
        torch.randperm(self.n, generator=self.gen, out=self.second)
        with torch.cuda.device(self.dev):                # the event below is recorded on this device's current stream
            self.wide.copy_(self.pin[i], non_blocking=True)
            self.copied[i] = torch.cuda.Event()
            self.copied[i].record()
            self.buf[epoch & 1].copy_(self.wide)         # int64 -> int32 on the device
        self.ready = self.buf[epoch & 1]

    def take(self, epoch: int, prefetch: bool) -> torch.Tensor:
        """Order of ``epoch`` (valid until the epoch after next is drawn)."""
        self._after = (epoch, prefetch)
        return self.ready

    def epoch_queued(self) -> None:
        """Call when every step of the epoch handed out last has been queued: draws the next order behind them."""
        epoch, prefetch = self._after
        if prefetch:
            self._draw(epoch + 1)


class NeuralAdmixture:
    """Trainer mirror (constructor signature of neural_admixture.py:248-249)."""
    engine_cls = Engine          # tests swap in an oracle-backed double to run the multi-rank logic on gloo / CPU
    engine_snp_cls = None        # set below (import cycle): snp_parallel.SnpShardedEngine
    # sample-sharded runs (csrc/nadm_step.hip): SNP ranges message B = [small | V] travels in (DDP's buckets, neural_admixture.py:315-319;
    # 1 = one message on the compute stream: on a rank's own timeline every further bucket costs more than the ~40 us of pass 3 /
    # pass 1 it can hide wire time under, profiles/r05_rank_emulation.txt -- a switch for the first multi-GPU node, like the next one),
    # and whether message A = [all P] gets a communicator of its own (A and B then share the links instead of queueing)
    dp_buckets = 1
    dp_second_comm = False

    def __init__(self, k, epochs, batch_size, learning_rate, device, seed, num_gpus, master, pack2bit=None,
                 min_k=None, max_k=None, supervised_loss_weight=100, loss_mode: str = "logged", parallelism: str = "dp"):
        self.k, self.min_k, self.max_k = k, min_k, max_k
        self.ks_list = [int(k)] if k is not None else list(range(int(min_k), int(max_k) + 1))
        self.num_gpus, self.device, self.master, self.seed = num_gpus, device, master, int(seed)
        self.epochs = int(epochs)
        self.global_batch = int(batch_size)
        self.batch_size = int(batch_size) // num_gpus if num_gpus > 0 else int(batch_size)   # :287
        self.lr = float(learning_rate)
        self.supervised_loss_weight = float(supervised_loss_weight)
        if parallelism not in ("dp", "snp"):
            raise ValueError("parallelism must be 'dp' (samples sharded, gradients summed over ranks) or 'snp' (SNPs sharded)")
        self.parallelism = parallelism
        self.loss_mode = loss_mode       # "logged": loss only on epochs that print it (:416); "always": every step; "steps": every step
        self.epoch_losses: dict = {}     # AND read back after each one like the reference's loss.item() (:414) -> step_losses (parity tests)
        self.step_losses: list = []
        self.logliks: Optional[list] = None   # filled by the SNP-sharded run (computed where the SNP slices live)

    # ---- batch order (src/loaders.py:8-35; sampler objects are torch's own) ----
    def _world(self):
        import torch.distributed as dist
        if self.num_gpus > 1 and dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
        return 1, 0

    def launch_training(self, P, data, hidden_size, num_features, V, M, N, pops=None):
        """P torch [sum(ks), M]; data uint8 CPU [N,M] (unpacked; this method packs the rank's rows);
        V torch [M,C].  Returns (Qs, Ps, model) like neural_admixture.py:392,530 (numpy lists on master)."""
        from torch.utils.data.distributed import DistributedSampler
        world, rank = self._world()
        dev = self.device
        C = int(num_features)
        infer_b = min(N, 1024)
        if self.parallelism == "snp" and world > 1:
            return self._launch_training_snp(P, data, hidden_size, C, V, M, N, pops, world, rank)
        if world > 1:
            # the step's collectives: RCCL over xGMI behind an nccl process group, torch.distributed callbacks otherwise (comm.py)
            from .comm import make_comm
            eng = self.engine_cls(M, C, hidden_size, self.ks_list, dev, max(self.batch_size, infer_b), mode="dp",
                                  comm=make_comm(dev, rank, world), n_buckets=self.dp_buckets,
                                  comm_a=make_comm(dev, rank, world) if self.dp_second_comm else None)
        else:
            eng = self.engine_cls(M, C, hidden_size, self.ks_list, dev, max(self.batch_size, infer_b))
        self.engine = eng
        small = init_encoder_weights(self.seed, C, hidden_size, self.ks_list)
        eng.load_params(V.detach().cpu().numpy(), P.detach().cpu().numpy(), small)

        if world > 1:
            shard = np.asarray(list(iter(DistributedSampler(range(N), num_replicas=world, rank=rank, shuffle=True, seed=self.seed))),
                               dtype=np.int64)            # set_epoch never called (loaders.py:27): fixed shard
            eng.pack_from_host(data, rows=shard)
            n_local = len(shard)
        else:
            shard = None
            eng.pack_from_host(data)
            n_local = N
            generator = torch.Generator().manual_seed(self.seed)     # neural_admixture.py:283
        if pops is not None:                               # supervised mode (:352-356, :434-474): labels follow the rows
            y = torch.as_tensor(pops).detach().cpu().numpy().astype(np.int64)
            eng.set_labels(y if shard is None else y[shard], self.ks_list[0], self.supervised_loss_weight)
        log_every = 5 if pops is None else 2               # :416 vs :457

        if self.master:
            log.info("")
            log.info("    Starting training...")
            log.info("")
        b = self.batch_size
        seq = torch.arange(n_local, dtype=torch.int32, device=dev)
        orders = _EpochOrders(generator, N, dev) if world == 1 else None
        host_threads = torch.get_num_threads()
        torch.set_num_threads(1)                           # the loop is launches only: no parallel regions (see _EpochOrders)
        try:
            for epoch in range(self.epochs):
                logged = (epoch % log_every == 0)
                with_loss = logged or self.loss_mode in ("always", "steps")
                if world > 1:
                    order = seq                                # rows are stored in shard order
                else:
                    order = orders.take(epoch, prefetch=epoch + 1 < self.epochs)
                for s in range(0, n_local, b):
                    bb = min(b, n_local - s)
                    eng.train_step(order[s:s + bb], bb, self.lr, with_loss)     # ONE C call, collectives included
                    if self.loss_mode == "steps":
                        self.step_losses.append(eng.read_loss(reset=False)[1])
                if orders is not None:
                    orders.epoch_queued()                      # next epoch's order: drawn and copied underneath this epoch's steps
                if with_loss:
                    loss_acc, _ = eng.read_loss(reset=True)
                    self.epoch_losses[epoch] = loss_acc
                    if logged and self.master:
                        log.info(f"            Loss in epoch {epoch:3d} on device {dev} is {loss_acc:,.0f}")
        finally:
            torch.set_num_threads(host_threads)
        eng.sync()                                         # what the last step left to "the next step" (engine.Engine.sync)
        # ---- final Q: sequential batches of <=1024, encoder only (:369-383) ----
        Qloc = [[] for _ in self.ks_list]
        for s in range(0, n_local, infer_b):
            bb = min(infer_b, n_local - s)
            for h, q in enumerate(eng.infer_q(seq[s:s + bb], bb)):
                Qloc[h].append(q)
        Qloc = [torch.cat(q, dim=0) for q in Qloc]
        if world > 1:
            import torch.distributed as dist
            Qs = []
            for h, q in enumerate(Qloc):
                bufs = [torch.empty_like(q) for _ in range(world)]
                dist.all_gather(bufs, q)
                shards = [torch.empty(n_local, dtype=torch.int64, device=dev) for _ in range(world)]
                dist.all_gather(shards, torch.as_tensor(shard, device=dev))
                full = torch.empty((N, q.shape[1]), dtype=q.dtype, device=dev)
                for r in range(world):
                    full[shards[r]] = bufs[r]
                Qs.append(full)
        else:
            Qs = Qloc
        if self.master:
            log.info("")
            log.info("    Training finished!")
            log.info("")
        self.raw_model = Q_P(hidden_size, C, ks_list=self.ks_list, engine=eng)
        self.display_divergences()
        if self.master:
            Ps = [eng.P(h).detach().cpu().numpy().copy() for h in range(len(self.ks_list))]
            Qs = [q.cpu().numpy() for q in Qs]
        else:
            Ps, Qs = [], []
        return Qs, Ps, self.raw_model

    def _launch_training_snp(self, P, data, hidden_size, C, V, M, N, pops, world, rank):
        """SNP-sharded run (snp_parallel.SnpShardedEngine): every rank holds all samples of its SNP slice and processes
        the GLOBAL batch -- the union of the per-rank batches the reference's DistributedSampler would hand out
        (src/loaders.py:25-27), in rank order -- so the trajectory is the sample-sharded one up to summation order."""
        import torch.distributed as dist
        from torch.utils.data.distributed import DistributedSampler
        dev = self.device
        infer_b = min(N, 1024)
        b_local = self.batch_size                                  # batch_size // num_gpus (:287)
        from .comm import make_comm
        eng = self.engine_snp_cls(M, C, hidden_size, self.ks_list, dev, max(b_local * world, infer_b), comm=make_comm(dev, rank, world))
        self.engine = eng
        small = init_encoder_weights(self.seed, C, hidden_size, self.ks_list)
        eng.load_params(V.detach().cpu().numpy(), P.detach().cpu().numpy(), small)
        eng.pack_from_host(data)
        shards = [np.asarray(list(iter(DistributedSampler(range(N), num_replicas=world, rank=r, shuffle=True, seed=self.seed))), dtype=np.int32)
                  for r in range(world)]
        n_local = len(shards[0])
        steps = []
        for s in range(0, n_local, b_local):
            steps.append(torch.as_tensor(np.concatenate([sh[s:s + b_local] for sh in shards])).to(dev))
        if pops is not None:
            eng.set_labels(torch.as_tensor(pops).detach().cpu().numpy().astype(np.int64), self.ks_list[0], self.supervised_loss_weight)
        log_every = 5 if pops is None else 2
        if self.master:
            log.info("")
            log.info("    Starting training (SNP-sharded)...")
            log.info("")
        for epoch in range(self.epochs):
            logged = (epoch % log_every == 0)
            with_loss = logged or self.loss_mode == "always"
            for idx in steps:                                       # set_epoch is never called: the same batches every epoch
                eng.train_step(idx, int(idx.numel()), self.lr, with_loss)
            if with_loss:
                loss_acc, _ = eng.read_loss(reset=True)             # collective: every rank calls it
                self.epoch_losses[epoch] = loss_acc
                if logged and self.master:                          # the global batch's loss (the reference's master prints its shard's)
                    log.info(f"            Loss in epoch {epoch:3d} on device {dev} is {loss_acc:,.0f}")
        seq = torch.arange(N, dtype=torch.int32, device=dev)
        Qloc = [[] for _ in self.ks_list]
        for s in range(0, N, infer_b):
            bb = min(infer_b, N - s)
            for h, q in enumerate(eng.infer_q(seq[s:s + bb], bb)):
                Qloc[h].append(q)
        Qs = [torch.cat(q, dim=0) for q in Qloc]                    # replicated: every rank has every row
        # log-likelihood report (train.py:134-146): every rank sums over its SNPs, one all-reduce of the doubles
        ll = torch.zeros(len(self.ks_list), dtype=torch.float64, device=dev)
        if dev.type == "cuda" and max(self.ks_list) <= 16:
            from .report import loglikelihood_hip
            for h in range(len(self.ks_list)):
                ll[h] = loglikelihood_hip(eng.xp, eng.M, eng.P(h).detach().cpu().numpy(), Qs[h].cpu().numpy())
            dist.all_reduce(ll)
            self.logliks = [float(v) for v in ll.cpu()]
        Pfull = [eng.gather_rows(eng.P(h)) for h in range(len(self.ks_list))]
        Vfull = eng.gather_rows(eng.V())
        if self.master:
            log.info("")
            log.info("    Training finished!")
            log.info("")
        self.raw_model = Q_P(hidden_size, C, ks_list=self.ks_list, engine=eng,
                             full_params={"V": Vfull, "P": Pfull} if self.master else None)
        self.display_divergences(Pfull if self.master else None)
        if self.master:
            return [q.cpu().numpy() for q in Qs], [p_.detach().cpu().numpy().copy() for p_ in Pfull], self.raw_model
        return [], [], self.raw_model

    def display_divergences(self, P_full=None) -> None:
        """Pairwise Hudson Fst between estimated populations (neural_admixture.py:476-509)."""
        if not self.master:
            return
        for i, k in enumerate(self.ks_list):
            dec = P_full[i] if P_full is not None else self.engine.P(i)
            fst = fst_table(dec)
            header = '\t'.join([f'Pop{p}' for p in range(k - 1)])
            log.info("    Results:")
            log.info(f'\n            Fst divergences between estimated populations: (K = {k})')
            log.info("")
            log.info(f'                \t{header}')
            log.info('            Pop0')
            for j in range(1, k):
                out = f'            Pop{j}'
                for l in range(j):
                    out += f"\t{float(fst[l, j]):0.3f}"
                log.info(out)
            log.info("\n")


from .snp_parallel import SnpShardedEngine as _SnpShardedEngine  # noqa: E402

NeuralAdmixture.engine_snp_cls = _SnpShardedEngine
