"""SNP-sharded ("tensor-parallel over M") training -- SURVEY.md section 8(f)-4.

Every rank owns a contiguous range of SNPs: those columns of the packed genotype matrix for ALL samples, the matching rows
of V and of every P_h, and their Adam state.  The small MLP is replicated.  A step processes the GLOBAL batch on every
rank's slice, so the only data exchanged per step are two small all-reduces -- the partial Z [B, C] before RMSNorm and
the partial dQ [B, sum K] before the MLP backward -- instead of the 4*M*(C+S)-byte gradient all-reduce of the
sample-sharded (DDP) step (32 MB at M=500k, K=8).  Mathematically it is the single-device step on the global batch with
the gradient scaled by 1/world, which is what the reference's DDP mean over per-rank sum-losses computes
(neural_admixture.py:287,315-319), up to summation order.

Sample sharding (engine.Engine.train_step_ddp) stays the default because it is what the reference does; this mode is
selected with ``parallelism="snp"`` (train(), NeuralAdmixture, CLI ``--parallelism snp``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from ._lib import lib, check, ptr
from .engine import Engine, _stream
from .layout import ModelLayout


def snp_slices(M: int, world: int, align: Optional[int] = None):
    """[m0, m1) per rank: contiguous, sizes as equal as the alignment allows.  Boundaries sit at multiples of ``align`` SNPs
    -- 1024 (256 packed bytes) by default, 64 or 4 (one byte) for matrices too small for that; a slice is copied into its
    own packed matrix, so a byte boundary is all that is required."""
    if align is None:
        align = next(a for a in (1024, 64, 4) if a == 4 or M >= 4 * a * world)
    units = (M + align - 1) // align
    cuts = [min(M, (units * r // world) * align) for r in range(world)] + [M]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class SnpShardedEngine(Engine):
    """Engine over this rank's SNP slice.  ``group`` is the torch.distributed process group (None = default)."""

    def __init__(self, M_total: int, C_: int, Hd: int, ks: Sequence[int], device: torch.device, max_batch: int,
                 rank: int, world: int, group=None):
        self.M_total, self.rank, self.world, self.group = int(M_total), int(rank), int(world), group
        self.m0, self.m1 = snp_slices(self.M_total, self.world)[self.rank]
        if self.m1 <= self.m0:
            raise RuntimeError(f"SNP-sharded run: rank {rank} of {world} would own no SNPs (M = {M_total})")
        super().__init__(self.m1 - self.m0, C_, Hd, ks, device, max_batch)
        L = self.lay
        self._zsum = torch.zeros(self.bmax * L.CP, dtype=torch.float32, device=device)
        self._dqsum = torch.zeros(self.bmax * L.SP, dtype=torch.float32, device=device)

    # ------------------------------------------------------------------ data / parameters: slice, then as the base class
    def pack_from_host(self, data_u8, rows=None, chunk_rows: int = 8192) -> None:
        if rows is not None:
            raise RuntimeError("SNP-sharded engines hold every sample; rows= is for the sample-sharded mode")
        if hasattr(data_u8, "packed"):                      # io.PackedGenotypes: cut the byte columns of the slice
            from .io import PackedGenotypes
            b0, nb = self.m0 // 4, (self.m1 - self.m0 + 3) // 4
            loc = torch.zeros((data_u8.N, self.ld), dtype=torch.uint8)
            loc[:, :nb] = data_u8.packed[:, b0:b0 + nb]
            if (self.m1 - self.m0) % 4:                      # the slice ends inside a byte only at the very end of the matrix
                assert self.m1 == self.M_total
            super().pack_from_host(PackedGenotypes(loc, data_u8.N, self.m1 - self.m0), None, chunk_rows)
        else:
            super().pack_from_host(data_u8[:, self.m0:self.m1].contiguous(), None, chunk_rows)
        self.rows_are_sharded = False

    def load_params(self, V_MC: np.ndarray, P_SM: np.ndarray, small: np.ndarray) -> None:
        super().load_params(np.ascontiguousarray(V_MC[self.m0:self.m1]), np.ascontiguousarray(P_SM[:, self.m0:self.m1]), small)

    # ------------------------------------------------------------------ step
    def sum_rows(self, src: torch.Tensor, rows: int, n: int, out: torch.Tensor) -> None:
        """out[:n] = sum of the ``rows`` rows of length n at the start of src (nadm_sum_rows: fixed order)."""
        check(lib.nadm_sum_rows(ptr(src), rows, n, ptr(out), _stream()), "sum_rows")

    def _all_reduce(self, t: torch.Tensor) -> None:
        if self.world > 1 or dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def forward(self, idx: torch.Tensor, b: int) -> None:
        """Partial Z over this rank's SNPs -> sum over ranks -> replicated MLP forward."""
        L = self.lay
        self.encode_partial(idx, b)
        zs = self._zsum[: b * L.CP]
        self.sum_rows(self.zpart, L.enc_chunks, b * L.CP, zs)
        self._all_reduce(zs)
        self.mlp_forward(b, zs, 1)

    def backward(self, idx: torch.Tensor, b: int, with_loss: bool = True, fused_adam=None, **_unused) -> None:
        """Pass 2 on the slice (dP is final and local), partial dQ -> sum over ranks -> replicated MLP backward (dZ and the
        small gradients come out identical on every rank), pass 3 on the slice (dV final and local)."""
        L = self.lay
        fa = {} if fused_adam is None else {"fused_adam": fused_adam}
        n_loss = self.decode_all(idx, b, with_loss, supervised=(self.rank == 0), **fa)   # the supervised term must enter the sum once
        dq_offs, _ = L.dq_offsets(b)
        dqs = self._dqsum[: b * L.SP]
        o = 0
        for h, kp in enumerate(L.kp):                          # per head: [chunks_h, b*kp] -> [b*kp], blocks laid back to back
            ch = L.dec_chunks[h]
            self.sum_rows(self.dqpart[dq_offs[h]:], ch, b * kp, dqs[o:])
            o += b * kp
        self._all_reduce(dqs)
        side = fused_adam is not None                         # fused step: weight-gradient partials ride on pass 3's launch
        self.mlp_backward(b, n_loss if with_loss else 0, dq_src=dqs, dq_M=1, weights=not side)
        self.encode_backward(idx, b, **fa, **({"side_weights": True} if side else {}))

    def train_step(self, idx: torch.Tensor, b: int, lr: float, with_loss: bool = True) -> None:
        """One step on the global batch idx; the 1/world gradient scale reproduces DDP's mean over ranks."""
        self.forward(idx, b)
        if self.fused_adam:          # dP and dV of the slice are final and local: Adam in the epilogues of passes 2 and 3
            self.step_count += 1
            self.backward(idx, b, with_loss, fused_adam=(lr, 1.0 / self.world))    # small parameters: nadm_small_grads
            self.p_unit = True                                # restrict_P ran in pass 2's epilogue
            return
        self.backward(idx, b, with_loss)
        self.adam(lr, 1.0 / self.world)

    def train_step_ddp(self, *a, **k):
        raise RuntimeError("SnpShardedEngine: use train_step (the step itself contains the collectives)")

    def read_loss(self, reset: bool = True):
        """Global loss: every rank accumulated the BCE terms of its SNPs (+ the supervised term on rank 0)."""
        self._all_reduce(self.loss_acc)
        return super().read_loss(reset)

    # ------------------------------------------------------------------ results
    def gather_rows(self, local_MK: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
        """Concatenate every rank's [M_local, k] block along SNPs on rank ``dst`` (None elsewhere)."""
        if self.world == 1:
            return local_MK.contiguous()
        sl = snp_slices(self.M_total, self.world)
        k = local_MK.shape[1]
        bufs = [torch.empty((m1 - m0, k), dtype=local_MK.dtype, device=local_MK.device) for m0, m1 in sl]
        if all(m1 - m0 == sl[0][1] - sl[0][0] for m0, m1 in sl):
            dist.all_gather(bufs, local_MK.contiguous(), group=self.group)
        else:
            self._all_gather_uneven(bufs, local_MK.contiguous())
        return torch.cat(bufs, dim=0) if self.rank == dst else None

    def _all_gather_uneven(self, bufs: List[torch.Tensor], mine: torch.Tensor) -> None:
        for r, buf in enumerate(bufs):                         # slices differ by at most one alignment unit: broadcast each
            if r == self.rank:
                buf.copy_(mine)
            dist.broadcast(buf, src=r, group=self.group)
