"""SNP-sharded ("tensor-parallel over M") training -- SURVEY.md section 8(f)-4.

Every rank owns a contiguous range of SNPs: those columns of the packed genotype matrix for ALL samples, the matching rows
of V and of every P_h, and their Adam state.  The small MLP is replicated.  A step processes the GLOBAL batch on every
rank's slice, so the only data exchanged per step are two small all-reduces -- the partial Z [B, C] before RMSNorm and
the partial dQ [B, sum K] before the MLP backward -- instead of the 4*M*(C+S)-byte gradient exchange of the
sample-sharded step (32 MB at M=500k, K=8).  Mathematically it is the single-device step on the global batch with
the gradient scaled by 1/world, which is what the reference's DDP mean over per-rank sum-losses computes
(neural_admixture.py:287,315-319), up to summation order.

The step is the same C call as everywhere (nadm_step, mode NADM_MODE_SNP: csrc/nadm_step.hip); this class slices data and
parameters and gathers results.  Sample sharding stays the default because it is what the reference does; this mode is
selected with ``parallelism="snp"`` (train(), NeuralAdmixture, CLI ``--parallelism snp``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .engine import Engine


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
    """Engine over this rank's SNP slice.  ``comm`` (comm.py) carries the two all-reduces of a step; ``group`` is the
    torch.distributed process group the result gathers and the loss read-back use (None = default)."""

    def __init__(self, M_total: int, C_: int, Hd: int, ks: Sequence[int], device: torch.device, max_batch: int,
                 comm=None, group=None):
        self.M_total, self.group = int(M_total), group
        world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
        self.m0, self.m1 = snp_slices(self.M_total, world)[rank]
        if self.m1 <= self.m0:
            raise RuntimeError(f"SNP-sharded run: rank {rank} of {world} would own no SNPs (M = {M_total})")
        super().__init__(self.m1 - self.m0, C_, Hd, ks, device, max_batch, mode="snp", comm=comm)

    # ------------------------------------------------------------------ data / parameters: slice, then as the base class
    def pack_from_host(self, data_u8, rows=None, chunk_rows=None) -> None:
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

    def read_loss(self, reset: bool = True):
        """Global loss: every rank accumulated the BCE terms of its SNPs (+ the supervised term on rank 0).  Collective."""
        if self.world > 1:
            dist.all_reduce(self.loss_acc, op=dist.ReduceOp.SUM, group=self.group)
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
