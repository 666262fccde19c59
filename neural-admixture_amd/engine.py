"""Device-side state and step sequencing for the training hot path.

One Engine per process / GPU.  It owns (through torch) every HBM buffer and drives the C-ABI
kernels on torch's current stream.  Mirrors what ``NeuralAdmixture._run_epoch`` / ``_run_step`` do
per batch (neural_admixture.py:394-432): gather + decode, forward, loss, backward, (all-reduce),
Adam, restrict_P -- without materialising any [b, M] tensor.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import math
import os

import numpy as np
import torch

from ._lib import lib, check, ptr, AdamArgs, MlpWeights
from .layout import ModelLayout

_f32 = torch.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    # tests/ subclass this with the oracle behind the kernel-calling methods to exercise the
    # distributed orchestration on CPU/gloo; the product class itself refuses to run without a GPU.
    _CPU_TEST_DOUBLE = False

    def __init__(self, M: int, C_: int, Hd: int, ks: Sequence[int], device: torch.device, max_batch: int):
        if device.type != "cuda" and not self._CPU_TEST_DOUBLE:
            raise RuntimeError("neural_admixture_amd.Engine needs a ROCm GPU device (no CPU fallback)")
        self.device = device
        self.lay = L = ModelLayout(M, C_, Hd, ks)
        self.M, self.ld = L.M, ModelLayout.row_stride(L.M)
        self.bmax = int(max_batch)
        z = lambda n, dt=_f32: torch.zeros(int(n), dtype=dt, device=device)
        # parameters, gradients, Adam moments
        # `big`, `mbig`, `vbig` (and `small` & co. below) are PROPERTIES: the data-parallel step leaves the P update to the prologue of the
        # next pass 2 and the single-GPU step leaves the small-parameter update to the next pass 1; any access from outside the step
        # sequence first applies what is owed, so nobody ever reads parameters or moments one step behind.  The step methods
        # themselves use the underscore names.
        self._big, self._mbig, self._vbig = z(L.n_big), z(L.n_big), z(L.n_big)
        # The small parameters, their gradient and moments are PROPERTIES (below): the single-GPU step leaves the sum of the
        # weight-gradient partials + Adam on them to side blocks of the NEXT step's pass 1 (nadm_encode_fwd_small), and any other
        # access first applies a pending update as a launch of its own -- nobody ever sees them one step behind
        self._pending_small = None                            # (splits, lr, grad_scale, step) of the update still owed
        self.defer_small = True                               # False: nadm_small_grads right after pass 3 (test hook / A-B)
        self._small, self._msmall, self._vsmall = z(L.n_small), z(L.n_small), z(L.n_small)
        # gradients live in ONE flat buffer [small | pad | V | P...] so that the data-parallel step needs two all-reduces:
        # the P part right after pass 2, and small + V together after pass 3
        self._ns_pad = (L.n_small + 63) // 64 * 64
        self.gflat = z(self._ns_pad + L.n_big)
        self._gsmall, self.gbig = self.gflat[: L.n_small], self.gflat[self._ns_pad:]
        # activations / scratch
        b = self.bmax
        self.zpart = z(L.enc_chunks * b * L.CP)
        self.Z, self.rinv, self.Zn = z(b * L.CP), z(b), z(b * L.CP)
        self.H, self.Q = z(b * L.Hd), z(b * L.SP)
        self.dL, self.dHpre, self.dgp, self.dZ = z(b * L.SP), z(b * L.Hd), z(b * L.CP), z(b * L.CP)
        self.dqpart = z(L.dq_offsets(b)[1])
        self.losspart = z(L.n_loss + 1)                  # last slot: supervised term (nadm_supervised_ce)
        self.labels: Optional[torch.Tensor] = None      # int32 [rows], supervised mode only
        self.n_classes, self.sup_weight = 0, 0.0
        self.small_part = z(int(lib.nadm_sample_splits(b)) * L.n_small)
        # Q as the bf16 MFMA operand images of pass 2, written by the MLP forward (nadm_mlp_fwd_images; heads with padded K <= 16):
        # zero-filled once, one region per head.  _qimg_b = batch size of the images that are valid for self.Q right now
        self.q_images = any(kp <= 16 for kp in L.kp)          # False: every block of pass 2 splits Q itself (test hook)
        self._qimg_head = int(lib.nadm_q_image_bytes(b))
        self.qimg = torch.zeros(len(L.ks) * self._qimg_head, dtype=torch.uint8, device=device) if self.q_images else None
        self._qimg_b = -1
        # dZ as the FP6 operand image of pass 3 (C <= 8; nadm_dz_image): built once per step from dZ
        self._dzimg = (torch.zeros(int(lib.nadm_dz_image_bytes(b)), dtype=torch.uint8, device=device)
                       if L.CP <= 8 and not self._CPU_TEST_DOUBLE else None)
        self._dzcnt = torch.zeros((b + 31) // 32, dtype=torch.int32, device=device)      # group counters of nadm_mlp_bwd_image
        self._dzimg_b = -1                               # batch size the image is valid for (-1: rebuild from dZ)
        self.loss_acc = torch.zeros(2, dtype=torch.float64, device=device)
        self.xp: Optional[torch.Tensor] = None          # packed genotypes [rows, ld]
        self.step_count = 0
        self.p_unit = True                              # every P entry in [0, 1] (see load_params)
        # pass 2 writes the batch as a copy of its own into xg for pass 3 (include/nadm.h, nadm_decode_bce_gather): missing calls
        # already 0 and tiled by pass 3's chunks, so that a block of pass 3 streams one contiguous region instead of gathering
        # 128-byte row pieces out of the resident matrix -- those arrive at 2.6 TB/s whatever the size of the matrix (pass 3: 50 us
        # against 40 at M = 500k; and at 12.5 GB resident 13 % of them miss the translation cache, profiles/r01_pmc_tlb.txt).  The
        # copy costs pass 2 b * ld bytes of writes it has the bandwidth for.  None = on a GPU: always (r02 / early r03: only above
        # 4 GB resident, when the copy was only known to cure the translation misses); False: pass 3 gathers the rows itself.
        self.gather_batch: Optional[bool] = None
        self._xg: Optional[torch.Tensor] = None
        self._iota: Optional[torch.Tensor] = None
        self._xg_key = None
        self.side_weights = device.type == "cuda"            # MLP weight-gradient partials as extra blocks of pass 3's launch
        self.fused_adam = device.type == "cuda"               # Adam in the epilogues of passes 2 and 3, see train_step
        self.timers: Optional[dict] = None                    # per-kernel timing (bench.py): {name: [(start, end) HIP events on the launch stream]}
        self.timed_names = None                               # restrict the timers to these kernel names (None = all)
        self.head_streams = 2                                 # decode_all: concurrent pass-2 launches of a multi-head model
        self._head_streams, self._head_events = None, None
        self._pending_ddp = None                              # (works, lr, grad_scale, step) of a deferred P update
        self._pending_vs = None                               # (lr, grad_scale, step) of a deferred update of V and the small parameters

    GATHER_MIN_BYTES = 0

    def _gather(self) -> bool:
        if self.gather_batch is not None:
            return bool(self.gather_batch) and self.lay.CP <= 8
        return (self.device.type == "cuda" and self.xp is not None and self.xp.numel() >= self.GATHER_MIN_BYTES
                and self.lay.CP <= 8)                         # the copy is tiled for the matrix-core pass 3 (C <= 8)

    def _xg_buf(self) -> torch.Tensor:
        if self._xg is None:
            self._xg = torch.empty(int(lib.nadm_batch_copy_bytes(self.bmax, self.M)), dtype=torch.uint8, device=self.device)   # tiled by pass 3's chunks
            self._iota = torch.arange(self.bmax, dtype=torch.int32, device=self.device)
        return self._xg

    def _timed(self, name):
        if self.timers is None or (self.timed_names is not None and name not in self.timed_names):
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.timers.setdefault(name, []).append(ev)
        ev[0].record()
        return ev

    # ------------------------------------------------------------------ data
    def set_packed(self, xp: torch.Tensor) -> None:
        if xp.dtype != torch.uint8 or xp.dim() != 2 or xp.shape[1] != self.ld or not xp.is_contiguous():
            raise RuntimeError(f"packed genotypes must be contiguous uint8 [rows, {self.ld}]")
        self.xp = xp

    def pack_from_host(self, data_u8: torch.Tensor, rows: Optional[np.ndarray] = None, chunk_rows: int = 8192) -> None:
        """uint8 [N,M] CPU tensor -> packed rows in HBM.  Packs on the host (2 bits/genotype cross
        PCIe instead of 8; the reference ships unpacked bytes in 1024-row chunks with a blocking
        sync per chunk, pack2bit.cu:79-115).  ``rows``: optional row selection/order (rank shard)."""
        self.rows_are_sharded = rows is not None
        if hasattr(data_u8, "packed"):                   # io.PackedGenotypes: already in the kernel layout, just ship the rows
            if data_u8.M != self.M or data_u8.packed.shape[1] != self.ld:
                raise RuntimeError("packed genotypes do not match the engine's SNP count / row stride")
            src = data_u8.packed if rows is None else data_u8.packed[torch.as_tensor(rows, dtype=torch.long)]
            self.xp = src.contiguous().to(self.device)
            return
        if data_u8.dtype != torch.uint8 or data_u8.dim() != 2 or data_u8.device.type != "cpu":
            raise RuntimeError("pack_from_host expects a uint8 [N,M] CPU tensor")
        N, M = data_u8.shape
        if M != self.M:
            raise RuntimeError("pack_from_host: SNP count mismatch")
        n_out = N if rows is None else len(rows)
        xp = torch.empty((n_out, self.ld), dtype=torch.uint8, device=self.device)
        stage = torch.empty((min(chunk_rows, max(n_out, 1)), self.ld), dtype=torch.uint8).pin_memory() \
            if torch.cuda.is_available() else None
        for s in range(0, n_out, chunk_rows):
            e = min(n_out, s + chunk_rows)
            src = data_u8[s:e] if rows is None else data_u8[torch.as_tensor(rows[s:e], dtype=torch.long)]
            src = src.contiguous()
            check(lib.nadm_pack2bit_host(ptr(src), ptr(stage), e - s, M, self.ld), "pack2bit_host")
            xp[s:e].copy_(stage[: e - s], non_blocking=False)
        self.xp = xp

    def set_labels(self, labels, n_classes: int, weight: float = 100.0) -> None:
        """Supervised mode (neural_admixture.py:460-474): class index per RESIDENT row (same order as xp)."""
        lab = torch.as_tensor(np.asarray(labels), dtype=torch.int32)
        if lab.dim() != 1 or (self.xp is not None and lab.numel() != self.xp.shape[0]):
            raise RuntimeError("labels must be one class index per resident genotype row")
        if len(self.lay.ks) != 1 or int(n_classes) != self.lay.ks[0]:
            raise RuntimeError(f"supervised mode needs a single head with K == number of classes ({n_classes})")
        if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) >= int(n_classes)):
            raise RuntimeError("label out of range")
        self.labels, self.n_classes, self.sup_weight = lab.to(self.device), int(n_classes), float(weight)

    # ------------------------------------------------------------------ parameters
    def load_params(self, V_MC: np.ndarray, P_SM: np.ndarray, small: np.ndarray) -> None:
        """V_MC [M,C]; P_SM [sum(ks), M] (reference P_init layout, train.py:63,67); small = flat
        g|W1|b1|Wk|bk in the nadm.h order."""
        L = self.lay
        self._pending_small = self._pending_ddp = self._pending_vs = None     # whatever was owed is overwritten
        self._qimg_b = -1
        big = np.zeros(L.n_big, dtype=np.float32)
        big[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C] = V_MC
        ini = 0
        for h, k in enumerate(L.ks):
            big[L.p_off[h]: L.p_off[h] + L.M * L.kp[h]].reshape(L.M, L.kp[h])[:, :k] = P_SM[ini:ini + k].T
            ini += k
        self._big.copy_(torch.from_numpy(big))
        # the loss value of pass 2 may skip the clamp of the reconstruction while every P entry lies in [0, 1]; true after the
        # first restrict_P, and for the GMM initialisation (clipped to [5e-6, 1 - 5e-6]) -- not for the supervised one
        self.p_unit = bool(np.min(P_SM) >= 0.0 and np.max(P_SM) <= 1.0) if np.size(P_SM) else True
        self.small.copy_(torch.from_numpy(np.ascontiguousarray(small, dtype=np.float32)))
        for t in (self._mbig, self._vbig, self.msmall, self.vsmall, self.gbig, self.gsmall):
            t.zero_()
        self.step_count = 0

    def V(self) -> torch.Tensor:
        L = self.lay
        return self._big[: L.M * L.CP].view(L.M, L.CP)[:, : L.C]

    def P(self, h: int) -> torch.Tensor:
        L = self.lay
        return self._big[L.p_off[h]: L.p_off[h] + L.M * L.kp[h]].view(L.M, L.kp[h])[:, : L.ks[h]]

    def gV(self) -> torch.Tensor:
        L = self.lay
        return self.gbig[: L.M * L.CP].view(L.M, L.CP)[:, : L.C]

    def gP(self, h: int) -> torch.Tensor:
        L = self.lay
        return self.gbig[L.p_off[h]: L.p_off[h] + L.M * L.kp[h]].view(L.M, L.kp[h])[:, : L.ks[h]]

    def flush_small(self) -> None:
        """Apply the small-parameter update the last step left to the next pass 1 (no-op if none is owed): the single-GPU step's
        (sum of the weight-gradient partials + Adam), or the data-parallel step's, which comes with the update of V."""
        if self._pending_vs is not None:
            lr, scale, step = self._pending_vs
            self.adam_v_small(lr, scale, step)
            self._pending_vs = None
        if self._pending_small is None:
            return
        splits, lr, scale, step = self._pending_small
        sa = AdamArgs(self._msmall.data_ptr(), self._vsmall.data_ptr(), lr, step, scale, 0)
        check(lib.nadm_small_grads(ptr(self.small_part), splits, self.lay.n_small, ptr(self._gsmall), ptr(self._small), C.byref(sa), _stream()),
              "small_grads")
        self._pending_small = None                            # only now: a refused launch must not lose the update

    def _flushed(self, t):
        self.flush_small()
        return t

    def _p_flushed(self, t):
        self.finish_ddp()
        return t

    big = property(lambda self: self._p_flushed(self._big), lambda self, t: setattr(self, "_big", t))
    mbig = property(lambda self: self._p_flushed(self._mbig), lambda self, t: setattr(self, "_mbig", t))
    vbig = property(lambda self: self._p_flushed(self._vbig), lambda self, t: setattr(self, "_vbig", t))

    def _set_q(self, t):
        self._Q = t
        self._qimg_b = -1                                     # Q rewritten from outside: its operand images are stale

    # Q is written by mlp_forward() through the underscore name; an assignment from outside (tests, a caller that edits Q before
    # decode_all) drops the bf16 operand images of pass 2, which then splits Q itself.  In-place edits of the tensor: invalidate_q().
    Q = property(lambda self: self._Q, _set_q)

    def invalidate_q(self) -> None:
        """Call after editing Q (or P through a raw view) in place: pass 2 rebuilds its operands from the fp32 values."""
        self._qimg_b = -1
        self.p_unit = False                                   # P may hold anything: the loss path clamps until the next restrict_P

    small = property(lambda self: self._flushed(self._small), lambda self, t: setattr(self, "_small", t))
    msmall = property(lambda self: self._flushed(self._msmall), lambda self, t: setattr(self, "_msmall", t))
    vsmall = property(lambda self: self._flushed(self._vsmall), lambda self, t: setattr(self, "_vsmall", t))
    gsmall = property(lambda self: self._flushed(self._gsmall), lambda self, t: setattr(self, "_gsmall", t))

    # ------------------------------------------------------------------ kernels
    def encode_partial(self, idx: torch.Tensor, b: int) -> None:
        """Pass 1: per-chunk partial sums of Z = X.V for the batch rows idx (int32 [b], device) into zpart."""
        L, st = self.lay, _stream()
        if b > self.bmax:
            raise RuntimeError("batch larger than the engine was sized for")
        ev = self._timed("encode_fwd")
        if self._pending_vs is not None and L.CP <= 8:
            # data-parallel step: the previous step's update of V (prologue of this launch, from the all-reduced gradient in gbig)
            # and of the small parameters (side blocks; the all-reduced flat gradient is their one "split")
            lr, scale, step = self._pending_vs
            av = AdamArgs(self._mbig.data_ptr(), self._vbig.data_ptr(), lr, step, scale, 1)
            sa = AdamArgs(self._msmall.data_ptr(), self._vsmall.data_ptr(), lr, step, scale, 0)
            check(lib.nadm_encode_fwd_step(ptr(self.xp), self.ld, ptr(idx), b, L.M, ptr(self._big), L.CP, ptr(self.zpart), ptr(self.gbig),
                                           C.byref(av), ptr(self._gsmall), 1, L.n_small, ptr(self._gsmall), ptr(self._small), C.byref(sa), st),
                  "encode_fwd_step")
            self._pending_vs = None
        elif self._pending_small is not None and L.CP <= 8:   # the previous step's small update rides in this launch
            splits, lr, scale, step = self._pending_small
            sa = AdamArgs(self._msmall.data_ptr(), self._vsmall.data_ptr(), lr, step, scale, 0)
            check(lib.nadm_encode_fwd_small(ptr(self.xp), self.ld, ptr(idx), b, L.M, ptr(self._big), L.CP, ptr(self.zpart),
                                            ptr(self.small_part), splits, L.n_small, ptr(self._gsmall), ptr(self._small), C.byref(sa), st),
                  "encode_fwd_small")
            self._pending_small = None                        # only now: a refused launch must not lose the update
        else:
            self.flush_small()
            check(lib.nadm_encode_fwd(ptr(self.xp), self.ld, ptr(idx), b, L.M, ptr(self._big), L.CP, ptr(self.zpart), st), "encode_fwd")
        if ev: ev[1].record()

    def mlp_forward(self, b: int, z_src: Optional[torch.Tensor] = None, n_chunks: Optional[int] = None) -> None:
        """RMSNorm + MLP + per-head softmax from partial sums [n_chunks, b, CP] (default: this engine's zpart).  Fills Z,
        rinv, Zn, H, Q."""
        L, st = self.lay, _stream()
        args = (C.byref(L.heads), ptr(self.small), ptr(self.zpart if z_src is None else z_src),
                L.enc_chunks if n_chunks is None else n_chunks, b, ptr(self.Z), ptr(self.rinv), ptr(self.Zn), ptr(self.H), ptr(self.Q))
        if self.q_images and self.qimg is not None:
            check(lib.nadm_mlp_fwd_images(*args, ptr(self.qimg), self._qimg_head, st), "mlp_fwd_images")
            self._qimg_b = b
        else:
            check(lib.nadm_mlp_fwd(*args, st), "mlp_fwd")
            self._qimg_b = -1

    def forward(self, idx: torch.Tensor, b: int) -> None:
        """idx int32 [b] device row indices into xp.  Fills Z, rinv, Zn, H, Q."""
        self.encode_partial(idx, b)
        self.mlp_forward(b)

    def _snp_ranges(self, n_parts: int, align: int):
        """[m0, m1) ranges covering the M SNPs, boundaries at multiples of ``align``."""
        M = self.lay.M
        units = (M + align - 1) // align
        n_parts = max(1, min(n_parts, units))
        cuts = [min(M, (units * i // n_parts) * align) for i in range(n_parts)] + [M]
        return [(cuts[i], cuts[i + 1]) for i in range(n_parts) if cuts[i + 1] > cuts[i]]

    def _adam_args(self, off_floats: int, fused) -> "AdamArgs":
        """nadm_adam_t for the rows of the big buffer that start at float offset ``off_floats``; fused = (lr, grad_scale) for an
        update in the kernel's epilogue with the current step count, or (lr, grad_scale, step) for the PREVIOUS step's update in
        the prologue of pass 2 (data-parallel step, nadm_adam_t.when = 1)."""
        if len(fused) == 3:
            return AdamArgs(self._mbig.data_ptr() + off_floats * 4, self._vbig.data_ptr() + off_floats * 4, fused[0], fused[2], fused[1], 1)
        lr, scale = fused
        return AdamArgs(self._mbig.data_ptr() + off_floats * 4, self._vbig.data_ptr() + off_floats * 4, lr, self.step_count, scale, 0)

    def decode_all(self, idx: torch.Tensor, b: int, with_loss: bool = True, on_grad_ready=None, p_parts=1,
                   supervised: bool = True, fused_adam=None) -> int:
        """Pass 2 for every head (optionally on SNP sub-ranges, see backward) + the supervised term.  Returns the number of
        loss slots the MLP backward has to add up."""
        L, st = self.lay, _stream()
        dq_offs, _ = L.dq_offsets(b)
        loss_offs = L.loss_offsets()
        fsz = 4
        ev = self._timed("decode_bce")
        # Several heads, no per-piece hand-off to a collective: the heads' launches are independent (own P rows, own dQ slab, own
        # loss slots; they share only Q and X), so they go round-robin onto a few HIP streams.  Each launch ends in a partly
        # filled round of resident blocks (M = 600k: 2344 blocks on 768 slots = 3.05 rounds); with two or three kernels in
        # flight the next head's blocks fill the slots the previous head's tail leaves empty.
        fan = self.head_streams if (len(L.ks) > 1 and on_grad_ready is None and self.device.type == "cuda") else 1
        if fan > 1:
            main = torch.cuda.current_stream()
            if self._head_streams is None or len(self._head_streams) < fan - 1:
                self._head_streams = [torch.cuda.Stream(device=self.device) for _ in range(fan - 1)]
                self._head_events = [torch.cuda.Event() for _ in range(fan)]
            self._head_events[0].record(main)
            for sd in self._head_streams[: fan - 1]:
                sd.wait_event(self._head_events[0])
        for h in range(len(L.ks)):
            kp = L.kp[h]
            if fan > 1 and h % fan:
                st = C.c_void_p(self._head_streams[h % fan - 1].cuda_stream)
            elif fan > 1:
                st = C.c_void_p(main.cuda_stream)
            csnps = int(lib.nadm_decode_chunk_snps(kp))
            align = csnps * 1024 // math.gcd(csnps, 1024)
            for m0, m1 in self._snp_ranges(p_parts, align):
                c0 = m0 // csnps
                args = (C.c_void_p(self.xp.data_ptr() + m0 // 4), self.ld, ptr(idx), b, m1 - m0,
                        C.c_void_p(self._big.data_ptr() + (L.p_off[h] + m0 * kp) * fsz), kp,
                        C.c_void_p(self.Q.data_ptr() + L.qoff[h] * fsz), L.SP,
                        C.c_void_p(self.gbig.data_ptr() + (L.p_off[h] + m0 * kp) * fsz),
                        C.c_void_p(self.dqpart.data_ptr() + (dq_offs[h] + c0 * b * kp) * fsz),
                        C.c_void_p(self.losspart.data_ptr() + (loss_offs[h] + c0) * fsz), (1 if self.p_unit else 3) if with_loss else 0)
                xg = C.c_void_p(self._xg_buf().data_ptr() + (m0 // 4) * b) if (h == 0 and self._gather()) else None   # (tiles of [b][128 B])
                if self.q_images and self._qimg_b == b and kp <= 16:      # Q operands ready-made by this step's MLP forward
                    check(lib.nadm_decode_bce_images(*args, xg, C.byref(self._adam_args(L.p_off[h] + m0 * kp, fused_adam)) if fused_adam is not None else None,
                                                     C.c_void_p(self.qimg.data_ptr() + h * self._qimg_head), st), "decode_bce_images")
                elif fused_adam is not None:                  # single-GPU step: Adam + clamp on these P rows in the kernel's epilogue
                    check(lib.nadm_decode_bce_step(*args, xg, C.byref(self._adam_args(L.p_off[h] + m0 * kp, fused_adam)), st), "decode_bce_step")
                elif xg is not None:                          # head 0's pass also leaves the batch's rows back to back in xg
                    check(lib.nadm_decode_bce_gather(*args, xg, st), "decode_bce_gather")
                else:
                    check(lib.nadm_decode_bce(*args, st), "decode_bce")
                if on_grad_ready is not None:
                    on_grad_ready(self._ns_pad + L.p_off[h] + m0 * kp, self._ns_pad + L.p_off[h] + m1 * kp)
        if fan > 1:                                          # join: everything after pass 2 waits for every head
            for j, sd in enumerate(self._head_streams[: fan - 1]):
                self._head_events[j + 1].record(sd)
                main.wait_event(self._head_events[j + 1])
            st = _stream()
        if ev: ev[1].record()
        self._xg_key = (idx.data_ptr(), b) if self._gather() else None   # pass 3 of THIS step, same batch: may read the copy
        n_loss = L.n_loss
        if self.labels is not None and supervised:
            check(lib.nadm_supervised_ce(ptr(self.Q), L.SP, L.ks[0], L.kp[0], ptr(self.labels), ptr(idx), b, self.n_classes,
                                         self.sup_weight, ptr(self.dqpart), C.c_void_p(self.losspart.data_ptr() + L.n_loss * fsz),
                                         st), "supervised_ce")
            n_loss += 1
        return n_loss

    def mlp_backward(self, b: int, n_loss: int, dq_src: Optional[torch.Tensor] = None, dq_M: Optional[int] = None,
                     weights: bool = True) -> None:
        """MLP / RMSNorm backward from the dQ partial slabs (default: this engine's dqpart over its M SNPs; ``dq_src`` with
        ``dq_M`` = 1 takes already reduced [b, kp_h] blocks).  n_loss > 0 adds that many loss slots to loss_acc.
        ``weights=False`` leaves the weight gradients to nadm_mlp_bwd_weights."""
        L, st = self.lay, _stream()
        self.flush_small()                                    # small_part is scratch of this call (no-op after a pass 1)
        args = (C.byref(L.heads), ptr(self.small), ptr(self.dqpart if dq_src is None else dq_src),
                L.M if dq_M is None else dq_M, b, ptr(self.Z), ptr(self.rinv), ptr(self.Zn),
                ptr(self.H), ptr(self.Q), ptr(self.dL), ptr(self.dHpre), ptr(self.dgp), ptr(self.small_part),
                ptr(self.dZ), ptr(self.gsmall) if weights else None, ptr(self.losspart), n_loss, ptr(self.loss_acc))
        if self._dzimg is not None:     # C <= 8: dZ also as the operand image of pass 3, built by the blocks that finish a 32-sample group
            check(lib.nadm_mlp_bwd_image(*args, ptr(self._dzimg), ptr(self._dzcnt), st), "mlp_bwd_image")
            self._dzimg_b = b
        else:
            check(lib.nadm_mlp_bwd(*args, st), "mlp_bwd")

    def _dz_image(self, b: int):
        """dZ [b, CP] as the operand image pass 3's matrix instruction consumes (C <= 8; include/nadm.h, nadm_dz_image): built once
        per step, read by every block of the pass."""
        L = self.lay
        if L.CP > 8:
            return None
        if self._dzimg_b != b:              # dZ did not come out of mlp_backward (tests that write dZ themselves call invalidate_dz)
            check(lib.nadm_dz_image(ptr(self.dZ), b, L.CP, ptr(self._dzimg), _stream()), "dz_image")
            self._dzimg_b = b
        return ptr(self._dzimg)

    def invalidate_dz(self) -> None:
        """Call after writing ``dZ`` from outside: the next pass 3 rebuilds its operand image."""
        self._dzimg_b = -1

    def encode_backward(self, idx: torch.Tensor, b: int, on_grad_ready=None, v_parts: int = 1, fused_adam=None,
                        side_weights: bool = False) -> None:
        """Pass 3: dV = X^T.dZ (optionally on SNP sub-ranges).  ``side_weights``: the MLP weight-gradient partials (left out by
        mlp_backward(weights=False)) are computed by extra blocks of the first launch, then summed into gsmall -- and, with
        ``fused_adam``, applied to the small parameters -- by one small launch (nadm_small_grads)."""
        L, st, fsz = self.lay, _stream(), 4
        mw = None
        if side_weights:
            mw = MlpWeights(C.pointer(L.heads), self.Zn.data_ptr(), self.H.data_ptr(), self.dL.data_ptr(), self.dHpre.data_ptr(),
                            self.dgp.data_ptr(), self.small_part.data_ptr())
        ev = self._timed("encode_bwd")
        # rows: the compact copy pass 2 of this step left in xg (rows 0..b-1 = the batch in order), else the resident matrix
        if self._xg_key == (idx.data_ptr(), b) and self._xg is not None:
            src, rows, xflags, rstride = self._xg, self._iota, 1, b   # NADM_X_CLEAN: pass 2's tiled copy of the batch (missing = 0)
        else:
            src, rows, xflags, rstride = self.xp, idx, 0, 1
        self._xg_key = None
        dzimg = self._dz_image(b)
        for i, (m0, m1) in enumerate(self._snp_ranges(v_parts, 1024)):
            side = mw if i == 0 else None
            if fused_adam is not None or side is not None:    # Adam on these V rows in the epilogue and / or the side blocks
                check(lib.nadm_encode_bwd_step(C.c_void_p(src.data_ptr() + (m0 // 4) * rstride), self.ld, ptr(rows), b, m1 - m0, ptr(self.dZ), dzimg, L.CP,
                                               C.c_void_p(self._big.data_ptr() + m0 * L.CP * fsz),
                                               C.c_void_p(self.gbig.data_ptr() + m0 * L.CP * fsz),
                                               C.byref(self._adam_args(m0 * L.CP, fused_adam)) if fused_adam is not None else None,
                                               C.byref(side) if side is not None else None, xflags, st), "encode_bwd_step")
                if side is not None and fused_adam is not None and self.defer_small:
                    # sum of the partials + Adam on the small parameters: owed to the next pass 1 (or to whoever looks first)
                    self._pending_small = (int(lib.nadm_sample_splits(b)), fused_adam[0], fused_adam[1], self.step_count)
                elif side is not None:
                    sa = None
                    if fused_adam is not None:
                        sa = C.byref(AdamArgs(self.msmall.data_ptr(), self.vsmall.data_ptr(), fused_adam[0], self.step_count, fused_adam[1]))
                    check(lib.nadm_small_grads(ptr(self.small_part), int(lib.nadm_sample_splits(b)), L.n_small, ptr(self.gsmall),
                                               ptr(self.small), sa, st), "small_grads")
            else:
                check(lib.nadm_encode_bwd(C.c_void_p(src.data_ptr() + (m0 // 4) * rstride), self.ld, ptr(rows), b, m1 - m0, ptr(self.dZ), dzimg, L.CP,
                                          C.c_void_p(self.gbig.data_ptr() + m0 * L.CP * fsz), xflags, st), "encode_bwd")
            if on_grad_ready is not None:                     # the first piece carries the small gradients in front of it
                on_grad_ready(0 if i == 0 else self._ns_pad + m0 * L.CP, self._ns_pad + m1 * L.CP)
        if ev: ev[1].record()

    def backward(self, idx: torch.Tensor, b: int, with_loss: bool = True, on_grad_ready=None, p_parts=1, v_parts: int = 1,
                 fused_adam=None, side_weights: bool = False, pre_adam=None) -> None:
        """Decoder + BCE fwd/bwd per head, MLP backward, dV.  Gradients land in gbig / gsmall (views of gflat).
        ``on_grad_ready(lo, hi)`` is invoked each time a contiguous piece gflat[lo:hi] of the big gradients is final and
        enqueued; passes 2 and 3 are launched on ``p_parts`` / ``v_parts`` SNP sub-ranges so that the data-parallel step can
        all-reduce one piece while the next is being computed (each head's P, or the two parts of a single head's P; the
        small gradients travel with the first piece of dV).  ``pre_adam`` = (lr, grad_scale, step): pass 2 first applies that
        (previous) step's Adam + clamp to its P rows from the gradient lying in gbig, then overwrites it (data-parallel step)."""
        n_loss = self.decode_all(idx, b, with_loss, on_grad_ready, p_parts, fused_adam=fused_adam if pre_adam is None else pre_adam)
        self.mlp_backward(b, n_loss if with_loss else 0, weights=not side_weights)
        self.encode_backward(idx, b, on_grad_ready, v_parts, fused_adam=fused_adam, side_weights=side_weights)

    def adam_part(self, part: str, lr: float, grad_scale: float = 1.0, stream=None) -> None:
        """Adam (+ clamp for P) on one part of the parameters -- "P", "V" or "small" -- for the CURRENT step_count."""
        L, fsz = self.lay, 4
        st = _stream() if stream is None else stream
        if part == "P":
            off = L.clamp_from * fsz
            check(lib.nadm_adam(C.c_void_p(self._big.data_ptr() + off), C.c_void_p(self.gbig.data_ptr() + off),
                                C.c_void_p(self._mbig.data_ptr() + off), C.c_void_p(self._vbig.data_ptr() + off),
                                L.n_big - L.clamp_from, 0, lr, self.step_count, grad_scale, st), "adam(P)")
        elif part == "V":
            check(lib.nadm_adam(ptr(self._big), ptr(self.gbig), ptr(self._mbig), ptr(self._vbig), L.clamp_from, L.clamp_from,
                                lr, self.step_count, grad_scale, st), "adam(V)")
        else:
            check(lib.nadm_adam(ptr(self.small), ptr(self.gsmall), ptr(self.msmall), ptr(self.vsmall), L.n_small, L.n_small,
                                lr, self.step_count, grad_scale, st), "adam(small)")

    def adam_v_small(self, lr: float, grad_scale: float, step: Optional[int] = None) -> None:
        """Adam on V and on the small parameters in ONE launch (nadm_adam2) for step count ``step`` (default: the current one): in
        the data-parallel step both become final together, behind the [small | dV] all-reduce."""
        L = self.lay
        check(lib.nadm_adam2(ptr(self._big), ptr(self.gbig), ptr(self._mbig), ptr(self._vbig), L.clamp_from, L.clamp_from,
                             ptr(self._small), ptr(self._gsmall), ptr(self._msmall), ptr(self._vsmall), L.n_small,
                             lr, self.step_count if step is None else step, grad_scale, _stream()), "adam2(V, small)")

    def adam_p_range(self, lo: int, hi: int, lr: float, grad_scale: float, step: int, stream=None) -> None:
        """Adam + clamp on elements [lo, hi) of the big buffer (a range inside the P matrices) for step count ``step``."""
        fsz = 4
        st = _stream() if stream is None else stream
        check(lib.nadm_adam(C.c_void_p(self._big.data_ptr() + lo * fsz), C.c_void_p(self.gbig.data_ptr() + lo * fsz),
                            C.c_void_p(self._mbig.data_ptr() + lo * fsz), C.c_void_p(self._vbig.data_ptr() + lo * fsz),
                            hi - lo, 0, lr, step, grad_scale, st), "adam(P range)")

    def adam(self, lr: float, grad_scale: float = 1.0) -> None:
        L, st = self.lay, _stream()
        self.step_count += 1
        ev = self._timed("adam")
        check(lib.nadm_adam(ptr(self._big), ptr(self.gbig), ptr(self._mbig), ptr(self._vbig), L.n_big, L.clamp_from,
                            lr, self.step_count, grad_scale, st), "adam(big)")
        self.adam_part("small", lr, grad_scale)
        self.p_unit = True                                    # the launch clamps P to [0, 1] (restrict_P)
        if ev: ev[1].record()

    def train_step(self, idx: torch.Tensor, b: int, lr: float, with_loss: bool = True) -> None:
        """One single-GPU step (neural_admixture.py:403-414 without the per-step host sync)."""
        if self._pending_ddp:                                 # a data-parallel step before this one left its P update to "the next pass 2":
            self.finish_ddp()                                 # this step's pass 2 has no prologue update (V / small: this step's pass 1 takes them)
        if self.fused_adam:
            # Adam on P and V where their gradients are completed (epilogues of passes 2 and 3, nadm_*_step): same element
            # update, same bits as the separate launches; the big gradient buffer is not written in this mode
            self.forward(idx, b)
            self.step_count += 1
            self.backward(idx, b, with_loss, fused_adam=(lr, 1.0), side_weights=True)    # small parameters: nadm_small_grads
            self.p_unit = True                                # restrict_P ran in pass 2's epilogue
            return
        self.forward(idx, b)
        self.backward(idx, b, with_loss)
        self.adam(lr)

    def train_step_ddp(self, idx: torch.Tensor, b: int, lr: float, world: int, with_loss: bool = True,
                       defer_tail: bool = False) -> None:
        """Sample-sharded data-parallel step: local gradients -> all-reduce(sum) over RCCL -> Adam with
        grad_scale 1/world (DDP's mean, neural_admixture.py:315-319).  Message plan: every head's P gradient is handed to RCCL
        when the pass-2 launch that completes it has been enqueued (asynchronous: it travels underneath the next head's pass 2,
        the MLP backward and pass 3); the small gradients and dV follow as ONE message right behind pass 3, issued on the
        compute stream itself -- it is the message the next step's pass 1 waits for, and a collective on the caller's stream
        costs no cross-stream event hand-off (torch >= 2.8 runs async_op=False collectives there).

        ``defer_tail`` (the trainer's and the bench's mode): NO Adam launch follows the messages.  The update of P is left to the
        prologue of the next step's pass 2 (every block updates its own rows from the all-reduced gradient before it uses them),
        the update of V to the prologue of the next step's pass 1, the small parameters' to side blocks of that launch -- or to
        finish_ddp().  Same arithmetic, same results; the parameter accessors (big, small, V(), P(), ...) apply what is pending
        first.

        (r02 cut a single head's pass 2 in two launches at its last round of resident blocks to put 3/4 of dP on the wire
        earlier: on a 1-rank group the second launch and the two extra stream hand-offs cost 41 us of a 0.48 ms step, and dP has
        until the NEXT step's pass 2 to arrive anyway -- profiles/r03_ddp_plan.txt.)"""
        import torch.distributed as dist
        L = self.lay
        works, pieces = [], []
        p_start = self._ns_pad + L.clamp_from

        def reduce_piece(lo, hi):                             # gflat = [small | pad | V | P heads]
            # A piece goes to RCCL right when the kernel that completes it has been enqueued -- BEFORE the next kernel is: the
            # collective waits on an event recorded at this point of the compute stream, so a piece handed over later would also
            # wait for whatever was enqueued in between
            works.append(dist.all_reduce(self.gflat[lo:hi], op=dist.ReduceOp.SUM, async_op=lo >= p_start))
            pieces.append((lo, hi))
        self.forward(idx, b)
        # The previous step's P update (defer_tail): its all-reduced gradient lies in gbig; this step's pass 2 applies Adam + clamp
        # to every block's own P rows in its prologue (nadm_adam_t.when = 1), so the update costs no launch and no extra read of P.
        pre = None
        if self._pending_ddp:
            pworks, plr, pscale, pstep = self._pending_ddp
            for w in pworks:
                if w is not None:
                    w.wait()                                  # the compute stream waits for the messages, the host does not
            pre = (plr, pscale, pstep)
            self._pending_ddp = None
        self.backward(idx, b, with_loss, on_grad_ready=reduce_piece, p_parts=1, v_parts=1, pre_adam=pre,
                      **({"side_weights": True} if self.side_weights else {}))
        scale = 1.0 / world
        self.step_count += 1
        off = self._ns_pad
        ev = self._timed("adam")
        p_works = []
        for w, (lo, hi) in zip(works, pieces):                # messages complete in the order they were enqueued
            if lo >= p_start and defer_tail:                  # a P piece: applied by the next step's pass 2 (or finish_ddp)
                p_works.append(w)
                continue
            if w is not None:
                w.wait()
            if lo >= p_start:                                 # a P piece: Adam on it while later messages are still in flight
                self.adam_p_range(lo - off, hi - off, lr, scale, self.step_count)
        if defer_tail:                                        # every [small | dV] piece is in: V and the small parameters are updated by
            self._pending_vs = (lr, scale, self.step_count)   # the next step's pass 1 (or by finish_ddp / the first accessor)
        else:
            self.adam_v_small(lr, scale)                      # one launch for both
        if ev: ev[1].record()
        self._pending_ddp = (p_works, lr, scale, self.step_count) if defer_tail else None
        self.p_unit = True                                    # P is clamped by its Adam launch / by the prologue of the pass that reads it next

    def finish_ddp(self) -> None:
        """Apply the updates train_step_ddp(defer_tail=True) left to the next step's passes 1 and 2 (end of training; the parameter
        accessors call it)."""
        self.flush_small()                                    # V + small parameters (or a single-GPU step's small update)
        if not self._pending_ddp:
            return
        works, lr, scale, step = self._pending_ddp
        for w in works:
            if w is not None:
                w.wait()
        L = self.lay
        self.adam_p_range(L.clamp_from, L.n_big, lr, scale, step)
        self._pending_ddp = None

    def infer_q(self, idx: torch.Tensor, b: int) -> List[torch.Tensor]:
        """Encoder-only pass (final Q, neural_admixture.py:369-383; src/inference.py:71-77)."""
        L = self.lay
        self.forward(idx, b)
        Q = self.Q[: b * L.SP].view(b, L.SP)
        return [Q[:, L.qoff[h]: L.qoff[h] + k].clone() for h, k in enumerate(L.ks)]

    def read_loss(self, reset: bool = True):
        """(running sum since last reset, last step) -- one host sync."""
        v = self.loss_acc.cpu().numpy().copy()
        if reset:
            self.loss_acc.zero_()
        return float(v[0]), float(v[1])
