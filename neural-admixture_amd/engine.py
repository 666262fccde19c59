"""Device-side state of the training hot path.

One Engine per process / GPU.  It owns (through torch) every HBM buffer and hands them to the C library as ONE plan
(include/nadm.h, nadm_plan_desc_t); ``train_step`` is then a single C call, nadm_step, which queues what
``NeuralAdmixture._run_epoch`` / ``_run_step`` do per batch (neural_admixture.py:394-432) -- gather + decode, forward, loss,
backward, the gradient exchange, Adam, restrict_P -- without materialising any [b, M] tensor.  The sequencing, the fused
epilogues and the hand-offs between launches live in csrc/nadm_step.hip; nothing of that is Python state.

``mode`` says how the work of a step is spread over the ranks of ``comm`` (comm.py):
  "single"  one GPU
  "dp"      samples sharded, the reference's scheme (neural_admixture.py:287,315-319): gradients summed over ranks, optimizer
            sharded -- this rank holds Adam moments for, and updates, 1/world of the parameters; all-gather of the result
  "snp"     SNPs sharded (snp_parallel.SnpShardedEngine)

``forward`` / ``backward`` / ``adam`` below are the same step as three plain phases with the gradients left in ``gflat`` -- what
autograd's ``loss.backward()`` + ``optimizer.step()`` expose in the reference.  The parity tests inspect gradients through them;
the trainer does not use them.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ._lib import lib, check, ptr, PlanDesc, MODE_SINGLE, MODE_DP, MODE_SNP, T_NAMES, MAX_BUCKETS
from .layout import ModelLayout

_f32 = torch.float32
_MODES = {"single": MODE_SINGLE, "dp": MODE_DP, "snp": MODE_SNP}
_PIN_BYTES = 256 << 20                                   # the two pinned staging buffers of pack_from_host together (1024 rows each at M = 500k:
                                                         # 0.40 s per 100k rows; 256-row chunks 0.48, the packing alone 0.34 -- profiles/r05_io_timing.txt)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    @classmethod
    def supports(cls, device: torch.device) -> bool:
        """Whether this engine class runs on ``device``: the HIP engine needs a ROCm GPU -- there is no CPU fallback.  (A subclass
        that computes the step some other way says so here; model.NeuralAdmixture.engine_cls is the extension point.)"""
        return device.type == "cuda"

    def __init__(self, M: int, C_: int, Hd: int, ks: Sequence[int], device: torch.device, max_batch: int,
                 mode: str = "single", comm=None, n_buckets: int = 1, comm_a=None, p3_whole: bool = False, debug: bool = False):
        if not self.supports(device):
            raise RuntimeError("neural_admixture_amd.Engine needs a ROCm GPU device (no CPU fallback)")
        if mode not in _MODES:
            raise ValueError("mode must be 'single', 'dp' or 'snp'")
        if mode == "single" and comm is not None and comm.world != 1:
            raise ValueError("mode 'single' with a communicator of several ranks")
        if comm_a is not None and (mode != "dp" or comm is None):
            raise ValueError("comm_a is the second communicator (message A) of mode 'dp'")
        self.device, self.mode, self.comm, self.comm_a = device, mode, comm, comm_a
        self.p3_whole, self.debug = bool(p3_whole), bool(debug)
        self.world, self.rank = (comm.world, comm.rank) if comm is not None else (1, 0)
        # "dp": message B = [small | V] travels as n_buckets SNP ranges (csrc/nadm_step.hip); the layout says how many M allows
        self.lay = L = ModelLayout(M, C_, Hd, ks, self.world if mode == "dp" else 1, n_buckets if mode == "dp" else 1)
        self.M, self.ld = L.M, ModelLayout.row_stride(L.M)
        self.bmax = b = int(max_batch)
        z = lambda n, dt=_f32: torch.zeros(int(n), dtype=dt, device=device)
        # parameters and gradients: flat [small | pad | V | P...] (layout.py).  Adam moments: flat like them, except in "dp" mode
        # with several ranks, where this rank keeps the moments of its own two slices only: [slice of message B | slice of message A]
        self.pflat, self.gflat = z(L.n_flat), z(L.n_flat)
        self.moments_sharded = mode == "dp" and self.world > 1
        n_mom = L.slice_b + L.slice_a if self.moments_sharded else L.n_flat
        self.mflat, self.vflat = z(n_mom), z(n_mom)
        # activations / scratch of a step
        self.zpart = z(L.enc_chunks * b * L.CP)
        self.Z, self.rinv, self.Zn = z(b * L.CP), z(b), z(b * L.CP)
        self.H, self._Q = z(b * L.Hd), z(b * L.SP)
        self.dL, self.dHpre, self.dgp, self._dZ = z(b * L.SP), z(b * L.Hd), z(b * L.CP), z(b * L.CP)
        self.dqpart = z(L.dq_offsets(b)[1])
        self.losspart = z(L.n_loss + 1)                  # last slot: supervised term (nadm_supervised_ce)
        self.small_part = z(int(lib.nadm_sample_splits(b)) * L.n_small)
        self.loss_acc = torch.zeros(2, dtype=torch.float64, device=device)
        self._zsum = z(b * L.CP) if mode == "snp" else None
        self._dqsum = z(b * L.SP) if mode == "snp" else None
        gpu = device.type == "cuda"
        # Q as the bf16 operand images of pass 2, written by the MLP forward (heads with padded K <= 16; zero-filled once, one
        # region per head); dZ as the FP6 operand image of pass 3, written by the MLP backward (C <= 8); the batch as a copy of
        # its own, tiled by pass 3's chunks, written by pass 2 (C <= 8) -- include/nadm.h
        self._qimg_head = int(lib.nadm_q_image_bytes(b))
        self.qimg = torch.zeros(len(L.ks) * self._qimg_head, dtype=torch.uint8, device=device) if gpu and any(kp <= 16 for kp in L.kp) else None
        tiled = gpu and L.CP <= 8
        self._dzimg = torch.zeros(int(lib.nadm_dz_image_bytes(b)), dtype=torch.uint8, device=device) if tiled else None
        self._dzcnt = torch.zeros((b + 31) // 32, dtype=torch.int32, device=device)      # group counters of nadm_mlp_bwd_image
        self._xg = torch.empty(int(lib.nadm_batch_copy_bytes(b, self.M)), dtype=torch.uint8, device=device) if tiled else None
        self._iota = torch.arange(b, dtype=torch.int32, device=device) if tiled else None
        # pass 2 in sample slices (include/nadm.h, nadm_decode_bce_sliced): where the SNP chunks alone leave CUs idle (M below ~330k)
        # every slice parks its partial of dP in a slab; one region per head, one counter per chunk
        self._p2_slab_off, self._p2_slices_cap = [0], []
        for kp in L.kp:
            self._p2_slices_cap.append(int(lib.nadm_decode_slices_max(b, self.M, kp)))
            self._p2_slab_off.append(self._p2_slab_off[-1] + int(lib.nadm_decode_slab_floats(self.M, kp, self._p2_slices_cap[-1])))
        self._p2_slab = z(self._p2_slab_off[-1]) if gpu and self._p2_slab_off[-1] else None
        self._p2_cnt = torch.zeros(L.n_loss, dtype=torch.int32, device=device) if self._p2_slab is not None else None
        # pass 3 likewise (nadm_encode_bwd_sliced, r06): an SNP-sharded rank's few chunks x many rows
        self._p3_slices_cap = int(lib.nadm_encode_slices_max(b, self.M, L.CP)) if tiled else 1
        n3 = int(lib.nadm_encode_slab_floats(self.M, L.CP, self._p3_slices_cap))
        self._p3_slab = z(n3) if n3 else None
        self._p3_cnt = torch.zeros(int(lib.nadm_encode_bwd_chunks(self.M)), dtype=torch.int32, device=device) if n3 else None
        # validity of the three by-products for the plain phases below (the plan's own step always produces what it consumes)
        self._qimg_b = self._dzimg_b = -1
        self._dz_last_b = 0
        self._xg_key = None
        self.xp: Optional[torch.Tensor] = None          # packed genotypes [rows, ld]
        self.labels: Optional[torch.Tensor] = None      # int32 [rows], supervised mode only
        self.n_classes, self.sup_weight = 0, 0.0
        self._pin: Optional[torch.Tensor] = None
        self._plan = None
        self._step_py, self._p_unit_py = 0, True         # the CPU double's counterparts of the plan's state
        if gpu:
            self._make_plan()

    # ------------------------------------------------------------------ the plan
    def _make_plan(self) -> None:
        L = self.lay
        d = PlanDesc()
        d.mode, d.bmax, d.M, d.ld, d.heads = _MODES[self.mode], self.bmax, L.M, self.ld, L.heads
        for name, t in (("params", self.pflat), ("grads", self.gflat), ("m", self.mflat), ("v", self.vflat), ("zpart", self.zpart),
                        ("Z", self.Z), ("rinv", self.rinv), ("Zn", self.Zn), ("H", self.H), ("Q", self._Q), ("dL", self.dL),
                        ("dHpre", self.dHpre), ("dgp", self.dgp), ("dZ", self._dZ), ("dqpart", self.dqpart), ("losspart", self.losspart),
                        ("small_part", self.small_part), ("zsum", self._zsum), ("dqsum", self._dqsum), ("qimg", self.qimg),
                        ("dzimg", self._dzimg), ("dzcnt", self._dzcnt), ("xg", self._xg), ("loss_acc", self.loss_acc),
                        ("p2_slab", self._p2_slab), ("p2_cnt", self._p2_cnt), ("p3_slab", self._p3_slab), ("p3_cnt", self._p3_cnt)):
            setattr(d, name, None if t is None else t.data_ptr())
        d.qimg_head_bytes = self._qimg_head
        d.n_buckets, d.p3_whole, d.debug = L.n_buckets, int(self.p3_whole), int(self.debug)
        for c, field in ((self.comm, "comm"), (self.comm_a, "comm_a")):
            if c is None:
                continue
            setattr(d, field, c.handle)
            tr = getattr(c, "transport", None)
            if tr is not None:                             # torch.distributed callbacks: the buffers the step communicates
                tr.buffers += [t for t in (self.pflat, self.gflat, self._zsum, self._dqsum) if t is not None]
        plan = C.c_void_p()
        check(lib.nadm_plan_create(C.byref(d), C.byref(plan)), "plan_create")
        self._plan = plan

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            try:
                lib.nadm_plan_destroy(plan)
            except Exception:
                pass

    def _comm_error(self):
        for c in (self.comm, self.comm_a):
            tr = getattr(c, "transport", None)
            if tr is not None and tr.error is not None:
                e, tr.error = tr.error, None
                raise e

    def train_step(self, idx: torch.Tensor, b: int, lr: float, with_loss: bool = True) -> None:
        """One training step on the batch rows idx (int32 [b], device): ONE C call (neural_admixture.py:403-414 without the
        per-step host sync; in "dp" / "snp" mode including the collectives)."""
        if b > self.bmax:
            raise RuntimeError("batch larger than the engine was sized for")
        if lib.nadm_step(self._plan, ptr(idx), b, lr, 1 if with_loss else 0, _stream()):
            self._comm_error()
            check(1, "step")
        self._qimg_b = self._dzimg_b = -1
        self._xg_key = None
        self._dz_last_b = b                              # (the plan's step wrote the image of dZ for this batch size: the plain phases' hygiene sees it)

    def sync(self) -> None:
        """Make the current stream see every parameter final: a step may leave its small-parameter update to the next step's
        pass 1, and in "dp" mode its P message completes on a side stream (csrc/nadm_step.hip).  Every accessor below calls it."""
        if self._plan is not None:
            check(lib.nadm_plan_flush(self._plan, _stream()), "plan_flush")

    def time_kernels(self, names: Optional[Sequence[str]]) -> None:
        """HIP events around these launch groups of every following step (_lib.T_NAMES; None / empty: off)."""
        mask = sum(1 << T_NAMES.index(n) for n in (names or ()))
        check(lib.nadm_plan_timing(self._plan, mask), "plan_timing")

    def kernel_ms(self) -> Dict[str, float]:
        """Mean duration [ms] of every timed group since the last call (synchronises).  With "sync_b" timed in "dp" mode also
        ``sync_b_buckets``: the list of per-bucket means (reduce-scatter -> Adam -> all-gather of each SNP range)."""
        ms, cnt = (C.c_float * len(T_NAMES))(), (C.c_int32 * len(T_NAMES))()
        bms, nb = (C.c_float * MAX_BUCKETS)(), C.c_int32(0)
        check(lib.nadm_plan_bucket_ms(self._plan, bms, C.byref(nb)), "plan_bucket_ms")
        check(lib.nadm_plan_kernel_ms(self._plan, ms, cnt), "plan_kernel_ms")
        out = {n: float(ms[i]) for i, n in enumerate(T_NAMES) if cnt[i]}
        if "sync_b" in out:
            out["sync_b_buckets"] = [float(bms[j]) for j in range(nb.value)]
        return out

    # step count and "every P entry lies in [0, 1]" live in the plan; the plain phases below read and advance them too
    def _get_step(self):
        return self._step_py if self._plan is None else int(lib.nadm_plan_step_count(self._plan))

    def _get_unit(self):
        return self._p_unit_py if self._plan is None else bool(lib.nadm_plan_p_in_unit_range(self._plan))

    def _set_state(self, step, unit):
        self._step_py, self._p_unit_py = int(step), bool(unit)
        if self._plan is not None:
            check(lib.nadm_plan_set_state(self._plan, int(step), 1 if unit else 0), "plan_set_state")

    step_count = property(_get_step, lambda self, t: self._set_state(t, self._get_unit()))
    p_unit = property(_get_unit, lambda self, u: self._set_state(self._get_step(), u))

    # ------------------------------------------------------------------ data
    def set_packed(self, xp: torch.Tensor) -> None:
        if xp.dtype != torch.uint8 or xp.dim() != 2 or xp.shape[1] != self.ld or not xp.is_contiguous():
            raise RuntimeError(f"packed genotypes must be contiguous uint8 [rows, {self.ld}]")
        self.xp = xp
        if self._plan is not None:
            check(lib.nadm_plan_set_rows(self._plan, ptr(xp)), "plan_set_rows")

    def pack_from_host(self, data_u8: torch.Tensor, rows: Optional[np.ndarray] = None, chunk_rows: Optional[int] = None) -> None:
        """uint8 [N,M] CPU tensor -> packed rows in HBM.  Packs on the host (2 bits/genotype cross PCIe instead of 8; the
        reference ships unpacked bytes in 1024-row chunks with a blocking sync per chunk, pack2bit.cu:79-115).  ``rows``:
        optional row selection/order (rank shard)."""
        self.rows_are_sharded = rows is not None
        if hasattr(data_u8, "packed"):                   # io.PackedGenotypes: already in the kernel layout, just ship the rows
            if data_u8.M != self.M or data_u8.packed.shape[1] != self.ld:
                raise RuntimeError("packed genotypes do not match the engine's SNP count / row stride")
            src = data_u8.packed if rows is None else data_u8.packed[torch.as_tensor(rows, dtype=torch.long)]
            self.set_packed(src.contiguous().to(self.device))
            return
        if data_u8.dtype != torch.uint8 or data_u8.dim() != 2 or data_u8.device.type != "cpu":
            raise RuntimeError("pack_from_host expects a uint8 [N,M] CPU tensor")
        N, M = data_u8.shape
        if M != self.M:
            raise RuntimeError("pack_from_host: SNP count mismatch")
        n_out = N if rows is None else len(rows)
        xp = torch.empty((n_out, self.ld), dtype=torch.uint8, device=self.device)
        if chunk_rows is None:                           # staging buffers sized by bytes (M = 500k: 1024 rows each), pinned once per engine
            chunk_rows = max(1, min(max(n_out, 1), _PIN_BYTES // (2 * self.ld)))
        if self._pin is None or self._pin.numel() < 2 * chunk_rows * self.ld:
            self._pin = torch.empty(2 * chunk_rows * self.ld, dtype=torch.uint8)
            if torch.cuda.is_available():
                self._pin = self._pin.pin_memory()
        # two staging buffers: chunk i is packed on the host threads while chunk i - 1 crosses PCIe on a stream of its own (one buffer and
        # a blocking copy: 0.33 s of packing + 0.22 s of copies for 100k x 500k, profiles/r05_io_timing.txt; now the longer of the two)
        stages = [self._pin[i * chunk_rows * self.ld: (i + 1) * chunk_rows * self.ld].view(chunk_rows, self.ld) for i in range(2)]
        if self.device.type != "cuda":                   # (the tests' CPU stand-in of the engine)
            for s in range(0, n_out, chunk_rows):
                e = min(n_out, s + chunk_rows)
                src = (data_u8[s:e] if rows is None else data_u8[torch.as_tensor(rows[s:e], dtype=torch.long)]).contiguous()
                check(lib.nadm_pack2bit_host(ptr(src), ptr(stages[0]), e - s, M, self.ld), "pack2bit_host")
                xp[s:e].copy_(stages[0][: e - s])
            self.set_packed(xp)
            return
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        freed = [None, None]
        for i, s in enumerate(range(0, n_out, chunk_rows)):
            e = min(n_out, s + chunk_rows)
            src = data_u8[s:e] if rows is None else data_u8[torch.as_tensor(rows[s:e], dtype=torch.long)]
            src = src.contiguous()
            stage = stages[i & 1]
            if freed[i & 1] is not None:
                freed[i & 1].synchronize()               # the copy that last read this buffer
            check(lib.nadm_pack2bit_host(ptr(src), ptr(stage), e - s, M, self.ld), "pack2bit_host")
            with torch.cuda.stream(side):
                xp[s:e].copy_(stage[: e - s], non_blocking=True)
                freed[i & 1] = torch.cuda.Event()
                freed[i & 1].record(side)
        torch.cuda.current_stream(self.device).wait_stream(side)
        side.synchronize()                               # (the staging buffers may be reused by the next call right away)
        self.set_packed(xp)

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
        if self._plan is not None:
            check(lib.nadm_plan_set_labels(self._plan, ptr(self.labels), self.n_classes, self.sup_weight), "plan_set_labels")

    # ------------------------------------------------------------------ parameters
    def load_params(self, V_MC: np.ndarray, P_SM: np.ndarray, small: np.ndarray) -> None:
        """V_MC [M,C]; P_SM [sum(ks), M] (reference P_init layout, train.py:63,67); small = flat
        g|W1|b1|Wk|bk in the nadm.h order."""
        L = self.lay
        self.sync()                                      # whatever a step still owes lands before it is overwritten
        flat = np.zeros(L.n_flat, dtype=np.float32)
        flat[: L.n_small] = np.ascontiguousarray(small, dtype=np.float32)
        big = flat[L.off_v:]
        big[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C] = V_MC
        ini = 0
        for h, k in enumerate(L.ks):
            big[L.p_off[h]: L.p_off[h] + L.M * L.kp[h]].reshape(L.M, L.kp[h])[:, :k] = P_SM[ini:ini + k].T
            ini += k
        self.pflat.copy_(torch.from_numpy(flat))
        for t in (self.mflat, self.vflat, self.gflat, self._dzcnt):
            t.zero_()
        self._qimg_b = self._dzimg_b = -1
        # the loss value of pass 2 may skip the clamp of the reconstruction while every P entry lies in [0, 1]; true after the
        # first restrict_P, and for the GMM initialisation (clipped to [5e-6, 1 - 5e-6]) -- not for the supervised one
        self._set_state(0, bool(np.min(P_SM) >= 0.0 and np.max(P_SM) <= 1.0) if np.size(P_SM) else True)

    def _synced(self, t):
        self.sync()
        return t

    def _full_moments(self, t):
        if self.moments_sharded:
            raise RuntimeError("'dp' mode over several ranks keeps the Adam moments of this rank's parameter slices only (mflat / vflat)")
        return self._synced(t)

    # views of the flat buffers; reading any of them first settles what the last step left to the next one
    small = property(lambda self: self._synced(self.pflat[: self.lay.n_small]))
    big = property(lambda self: self._synced(self.pflat[self.lay.off_v:]))
    gsmall = property(lambda self: self._synced(self.gflat[: self.lay.n_small]))
    gbig = property(lambda self: self._synced(self.gflat[self.lay.off_v:]))
    msmall = property(lambda self: self._full_moments(self.mflat[: self.lay.n_small]))
    vsmall = property(lambda self: self._full_moments(self.vflat[: self.lay.n_small]))
    mbig = property(lambda self: self._full_moments(self.mflat[self.lay.off_v:]))
    vbig = property(lambda self: self._full_moments(self.vflat[self.lay.off_v:]))

    def V(self) -> torch.Tensor:
        L = self.lay
        return self.big[: L.M * L.CP].view(L.M, L.CP)[:, : L.C]

    def P(self, h: int) -> torch.Tensor:
        L = self.lay
        return self.big[L.p_off[h]: L.p_off[h] + L.M * L.kp[h]].view(L.M, L.kp[h])[:, : L.ks[h]]

    def gV(self) -> torch.Tensor:
        L = self.lay
        return self.gbig[: L.M * L.CP].view(L.M, L.CP)[:, : L.C]

    def gP(self, h: int) -> torch.Tensor:
        L = self.lay
        return self.gbig[L.p_off[h]: L.p_off[h] + L.M * L.kp[h]].view(L.M, L.kp[h])[:, : L.ks[h]]

    # Q and dZ are written by the launches; a caller that assigns or edits them drops the operand images built from them
    def _set_q(self, t):
        if t is not self._Q:
            self._Q[: t.numel()].copy_(t.reshape(-1))     # (the plan holds the buffer's address: assign into it)
        self._qimg_b = -1

    def _set_dz(self, t):
        if t is not self._dZ:
            self._dZ[: t.numel()].copy_(t.reshape(-1))
        self._dzimg_b = -1

    def _touch_dz(self):
        self._dzimg_b = -1                               # handed out: may be edited in place -> the next pass 3 rebuilds its image
        return self._dZ

    Q = property(lambda self: self._Q, _set_q)
    dZ = property(_touch_dz, _set_dz)

    def invalidate_q(self) -> None:
        """Call after editing Q (or P through a raw view) in place: pass 2 rebuilds its operands from the fp32 values."""
        self._qimg_b = -1
        self.p_unit = False                                   # P may hold anything: the loss path clamps until the next restrict_P

    # ------------------------------------------------------------------ the step as three plain phases (gradients visible)
    def encode_partial(self, idx: torch.Tensor, b: int) -> None:
        """Pass 1: per-chunk partial sums of Z = X.V for the batch rows idx (int32 [b], device) into zpart."""
        L = self.lay
        if b > self.bmax:
            raise RuntimeError("batch larger than the engine was sized for")
        check(lib.nadm_encode_fwd(ptr(self.xp), self.ld, ptr(idx), b, L.M, ptr(self.big[: L.M * L.CP]), L.CP, ptr(self.zpart), _stream()), "encode_fwd")

    def mlp_forward(self, b: int, z_src: Optional[torch.Tensor] = None, n_chunks: Optional[int] = None) -> None:
        """RMSNorm + MLP + per-head softmax from partial sums [n_chunks, b, CP] (default: this engine's zpart).  Fills Z,
        rinv, Zn, H, Q (and Q's operand images for pass 2)."""
        L = self.lay
        args = (C.byref(L.heads), ptr(self.small), ptr(self.zpart if z_src is None else z_src),
                L.enc_chunks if n_chunks is None else n_chunks, b, ptr(self.Z), ptr(self.rinv), ptr(self.Zn), ptr(self.H), ptr(self._Q))
        if self.qimg is not None:
            check(lib.nadm_mlp_fwd_images(*args, ptr(self.qimg), self._qimg_head, _stream()), "mlp_fwd_images")
            self._qimg_b = b
        else:
            check(lib.nadm_mlp_fwd(*args, _stream()), "mlp_fwd")

    def forward(self, idx: torch.Tensor, b: int) -> None:
        """idx int32 [b] device row indices into xp.  Fills Z, rinv, Zn, H, Q."""
        self.encode_partial(idx, b)
        self.mlp_forward(b)

    def decode_all(self, idx: torch.Tensor, b: int, with_loss: bool = True, supervised: bool = True) -> int:
        """Pass 2 for every head (gradient dP -> gbig, dQ slabs -> dqpart) + the supervised term.  Returns the number of loss
        slots the MLP backward has to add up."""
        L, st, fsz = self.lay, _stream(), 4
        dq_offs, _ = L.dq_offsets(b)
        loss_offs = L.loss_offsets()
        big, gbig = self.big, self.gflat[L.off_v:]
        flags = (1 if self.p_unit else 3) if with_loss else 0
        for h, kp in enumerate(L.kp):
            args = (ptr(self.xp), self.ld, ptr(idx), b, L.M, C.c_void_p(big.data_ptr() + L.p_off[h] * fsz), kp,
                    C.c_void_p(self._Q.data_ptr() + L.qoff[h] * fsz), L.SP, C.c_void_p(gbig.data_ptr() + L.p_off[h] * fsz),
                    C.c_void_p(self.dqpart.data_ptr() + dq_offs[h] * fsz), C.c_void_p(self.losspart.data_ptr() + loss_offs[h] * fsz), flags)
            xg = ptr(self._xg) if (h == 0 and self._xg is not None) else None     # head 0's launch leaves the batch copy for pass 3
            qi = C.c_void_p(self.qimg.data_ptr() + h * self._qimg_head) if (self.qimg is not None and self._qimg_b == b and kp <= 16) else None
            slices = int(lib.nadm_decode_slices(b, L.M, kp)) if self._p2_slab is not None else 1
            if slices > self._p2_slices_cap[h]:
                raise RuntimeError("pass 2 would be cut into more sample slices than this engine's slab was sized for")
            if slices > 1:                                                        # the library's cut of the batch, as in the step
                check(lib.nadm_decode_bce_sliced(*args, xg, None, qi, slices, C.c_void_p(self._p2_slab.data_ptr() + self._p2_slab_off[h] * fsz),
                                                 C.c_void_p(self._p2_cnt.data_ptr() + loss_offs[h] * 4), st), "decode_bce_sliced")
            elif qi is not None:                                                  # Q operands ready-made by this step's MLP forward
                check(lib.nadm_decode_bce_images(*args, xg, None, qi, st), "decode_bce_images")
            elif xg is not None:
                check(lib.nadm_decode_bce_gather(*args, xg, st), "decode_bce_gather")
            else:
                check(lib.nadm_decode_bce(*args, st), "decode_bce")
        self._xg_key = (idx.data_ptr(), b) if self._xg is not None else None      # pass 3 of THIS step, same batch: may read the copy
        n_loss = L.n_loss
        if self.labels is not None and supervised:
            check(lib.nadm_supervised_ce(ptr(self._Q), L.SP, L.ks[0], L.kp[0], ptr(self.labels), ptr(idx), b, self.n_classes,
                                         self.sup_weight, ptr(self.dqpart), C.c_void_p(self.losspart.data_ptr() + L.n_loss * fsz), st), "supervised_ce")
            n_loss += 1
        return n_loss

    def mlp_backward(self, b: int, n_loss: int, dq_src: Optional[torch.Tensor] = None, dq_M: Optional[int] = None) -> None:
        """MLP / RMSNorm backward from the dQ partial slabs (default: this engine's dqpart over its M SNPs; ``dq_src`` with
        ``dq_M`` = 1 takes already reduced [b, kp_h] blocks): dZ, the small gradients -> gsmall; n_loss > 0 adds that many loss
        slots to loss_acc."""
        L, st = self.lay, _stream()
        args = (C.byref(L.heads), ptr(self.small), ptr(self.dqpart if dq_src is None else dq_src),
                L.M if dq_M is None else dq_M, b, ptr(self.Z), ptr(self.rinv), ptr(self.Zn),
                ptr(self.H), ptr(self._Q), ptr(self.dL), ptr(self.dHpre), ptr(self.dgp), ptr(self.small_part),
                ptr(self._dZ), ptr(self.gflat), ptr(self.losspart), n_loss, ptr(self.loss_acc))
        if self._dzimg is not None:     # C <= 8: dZ also as the operand image of pass 3, built by the blocks that finish a 32-sample group
            if b < self._dz_last_b and b % 128:          # a shorter batch: clear the image's last tile first (include/nadm.h, as nadm_step does)
                tb = int(lib.nadm_dz_image_tile_bytes())
                self._dzimg[(b // 128) * tb: (b // 128 + 1) * tb].zero_()
            self._dz_last_b = b
            check(lib.nadm_mlp_bwd_image(*args, ptr(self._dzimg), ptr(self._dzcnt), st), "mlp_bwd_image")
            self._dzimg_b = b
        else:
            check(lib.nadm_mlp_bwd(*args, st), "mlp_bwd")

    def encode_backward(self, idx: torch.Tensor, b: int) -> None:
        """Pass 3: dV = X^T.dZ -> gbig, from the batch copy pass 2 of this step left (C <= 8), else from the resident matrix."""
        L, st = self.lay, _stream()
        dzimg = None
        if self._dzimg is not None:
            if self._dzimg_b != b:                      # dZ did not come straight out of mlp_backward
                check(lib.nadm_dz_image(ptr(self._dZ), b, L.CP, ptr(self._dzimg), st), "dz_image")
                self._dzimg_b = b
            dzimg = ptr(self._dzimg)
        if self._xg_key == (idx.data_ptr(), b):
            src, rows, flags = self._xg, self._iota, 1   # NADM_X_CLEAN
        else:
            src, rows, flags = self.xp, idx, 0
        self._xg_key = None
        slices = int(lib.nadm_encode_slices(b, L.M, L.CP)) if (self._p3_slab is not None and flags) else 1
        if slices > self._p3_slices_cap:
            raise RuntimeError("pass 3 would be cut into more sample slices than this engine's slab was sized for")
        if slices > 1:                                    # the library's cut of the batch, as in the step
            check(lib.nadm_encode_bwd_sliced(ptr(src), self.ld, ptr(rows), b, L.M, ptr(self._dZ), dzimg, L.CP, None, ptr(self.gflat[L.off_v:]), None, None,
                                             flags, slices, ptr(self._p3_slab), ptr(self._p3_cnt), st), "encode_bwd_sliced")
            return
        check(lib.nadm_encode_bwd(ptr(src), self.ld, ptr(rows), b, L.M, ptr(self._dZ), dzimg, L.CP, ptr(self.gflat[L.off_v:]), flags, st), "encode_bwd")

    def backward(self, idx: torch.Tensor, b: int, with_loss: bool = True) -> None:
        """Decoder + BCE fwd/bwd per head, MLP backward, dV.  Every gradient lands in gflat (gsmall / gbig are views)."""
        n_loss = self.decode_all(idx, b, with_loss)
        self.mlp_backward(b, n_loss if with_loss else 0)
        self.encode_backward(idx, b)

    def adam(self, lr: float, grad_scale: float = 1.0) -> None:
        """optimizer.step() + restrict_P (neural_admixture.py:187-204,411-412) on every parameter from the gradients in gflat."""
        L = self.lay
        if self.moments_sharded:
            raise RuntimeError("adam(): this engine holds the moments of its own parameter slices only; use train_step")
        self.sync()
        t = self.step_count + 1
        check(lib.nadm_adam(ptr(self.pflat), ptr(self.gflat), ptr(self.mflat), ptr(self.vflat), L.n_flat, L.off_v + L.clamp_from,
                            lr, t, grad_scale, _stream()), "adam")
        self._set_state(t, True)                         # the launch clamps P to [0, 1] (restrict_P)
        self._qimg_b = -1

    # ------------------------------------------------------------------ results
    def infer_q(self, idx: torch.Tensor, b: int) -> List[torch.Tensor]:
        """Encoder-only pass (final Q, neural_admixture.py:369-383; src/inference.py:71-77)."""
        L = self.lay
        if self._plan is not None:
            if lib.nadm_plan_infer(self._plan, ptr(idx), b, _stream()):
                self._comm_error()
                check(1, "plan_infer")
            self._qimg_b = -1
        else:
            self.forward(idx, b)
        Q = self._Q[: b * L.SP].view(b, L.SP)
        return [Q[:, L.qoff[h]: L.qoff[h] + k].clone() for h, k in enumerate(L.ks)]

    def read_loss(self, reset: bool = True):
        """(running sum since last reset, last step) -- one host sync."""
        v = self.loss_acc.cpu().numpy().copy()
        if reset:
            self.loss_acc.zero_()
        return float(v[0]), float(v[1])
