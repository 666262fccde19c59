"""Parameter layout in HBM (mirrors include/nadm.h).

big   = V [M,CP] | P_0 [M,KP_0] | P_1 [M,KP_1] ...      (flat float32)
small = g[C] | W1[Hd,C] | b1[Hd] | Wk_0[k_0,Hd] | bk_0[k_0] | ...

The reference keeps the same tensors as separate nn.Parameters (Q_P, neural_admixture.py:100-150);
flat buffers make the gradient all-reduce one message per buffer and Adam one launch per buffer.
"""
import ctypes as C
from typing import List, Sequence

from ._lib import lib, Heads, check


class ModelLayout:
    def __init__(self, M: int, C_: int, Hd: int, ks: Sequence[int]):
        ks = sorted(int(k) for k in ks)
        self.M, self.C, self.Hd, self.ks = int(M), int(C_), int(Hd), ks
        self.heads = Heads()
        arr = (C.c_int32 * len(ks))(*ks)
        check(lib.nadm_heads_init(C.byref(self.heads), self.C, self.Hd, arr, len(ks)), "heads_init")
        h = self.heads
        self.CP, self.SP, self.n_small = h.CP, h.SP, h.n_small
        self.kp: List[int] = [h.kp[i] for i in range(len(ks))]
        self.qoff: List[int] = [h.qoff[i] for i in range(len(ks))]
        self.v_off = 0
        self.p_off: List[int] = []
        off = self.M * self.CP
        for kp in self.kp:
            self.p_off.append(off)
            off += self.M * kp
        self.n_big = off
        self.clamp_from = self.M * self.CP                 # restrict_P applies to the P part only
        self.enc_chunks = int(lib.nadm_encode_chunks(self.M))
        self.dec_chunks = [int(lib.nadm_decode_chunks(self.M, kp)) for kp in self.kp]
        self.n_loss = sum(self.dec_chunks)

    # element offsets of the per-head dQ partial slabs for batch size b
    def dq_offsets(self, b: int):
        offs, o = [], 0
        for ch, kp in zip(self.dec_chunks, self.kp):
            offs.append(o)
            o += ch * b * kp
        return offs, o

    def loss_offsets(self):
        offs, o = [], 0
        for ch in self.dec_chunks:
            offs.append(o)
            o += ch
        return offs

    @staticmethod
    def row_stride(M: int) -> int:
        """Packed row stride in bytes: ceil(M/4) rounded up to 16 (aligned 16 B loads; 64 / 128 / 256-byte row alignment was
        measured and changes nothing: the genotype passes are issue-bound, not request-bound)."""
        return ((int(M) + 3) // 4 + 15) // 16 * 16
