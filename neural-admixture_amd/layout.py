"""Parameter layout in HBM (mirrors include/nadm.h, nadm_flat_layout).

flat  = small | pad | V [M,CP] | gap | P_0 [M,KP_0] | P_1 [M,KP_1] ... | gap          (float32; one buffer each for parameters,
                                                                                       gradients and Adam moments)
small = g[C] | W1[Hd,C] | b1[Hd] | Wk_0[k_0,Hd] | bk_0[k_0] | ...
big   = flat[off_v:]  (V and every P; ``p_off`` / ``clamp_from`` below are offsets into it)

The reference keeps the same tensors as separate nn.Parameters (Q_P, neural_admixture.py:100-150); one flat buffer makes the
sample-sharded step's gradient exchange two messages -- B = [small | V], A = [all P] -- each cut into ``world`` equal slices
(reduce-scatter -> optimizer on the own slice -> all-gather); the gaps (zeros, world = 1: none) make the cuts come out even.
Message B travels as ``n_buckets`` SNP ranges (``bkt_*``: bucket j = flat[bkt_off[j]:bkt_off[j+1]] = V's rows of the SNPs
[bkt_m0[j], bkt_m0[j+1]), bucket 0 with the small parameters in front; every bucket is ``world`` slices of its own).
"""
import ctypes as C
from typing import List, Sequence

from ._lib import lib, Heads, FlatLayout, check


class ModelLayout:
    def __init__(self, M: int, C_: int, Hd: int, ks: Sequence[int], world: int = 1, n_buckets: int = 1):
        ks = sorted(int(k) for k in ks)
        self.M, self.C, self.Hd, self.ks, self.world = int(M), int(C_), int(Hd), ks, int(world)
        self.heads = Heads()
        arr = (C.c_int32 * len(ks))(*ks)
        check(lib.nadm_heads_init(C.byref(self.heads), self.C, self.Hd, arr, len(ks)), "heads_init")
        h = self.heads
        self.CP, self.SP, self.n_small = h.CP, h.SP, h.n_small
        self.kp: List[int] = [h.kp[i] for i in range(len(ks))]
        self.qoff: List[int] = [h.qoff[i] for i in range(len(ks))]
        fl = FlatLayout()
        check(lib.nadm_flat_layout(C.byref(self.heads), self.M, self.world, int(n_buckets), C.byref(fl)), "flat_layout")
        self.n_flat, self.off_v = int(fl.n_flat), int(fl.off_v)
        self.slice_b, self.slice_a, self.msg_a_off = int(fl.slice_b), int(fl.slice_a), int(fl.msg_a_off)
        nb = self.n_buckets = int(fl.n_buckets)                    # the number actually cut (M may allow fewer than asked for)
        self.bkt_off: List[int] = [int(fl.bkt_off[j]) for j in range(nb + 1)]
        self.bkt_slice: List[int] = [int(fl.bkt_slice[j]) for j in range(nb)]
        self.bkt_m0: List[int] = [int(fl.bkt_m0[j]) for j in range(nb + 1)]
        self.bkt_mom: List[int] = [int(fl.bkt_mom[j]) for j in range(nb)]
        self.v_off = 0
        self.p_off: List[int] = [int(fl.off_p[i]) - self.off_v for i in range(len(ks))]     # into big = flat[off_v:]
        self.n_big = self.n_flat - self.off_v
        self.clamp_from = self.p_off[0]                     # restrict_P applies to the P part only
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
        """Packed row stride in bytes: ceil(M/4) rounded up to 128, the size of the L2's requests to memory (the kernels need 16).
        A gathered row piece then never straddles a request: with 16-byte-aligned rows pass 2's 64-byte pieces cost 1.37 requests
        each and pass 1's 512-byte pieces five instead of four (request-size counters, profiles/r04_pmc_req.json).  Time is
        unchanged -- the genotype passes are issue-bound -- the bytes moved are not."""
        return ((int(M) + 3) // 4 + 127) // 128 * 128
