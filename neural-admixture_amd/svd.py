"""Randomized SVD initialisation of V (8(f)-2), same algorithm as the reference's ``RSVD`` (src/svd.py:39-83):
Omega ~ N(0,1) [M,k'] from ``np.random.default_rng(seed)``, k' = max(k+10, 20); Y = A.Omega; 2 power iterations
with QR; QR; B = Q^T.A; SVD(B); sign flip (svd.py:16-37); returns Vt[:k] [k,M] float32.  A = the RAW genotype
codes (0,1,2,3 with missing kept as 3, no centering), exactly like the reference.

The two tall-skinny products that the reference runs as naive Cython triple loops on the CPU
(src/utils_c/rsvd.pyx:16-50) run here on the GPU straight from the 2-bit packed matrix: rows are decoded
chunk-wise in HBM (nadm_unpack2bit) and multiplied with torch (plumbing, init-time only).  Without a GPU the same
code runs through numpy on the host."""
from __future__ import annotations

import logging
import sys
import time

import numpy as np
import torch

log = logging.getLogger(__name__)


def svd_flip(V: np.ndarray, U: np.ndarray) -> np.ndarray:
    idx = np.argmax(np.abs(U), axis=0)
    signs = np.sign(U[idx, np.arange(U.shape[1])])
    return V * signs[:, np.newaxis]


class _Rows:
    """Chunked access to the raw code matrix A [N,M] as float32, from uint8 [N,M] or io.PackedGenotypes."""

    def __init__(self, data, device):
        self.dev = device
        self.packed = hasattr(data, "packed")
        self.N, self.M = data.shape
        self.data = data
        self.xp = None
        if self.packed and device is not None and device.type == "cuda":
            self.xp = data.packed.to(device)

    def chunk(self, s, e):
        if self.xp is not None:
            from ._lib import lib, check, ptr
            out = torch.empty((e - s, self.M), dtype=torch.uint8, device=self.dev)
            check(lib.nadm_unpack2bit(ptr(self.xp[s:e]), ptr(out), e - s, self.M, self.xp.shape[1], None), "unpack2bit")
            return out.float()
        a = self.data.unpack_rows(s, e) if self.packed else np.asarray(self.data[s:e])
        t = torch.from_numpy(np.ascontiguousarray(a))
        return (t.to(self.dev) if self.dev is not None else t).float()


def RSVD(A_uint8, N: int, M: int, k: int = 8, seed: int = 42, oversampling: int = 10, power_iterations: int = 2,
         device: torch.device = None, rows: int = 2048) -> np.ndarray:
    if device is None:
        device = torch.device("cuda:0") if torch.cuda.is_available() else None
    old_prec = torch.get_float32_matmul_precision()
    torch.set_float32_matmul_precision("highest")
    try:
        src = _Rows(A_uint8 if hasattr(A_uint8, "shape") else np.asarray(A_uint8), device)
        rng = np.random.default_rng(seed)
        kp = max(k + oversampling, 20)
        t0 = time.time()
        Omega = rng.standard_normal(size=(M, kp), dtype=np.float32)
        to_dev = (lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)) if device is not None else (lambda x: torch.from_numpy(np.ascontiguousarray(x)))

        def A_times(B_np):          # [N,M] @ [M,kp] -> [N,kp]   (rsvd.pyx multiply_A_omega)
            B = to_dev(B_np)
            return torch.cat([src.chunk(s, min(N, s + rows)) @ B for s in range(0, N, rows)], dim=0).cpu().numpy()

        def QT_times_A(Q_np):       # [kp,N] @ [N,M] -> [kp,M]   (rsvd.pyx multiply_QT_A)
            QT = to_dev(Q_np)
            acc = None
            for s in range(0, N, rows):
                e = min(N, s + rows)
                part = QT[:, s:e] @ src.chunk(s, e)
                acc = part if acc is None else acc + part
            return acc.cpu().numpy()

        Y = A_times(Omega)
        for _ in range(power_iterations):
            Qy, _ = np.linalg.qr(Y, mode="reduced")
            Bt = QT_times_A(np.ascontiguousarray(Qy.T))
            Y = A_times(np.ascontiguousarray(Bt.T))
        Q, _ = np.linalg.qr(Y, mode="reduced")
        B = QT_times_A(np.ascontiguousarray(Q.T))
        Ut, St, Vt = np.linalg.svd(B, full_matrices=False)
        Vt = svd_flip(Vt, Ut)
        log.info(f"    Total time SVD: {time.time() - t0:.4f}s")
        return np.ascontiguousarray(Vt[:k, :].astype(np.float32))
    finally:
        torch.set_float32_matmul_precision(old_prec)
