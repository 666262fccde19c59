"""Randomized SVD initialisation of V (8(f)-2), same algorithm as the reference's ``RSVD`` (src/svd.py:39-83):
Omega ~ N(0,1) [M,k'] from ``np.random.default_rng(seed)``, k' = max(k+10, 20); Y = A.Omega; 2 power iterations
with QR; QR; B = Q^T.A; SVD(B); sign flip (svd.py:16-37); returns Vt[:k] [k,M] float32.  A = the RAW genotype
codes (0,1,2,3 with missing kept as 3, no centering), exactly like the reference.

The two tall-skinny products that the reference runs as naive Cython triple loops on the CPU
(src/utils_c/rsvd.pyx:16-50) run here on the GPU straight from the 2-bit packed matrix with the pass-1 / pass-3
matrix-core kernels (nadm_pca_project / nadm_pca_project_t: genotype/2 with missing = 1.5, i.e. half the raw code,
fp32-exact bf16 splitting), eight of the k' columns per launch.  What is left is k' = 20 columns wide: the QRs of [N, k'] run on
the host with the reference's own numpy call, the SVD of the wide B comes from the Cholesky factor of its float64 Gram matrix --
no device BLAS / solver call anywhere (r05: their first-use cost, ~0.2 s each, was most of this function).  Without a GPU the
same algorithm runs through torch/numpy on the host (unpacked chunks)."""
from __future__ import annotations

import logging
import time

import numpy as np
import torch

log = logging.getLogger(__name__)


def svd_flip(V: np.ndarray, U: np.ndarray) -> np.ndarray:
    idx = np.argmax(np.abs(U), axis=0)
    signs = np.sign(U[idx, np.arange(U.shape[1])])
    return V * signs[:, np.newaxis]


class _Rows:
    """Access to the raw code matrix A [N,M] from uint8 [N,M] or io.PackedGenotypes: resident 2-bit packed in HBM when a
    GPU is given (products through the HIP kernels), float32 row chunks on the host otherwise."""

    def __init__(self, data, device):
        self.dev = device
        self.packed = hasattr(data, "packed")
        self.N, self.M = data.shape
        self.data = data
        self.xp = None
        if device is not None and device.type == "cuda":
            from .io import packed_chunks
            from .layout import ModelLayout
            ld = ModelLayout.row_stride(self.M)
            if self.packed:
                self.xp = data.packed.to(device)
            else:
                self.xp = torch.empty((self.N, ld), dtype=torch.uint8, device=device)
                for s, e, pk in packed_chunks(data, ld, 8192):
                    self.xp[s:e].copy_(pk)
            self.ld = ld

    def chunk(self, s, e):
        a = self.data.unpack_rows(s, e) if self.packed else np.asarray(self.data[s:e])
        return torch.from_numpy(np.ascontiguousarray(a)).float()

    # ---- GPU products (A = 2 * (G/2 with missing 1.5)) ----
    def a_times(self, B_np, rows: int = 8192, keep_on_device: bool = False):
        """[N,M] @ [M,kp] -> [N,kp]   (B: numpy array or a tensor already on the device)"""
        from ._lib import lib, check, ptr
        N, M, dev = self.N, self.M, self.dev
        kp = B_np.shape[1]
        chunks = int(lib.nadm_encode_chunks(M))
        rb = min(N, rows)
        zpart = torch.empty(chunks * rb * 8, dtype=torch.float32, device=dev)
        fold = torch.empty(rb * 8, dtype=torch.float32, device=dev)       # the chunks' partial sums folded in a fixed order (nadm_sum_rows)
        idx = torch.arange(N, dtype=torch.int32, device=dev)
        Bd = B_np if torch.is_tensor(B_np) else torch.from_numpy(np.ascontiguousarray(B_np, dtype=np.float32)).to(dev)
        out = torch.empty((N, kp), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for g in range(0, kp, 8):
            w = min(8, kp - g)
            Vd = torch.zeros((M, 8), dtype=torch.float32, device=dev)
            Vd[:, :w] = Bd[:, g:g + w]
            for s in range(0, N, rb):
                e = min(N, s + rb)
                check(lib.nadm_pca_project(ptr(self.xp), self.ld, ptr(idx[s:e]), e - s, M, ptr(Vd), 8, ptr(zpart), st), "pca_project")
                check(lib.nadm_sum_rows(ptr(zpart), chunks, (e - s) * 8, ptr(fold), st), "sum_rows")
                out[s:e, g:g + w] = fold[: (e - s) * 8].view(e - s, 8)[:, :w]
        out *= 2.0
        return out if keep_on_device else out.cpu().numpy()

    def qt_times(self, Q_nk, keep_on_device: bool = False):
        """Q [N,kp] (numpy or device tensor) -> Q^T @ A as its TRANSPOSE [M,kp] when kept on the device (the layout the next
        product consumes), as [kp,M] numpy otherwise."""
        from ._lib import lib, check, ptr
        N, M, dev = self.N, self.M, self.dev
        kp = Q_nk.shape[1]
        idx = torch.arange(N, dtype=torch.int32, device=dev)
        Qd = Q_nk if torch.is_tensor(Q_nk) else torch.from_numpy(np.ascontiguousarray(Q_nk, dtype=np.float32)).to(dev)      # [N,kp]
        out = torch.empty((M, kp), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for g in range(0, kp, 8):
            w = min(8, kp - g)
            Y = torch.zeros((N, 8), dtype=torch.float32, device=dev)
            Y[:, :w] = Qd[:, g:g + w]
            o = torch.empty((M, 8), dtype=torch.float32, device=dev)
            yimg = torch.empty(int(lib.nadm_dz_image_bytes(N)), dtype=torch.uint8, device=dev)     # operand image of Y (include/nadm.h)
            check(lib.nadm_dz_image(ptr(Y), N, 8, ptr(yimg), st), "dz_image")
            check(lib.nadm_pca_project_t(ptr(self.xp), self.ld, ptr(idx), N, M, ptr(Y), ptr(yimg), 8, ptr(o), st), "pca_project_t")
            out[:, g:g + w] = o[:, :w]
        out *= 2.0
        return out if keep_on_device else np.ascontiguousarray(out.t().cpu().numpy())


def gram64(A: torch.Tensor, chunk: int = 32768) -> torch.Tensor:
    """A^T A in float64 for a tall, narrow A [n, c] (c <= ~32) on its own device, from elementwise products and sums -- no BLAS call.
    The first GEMM / solver call of a process loads hipBLASLt / rocBLAS / rocSOLVER: 0.15-0.2 s each on this stack, a third of a
    default run on a 1000-Genomes-sized matrix (profiles/r05_init_profile_before.txt), for products a few reductions do in a millisecond."""
    n, c = A.shape
    G = torch.zeros((c, c), dtype=torch.float64, device=A.device)
    for s in range(0, n, chunk):
        blk = A[s:s + chunk].to(torch.float64)
        G += (blk[:, :, None] * blk[:, None, :]).sum(dim=0)
    return G


def _omega_to_device(rng, M: int, kp: int, device: torch.device, chunk: int = 32768, after=None) -> torch.Tensor:
    """Omega [M, k'] ~ N(0, 1) float32 from the reference's own stream (``rng.standard_normal(size=(M, kp), dtype=float32)``,
    src/svd.py:47-48) straight into HBM: generated ``chunk`` rows at a time into two small pinned buffers that are copied while the
    next chunk is drawn.  Chunked draws continue the generator's stream exactly (checked by tests/test_abi_and_host.py); what
    changes is where the time goes: one call into a fresh 48 MB host array (M = 600k) takes 0.25-0.36 s, two thirds of it page
    faults; the same numbers through a 2.6 MB ring take 0.15 s, underneath which the copies run (r05, profiles/r05_init_profile_*.txt)."""
    out = torch.empty((M, kp), dtype=torch.float32, device=device)
    ring = [torch.empty((min(chunk, M), kp), dtype=torch.float32).pin_memory() for _ in range(2)]
    done = [None, None]
    with torch.cuda.device(device):
        if after is not None:                                # called on a helper thread (its own current stream): `out` may be a recycled block
            torch.cuda.current_stream().wait_stream(after)   # whose last use is still queued on the caller's stream
        for i, s in enumerate(range(0, M, chunk)):
            e = min(M, s + chunk)
            if done[i & 1] is not None:
                done[i & 1].synchronize()                    # the copy that last read this buffer
            rng.standard_normal(dtype=np.float32, out=ring[i & 1].numpy()[: e - s])
            out[s:e].copy_(ring[i & 1][: e - s], non_blocking=True)
            done[i & 1] = torch.cuda.Event()
            done[i & 1].record()
        for ev in done:                                      # Omega is complete in HBM when this returns, whatever stream the caller is on
            if ev is not None:
                ev.synchronize()
    return out


def _preload_mixture_library(N: int) -> None:
    """The decoder-init mixture fit that follows the SVD (train.gmm_p_init) calls scikit-learn for N <= 20000, and importing
    scikit-learn takes 1.0 s on this image -- half of a default run on a 1000-Genomes-sized matrix.  Started here, on a thread,
    the import runs underneath the SVD's GPU work (the main thread mostly waits for the device)."""
    if N > 20_000:
        return
    import threading

    def _imp():
        try:
            import sklearn.mixture  # noqa: F401
        except Exception:                                        # the fit itself will report what is wrong
            pass
    # not a daemon: a run that ends (or fails) early waits for the import to finish instead of tearing the interpreter down under it
    threading.Thread(target=_imp, name="nadm-preload-sklearn", daemon=False).start()


def RSVD(A_uint8, N: int, M: int, k: int = 8, seed: int = 42, oversampling: int = 10, power_iterations: int = 2,
         device: torch.device = None, rows: int = 2048, preload_mixture: bool = False, phases: dict = None) -> np.ndarray:
    """``phases``: a dict that receives the wall-clock of the GPU path's stages (the device is synchronised at every boundary:
    for profiling, tools/full_run.py)."""
    if device is None:
        device = torch.device("cuda:0") if torch.cuda.is_available() else None
    if preload_mixture:               # (only callers that will fit with the library itself, train.gmm_p_init(fit="sklearn"), ask for it)
        _preload_mixture_library(N)
    old_prec = torch.get_float32_matmul_precision()
    torch.set_float32_matmul_precision("highest")
    try:
        rng = np.random.default_rng(seed)
        kp = max(k + oversampling, 20)
        t0 = time.time()
        on_gpu = device is not None and device.type == "cuda"
        t_mark = [t0]

        def mark(name):
            if phases is not None:
                if on_gpu:
                    torch.cuda.synchronize(device)
                now = time.time()
                phases[name] = phases.get(name, 0.0) + now - t_mark[0]
                t_mark[0] = now
        if on_gpu:
            # Omega is drawn (numpy releases the GIL) and shipped on a helper thread while this one moves the packed matrix to HBM --
            # the draw is the longest single item of the GPU path
            import threading
            box = {}
            caller_stream = torch.cuda.current_stream(device)

            def _draw():
                try:
                    box["omega"] = _omega_to_device(rng, M, kp, device, after=caller_stream)
                except BaseException as e:                   # re-raised on the caller's thread
                    box["error"] = e
            th = threading.Thread(target=_draw, name="nadm-rsvd-omega")
            th.start()
        src = _Rows(A_uint8 if hasattr(A_uint8, "shape") else np.asarray(A_uint8), device)
        mark("rows_to_hbm")
        if on_gpu:
            th.join()
            if "error" in box:
                raise box["error"]
            Omega = box["omega"]
            mark("omega_wait")
        else:
            Omega = rng.standard_normal(size=(M, kp), dtype=np.float32)

        if src.xp is not None:
            # GPU path: the two tall products run on the HIP kernels and stay in HBM; everything else is k' = 20 columns wide and is
            # done WITHOUT the device's BLAS / solver libraries (their first call costs more than this whole function, gram64):
            #   * the QR of Y [N, k'] on the host with numpy's LAPACK -- the reference's own call (src/svd.py:56,64), so Q carries its
            #     sign convention; Y is 200 KB at N = 2504, 8 MB at 100k
            #   * the SVD of the wide B [k', M] from the Cholesky factor of its float64 Gram matrix: B^T = Q2 R2 with R2 = chol(B B^T)^T
            #     -> svd(R2^T) = U S W^T gives B = U S (Q2 W)^T, i.e. Vt = W^T R2^-T B.  (Which orthonormal factor of B^T is used does
            #     not touch U, so the sign rule below sees what the reference's svd(B) sees; kappa(B)^2 ~ 1e4-1e5 is nothing in float64.)
            def host_qr(Yd):
                q = np.linalg.qr(Yd.cpu().numpy(), mode="reduced")[0]
                return torch.from_numpy(np.ascontiguousarray(q)).to(device)
            Y = src.a_times(Omega, keep_on_device=True)
            mark("products")
            for _ in range(power_iterations):
                Qy = host_qr(Y)
                mark("qr_small")
                Bt = src.qt_times(Qy, keep_on_device=True)                 # [M,k']
                Y = src.a_times(Bt, keep_on_device=True)
                mark("products")
            Q = host_qr(Y)
            mark("qr_small")
            Bt = src.qt_times(Q, keep_on_device=True)                      # B^T [M,k']
            mark("products")
            try:
                R2 = np.linalg.cholesky(gram64(Bt).cpu().numpy()).T        # upper triangular, B^T = Q2 R2
            except np.linalg.LinAlgError:
                # B without full row rank (fewer samples than k', constant data): no Cholesky factor -- the reference's own last two lines
                # on the host (src/svd.py:79-82; 96 MB at M = 600k, rare)
                Ut, St, Vt_h = np.linalg.svd(Bt.cpu().numpy().astype(np.float64).T, full_matrices=False)
                log.info(f"    Total time SVD: {time.time() - t0:.4f}s")
                return np.ascontiguousarray(svd_flip(Vt_h, Ut)[:k].astype(np.float32))
            Ut, St, Wt = np.linalg.svd(R2.T, full_matrices=False)
            signs = np.sign(Ut[np.argmax(np.abs(Ut), axis=0), np.arange(Ut.shape[1])])     # svd_flip on U (svd.py:16-37)
            T = (signs[:, None] * np.linalg.solve(R2, Wt.T).T)[:k]          # rows of W^T R2^-T, sign-flipped: Vt = T B
            Td = torch.from_numpy(np.ascontiguousarray(T)).to(device)       # float64 [k, k']
            Bt64 = Bt.to(torch.float64)
            Vt = torch.stack([(Bt64 * Td[i]).sum(dim=1) for i in range(k)]).to(torch.float32).cpu().numpy()
            mark("svd_and_result")
            log.info(f"    Total time SVD: {time.time() - t0:.4f}s")
            return np.ascontiguousarray(Vt.astype(np.float32))

        def A_times(B_np):          # [N,M] @ [M,kp] -> [N,kp]   (rsvd.pyx multiply_A_omega)
            B = torch.from_numpy(np.ascontiguousarray(B_np))
            return torch.cat([src.chunk(s, min(N, s + rows)) @ B for s in range(0, N, rows)], dim=0).numpy()

        def QT_times_A(Q_np):       # [kp,N] @ [N,M] -> [kp,M]   (rsvd.pyx multiply_QT_A)
            QT = torch.from_numpy(np.ascontiguousarray(Q_np))
            acc = None
            for s in range(0, N, rows):
                e = min(N, s + rows)
                part = QT[:, s:e] @ src.chunk(s, e)
                acc = part if acc is None else acc + part
            return acc.numpy()

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
