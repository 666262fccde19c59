"""Drop-in for ``neural_admixture.model.train.train`` (train.py:19-149): same positional signature,
same returns ``(Ps, Qs, model)``.  The PCA-space GMM decoder init is the reference's scikit-learn call for
1000-Genomes-sized inputs and its float64 restatement in device ops (_gmm_em.py, same means to 1e-13) for N > 20000;
packing, the step loop and the final-Q pass run on the MI355X engine."""
from __future__ import annotations

import contextlib
import logging
import sys
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from .model import NeuralAdmixture
from .report import loglikelihood_packed

logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(message)s")
log = logging.getLogger(__name__)


def pca_project_gpu(data, V_CM: np.ndarray, device: torch.device, chunk_rows: int = 4096) -> np.ndarray:
    """X_pca [N, C] float32 = (G/2) @ V.T with a missing call counted as 1.5 (train.py:49-55), computed by the pass-1
    kernel (nadm_pca_project) from 2-bit packed rows streamed to the GPU ``chunk_rows`` at a time."""
    from ._lib import lib, check, ptr
    from .layout import ModelLayout
    from .io import packed_chunks
    C_, M = V_CM.shape
    N = data.shape[0]
    ld, CP = ModelLayout.row_stride(M), 8
    Vd = torch.zeros((M, CP), dtype=torch.float32, device=device)
    Vd[:, :C_] = torch.as_tensor(np.ascontiguousarray(V_CM.T), dtype=torch.float32).to(device)
    chunks = int(lib.nadm_encode_chunks(M))
    zpart = torch.empty(chunks * min(N, chunk_rows) * CP, dtype=torch.float32, device=device)
    fold = torch.empty(min(N, chunk_rows) * CP, dtype=torch.float32, device=device)     # chunk partials folded in a fixed order
    idx = torch.arange(min(N, chunk_rows), dtype=torch.int32, device=device)
    out = np.empty((N, C_), dtype=np.float32)
    st = torch.cuda.current_stream().cuda_stream
    for s, e, pk in packed_chunks(data, ld, chunk_rows):
        xp = pk.to(device)
        check(lib.nadm_pca_project(ptr(xp), ld, ptr(idx), e - s, M, ptr(Vd), CP, ptr(zpart), st), "pca_project")
        check(lib.nadm_sum_rows(ptr(zpart), chunks, (e - s) * CP, ptr(fold), st), "sum_rows")
        out[s:e] = fold[: (e - s) * CP].view(e - s, CP)[:, :C_].cpu().numpy()
    return out


def gmm_p_init(data_np, V_CM: np.ndarray, K: Optional[int], min_k, max_k, n_components: int, seed: int,
               device: Optional[torch.device] = None, fit: str = "auto") -> np.ndarray:
    """P_init [sum(ks), M] = clip(GMM means @ V, 5e-6, 1-5e-6) in the PCA subspace (train.py:49-68).
    Note the projection keeps missing (3) as 1.5, exactly like the reference (train.py:52).
    ``data_np``: uint8 [N,M] array or an io.PackedGenotypes.  With a GPU ``device`` and n_components <= 8 the
    projection runs on the GPU (pca_project_gpu); otherwise on the host, 1024 rows at a time like the reference.
    The mixture fit (``fit``): "sklearn" = the reference's scikit-learn call, "em" = its float64 restatement in device ops
    (_gmm_em.py), "native" = the same on host threads (gmm.py + csrc/nadm_gmm.cpp), "device" = the same with the sums over the samples
    in HIP kernels (csrc/nadm_gmm_dev.hip: 8 components, K <= 16), "auto" = "device" where it applies and pays (a GPU run with
    N >= 20000), else "native" -- all give the library's means to 1e-10 (the device form to 1e-9)."""
    N = data_np.shape[0]
    if device is not None and device.type == "cuda" and n_components <= 8:
        X_pca = pca_project_gpu(data_np, V_CM, device)
    else:
        rows = data_np.unpack_rows if hasattr(data_np, "unpack_rows") else (lambda s, e: data_np[s:e])
        X_pca = np.zeros((N, n_components), dtype=np.float32)
        for i in range(0, N, 1024):
            X_pca[i:i + 1024] = (rows(i, min(N, i + 1024)).astype(np.float32) / 2) @ V_CM.T
    X_pca = X_pca.astype("float64")
    ks = [K] if K is not None else list(range(min_k, max_k + 1))
    how = fit
    from . import gmm as _gmm
    # "auto" / "native": the library's algorithm restated on the host (gmm.py + csrc/nadm_gmm.cpp: same seeding draws, same EM, means equal
    # to 1e-10 -- tests/test_abi_and_host.py): ~0.05 s at N = 2504 where the library takes 0.55 s + a 1.0 s import, ~0.2 s at N = 100k
    # (restarts and sample ranges on threads) where it takes 22-45 s; several K run concurrently.  "em": the same in float64 device ops
    # (_gmm_em.py; pays the device BLAS's first-use cost).  "sklearn": the library itself, several K as concurrent child processes.
    if how == "em":
        from ._gmm_em import fit_means as fit_means_device
        means = [fit_means_device(X_pca, k, seed, device) for k in ks]
    elif how == "device" or (how == "auto" and device is not None and device.type == "cuda"
                             and all(_gmm.device_form_applies(N, n_components, k) for k in ks)):
        with torch.cuda.device(device):                            # (a few launches per iteration: the K of a multi-head run one after the other)
            means = [_gmm.fit_means(X_pca, k, seed, stream=torch.cuda.current_stream().cuda_stream) for k in ks]
    elif how in ("auto", "native"):
        from .gmm import fit_means as fit_means_host
        if len(ks) > 1:
            from concurrent.futures import ThreadPoolExecutor     # (the C call releases the GIL; each fit runs its restarts on threads of its own)
            # (each fit sizes its restart x worker threads for the whole host, csrc/nadm_gmm.cpp: two at a time keep K = 2..10 from
            # putting several hundred spinning threads on it -- ADVICE r05)
            with ThreadPoolExecutor(max_workers=min(2, len(ks))) as pool:
                means = list(pool.map(lambda k_: fit_means_host(X_pca, k_, seed), ks))
        else:
            means = [fit_means_host(X_pca, ks[0], seed)]
    else:
        means = _gmm_means_parallel(X_pca, ks, seed) if len(ks) > 2 else None
        if means is None:
            from ._gmm_fit import fit_means
            means = [fit_means(X_pca, k, seed) for k in ks]
    return np.concatenate([np.clip(m @ V_CM, 5e-6, 1 - 5e-6) for m in means], axis=0)


def _gmm_means_parallel(X_pca: np.ndarray, ks, seed: int):
    """The multi-head run fits one GMM per K (train.py:65-67, a list comprehension in the reference).  The fits are
    independent and deterministic (random_state), so they run as concurrent child processes (_gmm_fit.py as a script:
    numpy + sklearn only, no fork of the process that owns the GPU context).  Returns None on any failure -- the caller
    then fits sequentially in-process, which gives the same numbers."""
    import os
    import subprocess
    import tempfile
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gmm_fit.py")
    procs = []
    try:
        with tempfile.TemporaryDirectory() as td:
            xp = os.path.join(td, "x.npy")
            np.save(xp, X_pca)
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            procs = [subprocess.Popen([sys.executable, script, xp, str(k), str(seed), os.path.join(td, f"m{k}.npy")], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for k in ks]
            codes = [p.wait(timeout=600) for p in procs]
            if any(codes):
                log.info(f"    concurrent GMM fits failed (exit codes {codes}); fitting in-process instead")
                return None
            return [np.load(os.path.join(td, f"m{k}.npy")) for k in ks]
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        log.info(f"    concurrent GMM fits failed ({type(e).__name__}: {e}); fitting in-process instead")
        return None
    finally:
        for p in procs:                         # no orphans: a timed-out / failed batch must not keep running beside the fallback
            if p.poll() is None:
                p.kill()
                p.wait()


def supervised_init(data_np, pops, K: int):
    """Supervised mode (train.py:74-83): population names -> class indices in sorted order, and
    P_init [K,M] = per-class mean of the RAW codes (0,1,2, missing 3; not halved, not clipped -- the first
    restrict_P brings it into [0,1]).  ``data_np``: uint8 [N,M] or an io.PackedGenotypes."""
    names = sorted(np.unique([a for a in pops]))
    if K is None or len(names) != K:
        raise AssertionError(f"Number of ancestries in training ground truth ({len(names)}) is not equal to the value of K ({K})")
    lut = {a: i for i, a in enumerate(names)}
    y = np.asarray([lut[a] for a in pops], dtype=np.int64)
    N, M = data_np.shape
    if len(y) != N:
        raise RuntimeError("pops must hold one population label per sample")
    rows = data_np.unpack_rows if hasattr(data_np, "unpack_rows") else (lambda s, e: data_np[s:e])
    sums = np.zeros((K, M), dtype=np.float64)
    for i in range(0, N, 1024):
        blk, yb = rows(i, min(N, i + 1024)), y[i:i + 1024]
        for k in range(K):
            if np.any(yb == k):
                sums[k] += blk[yb == k].sum(axis=0, dtype=np.float64)
    P = (sums / np.bincount(y, minlength=K)[:, None]).astype(np.float32)
    return y, P


@contextlib.contextmanager
def capped_host_threads(limit: int = 4):
    """Caps torch's intra-op pool and the BLAS / OpenMP pools (threadpoolctl) at ``limit`` threads for the duration of the
    block, never raising a pool that is smaller already.  The host side of a run is a launch loop plus a few small numpy /
    scikit-learn calls; with the default pools of a large host (256 threads here, shared with other jobs) their barriers
    spin long enough to stall the HIP runtime's own threads: a full default run on one MI355X takes 2.5 / 5.1 / 14.3-17 s
    (2504 x 600k K=7 / K=2..10 / 100k x 500k K=8) with the default pools and 1.9 / 4.8 / 13.1 s with them capped, the
    mixture fit alone 1.6 -> 1.0 s.  The reference's CLI caps them with --threads (entry.py:138-146); this makes the
    boundary function behave the same when it is called directly."""
    prev = torch.get_num_threads()
    torch.set_num_threads(max(1, min(prev, limit)))
    limits = None
    try:
        from threadpoolctl import ThreadpoolController, threadpool_limits
        per_api = {}
        for lib_info in ThreadpoolController().info():
            api, n = lib_info.get("user_api"), lib_info.get("num_threads")
            if api and n:
                per_api[api] = min(per_api.get(api, limit), n, limit)
        limits = threadpool_limits(limits=per_api) if per_api else None
    except ImportError:                          # threadpoolctl comes with scikit-learn; without it only torch's pool is capped
        pass
    try:
        yield
    finally:
        if limits is not None:
            limits.restore_original_limits()
        torch.set_num_threads(prev)


def train(epochs: int, batch_size: int, learning_rate: float, K: int, seed: int, data: torch.Tensor, device: torch.device,
          num_gpus: int, hidden_size: int, master: bool, V: np.ndarray, pops, min_k: int = None, max_k: int = None,
          n_components: int = None, *, parallelism: str = "dp", host_threads: int = 4, gmm: str = "auto"):
    """The reference's boundary function (see the module docstring and _train).  The host thread pools (torch intra-op, BLAS,
    OpenMP) are capped at ``host_threads`` while it runs -- the reference's CLI does that with --threads (entry.py:138-146), and
    this package's CLI passes its --threads here; the epoch loop itself always runs with torch's pool at 1 (model.py).
    ``gmm`` (keyword, CLI --gmm): who fits the decoder-init mixture (gmm_p_init's ``fit``): "sklearn" is the reference's own call."""
    with capped_host_threads(max(1, int(host_threads))):
        return _train(epochs, batch_size, learning_rate, K, seed, data, device, num_gpus, hidden_size, master, V, pops, min_k, max_k,
                      n_components, parallelism=parallelism, gmm=gmm)


def _train(epochs: int, batch_size: int, learning_rate: float, K: int, seed: int, data: torch.Tensor, device: torch.device,
           num_gpus: int, hidden_size: int, master: bool, V: np.ndarray, pops, min_k: int = None, max_k: int = None,
           n_components: int = None, *, parallelism: str = "dp", gmm: str = "auto"):
    """See module docstring.  ``data`` uint8 [N,M] CPU tensor (or an ``io.PackedGenotypes``, e.g. from
    ``io.read_bed_packed``); ``V`` numpy [C,M] (RSVD output,
    svd.py:83); returns Ps (list of [M,k] float32), Qs (list of [N,k] float32), model.
    ``parallelism`` (keyword, not in the reference): "dp" = samples sharded over the GPUs, gradients summed over ranks like the reference's DDP (here: reduce-scatter,
    optimizer on the rank's slice, all-gather); "snp" = SNPs sharded (snp_parallel.py): same trajectory up to summation order, two tiny all-reduces
    per step instead of the 4*M*(C+S)-byte one."""
    eng_cls = NeuralAdmixture.engine_snp_cls if parallelism == "snp" else NeuralAdmixture.engine_cls
    if not eng_cls.supports(device):
        raise RuntimeError("neural_admixture_amd.train requires a ROCm GPU device; the CPU path is the reference's own")
    N, M = data.shape
    if n_components is None:
        n_components = V.shape[0]
    total_K = K if K is not None else sum(range(min_k, max_k + 1))
    y_num = None
    if master and pops is None:
        log.info("")
        log.info("    Running Gaussian Mixture in PCA subspace...")
        log.info("")
        P = gmm_p_init(data if hasattr(data, "unpack_rows") else data.numpy(), V, K, min_k, max_k, n_components, seed, device, fit=gmm)
    elif master:
        log.info("")
        log.info("    Running Supervised Mode...")
        log.info("")
        y_num, P = supervised_init(data if hasattr(data, "unpack_rows") else data.numpy(), pops, K)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if num_gpus > 1 and dist.is_available() and dist.is_initialized():
        if master:
            P_init = torch.as_tensor(P, dtype=torch.float32, device=device).contiguous()
            Vt = torch.as_tensor(V.T, dtype=torch.float32, device=device).contiguous()
            log.info("    Broadcasting to all GPUs...")
            if pops is not None:
                pops = torch.as_tensor(y_num, dtype=torch.int64, device=device)
        else:
            P_init = torch.empty((total_K, M), dtype=torch.float32, device=device)
            Vt = torch.empty((M, n_components), dtype=torch.float32, device=device)
            if pops is not None:
                pops = torch.empty(len(pops), dtype=torch.int64, device=device)
        if pops is not None:
            dist.broadcast(pops, src=0)        # train.py:107-108
        dist.broadcast(P_init, src=0)          # train.py:109
        dist.broadcast(Vt, src=0)              # train.py:110
        dist.barrier()
        if master:
            log.info("    Finished broadcasting!")
    else:
        P_init = torch.as_tensor(P, dtype=torch.float32).contiguous()
        Vt = torch.as_tensor(V.T, dtype=torch.float32).contiguous()
        if pops is not None:
            pops = torch.as_tensor(y_num, dtype=torch.int64)

    model = NeuralAdmixture(K, epochs, batch_size, learning_rate, device, seed, num_gpus, master, None, min_k, max_k,
                            parallelism=parallelism)
    Qs, Ps, raw = model.launch_training(P_init, data, hidden_size, Vt.shape[1], Vt, M, N, pops)

    if master:
        ks = [K] if K is not None else list(range(min_k, max_k + 1))
        for i, k in enumerate(ks):
            logl = model.logliks[i] if model.logliks is not None else loglikelihood_packed(model.engine, data, Ps[i], Qs[i])
            if K is not None:
                log.info(f"    Log-likelihood: {logl:2f}.")
            else:
                log.info(f"    Log-likelihood for K={k}: {logl:2f}.")
    return Ps, Qs, raw
