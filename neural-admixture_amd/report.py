"""Post-training reports (not on the step path).  loglikelihood mirrors the reference's Cython float64
reduction (src/utils_c/utils.pyx:8-40) with torch float64 ops on the GPU, chunked by rows.  Rows come either from
the packed matrix already resident in HBM (decoded with nadm_unpack2bit, no PCIe traffic) or from a host tensor."""
import numpy as np
import torch

from ._lib import lib, check, ptr


def device_rows_from_packed(xp: torch.Tensor, M: int):
    """Row provider over a packed device matrix [rows, ld]: (s, e) -> uint8 [e-s, M] on the same device."""
    ld = xp.shape[1]

    def get(s, e):
        out = torch.empty((e - s, M), dtype=torch.uint8, device=xp.device)
        check(lib.nadm_unpack2bit(ptr(xp[s:e]), ptr(out), e - s, M, ld, None), "unpack2bit")
        return out
    return get


def host_rows(data_u8_cpu: torch.Tensor, dev):
    return lambda s, e: data_u8_cpu[s:e].to(dev)


def loglikelihood_rows(get_rows, N: int, P: np.ndarray, Q: np.ndarray, dev, eps: float = 1e-6, rows: int = 256) -> float:
    """sum over non-missing of g*log(rec) + (2-g)*log1p(-rec), rec = clip(Q_i.P_j, eps, 1-eps), g clipped to
    [eps, 2-eps], all float64 (utils.pyx:24-40)."""
    P64 = torch.as_tensor(P, dtype=torch.float64, device=dev)
    Q64 = torch.as_tensor(Q, dtype=torch.float64, device=dev)
    total = torch.zeros((), dtype=torch.float64, device=dev)
    for s in range(0, N, rows):
        e = min(N, s + rows)
        g = get_rows(s, e)
        rec = torch.clamp(Q64[s:e] @ P64.T, eps, 1.0 - eps)
        gd = torch.clamp(g.to(torch.float64), eps, 2.0 - eps)
        term = gd * torch.log(rec) + (2.0 - gd) * torch.log1p(-rec)
        total += torch.where(g != 3, term, torch.zeros_like(term)).sum()
    return float(total.item())


def loglikelihood_hip(xp: torch.Tensor, M: int, P: np.ndarray, Q: np.ndarray, eps: float = 1e-6) -> float:
    """The same float64 reduction in one HIP kernel over the resident packed matrix (nadm_loglik)."""
    dev = xp.device
    Pd = torch.as_tensor(np.ascontiguousarray(P, dtype=np.float32), device=dev)
    Qd = torch.as_tensor(np.ascontiguousarray(Q, dtype=np.float32), device=dev)
    K = Pd.shape[1]
    part = torch.empty(int(lib.nadm_loglik_blocks(M)), dtype=torch.float64, device=dev)
    check(lib.nadm_loglik(ptr(xp), xp.shape[1], xp.shape[0], M, ptr(Pd), ptr(Qd), K, K, float(eps), ptr(part),
                          torch.cuda.current_stream().cuda_stream), "loglik")
    return float(part.cpu().numpy().sum())


def loglikelihood_packed(engine, data_u8_cpu, P: np.ndarray, Q: np.ndarray, eps: float = 1e-6, rows: int = 256) -> float:
    """Single-GPU runs decode the rows from the resident packed matrix (engine rows are in sample order there);
    sharded runs fall back to the host copy, which is in sample order like Q."""
    N = Q.shape[0]
    # resident = the engine holds ALL samples in sample order AND all M SNPs.  A SnpShardedEngine holds every row but only
    # its own SNP slice (engine.M < P.shape[0]): its rows cannot be matched against the gathered [M_total, k] matrix.
    resident = (engine.xp is not None and engine.xp.shape[0] == N and not getattr(engine, "rows_are_sharded", False)
                and engine.M == P.shape[0] and engine.device.type == "cuda")
    if resident and P.shape[1] <= 16 and engine.device.type == "cuda":
        return loglikelihood_hip(engine.xp, engine.M, P, Q, eps)
    if resident:
        return loglikelihood_rows(device_rows_from_packed(engine.xp, engine.M), N, P, Q, engine.device, eps, rows)
    if hasattr(data_u8_cpu, "unpack_rows"):          # PackedGenotypes on the host
        get = lambda s, e: torch.from_numpy(data_u8_cpu.unpack_rows(s, e)).to(engine.device)
        return loglikelihood_rows(get, N, P, Q, engine.device, eps, rows)
    return loglikelihood_rows(host_rows(data_u8_cpu, engine.device), N, P, Q, engine.device, eps, rows)
