"""Post-training reports (not on the step path).  loglikelihood mirrors the reference's Cython
float64 reduction (src/utils_c/utils.pyx:8-40) with torch float64 ops on the GPU, chunked by rows."""
import numpy as np
import torch


def loglikelihood_packed(engine, data_u8_cpu: torch.Tensor, P: np.ndarray, Q: np.ndarray, eps: float = 1e-6, rows: int = 256) -> float:
    dev = engine.device
    P64 = torch.as_tensor(P, dtype=torch.float64, device=dev)
    Q64 = torch.as_tensor(Q, dtype=torch.float64, device=dev)
    total = torch.zeros((), dtype=torch.float64, device=dev)
    N = data_u8_cpu.shape[0]
    for s in range(0, N, rows):
        g = data_u8_cpu[s:s + rows].to(dev)
        rec = torch.clamp(Q64[s:s + rows] @ P64.T, eps, 1.0 - eps)
        gd = torch.clamp(g.to(torch.float64), eps, 2.0 - eps)
        term = gd * torch.log(rec) + (2.0 - gd) * torch.log1p(-rec)
        total += torch.where(g != 3, term, torch.zeros_like(term)).sum()
    return float(total.item())
