"""The training step restated in the REFERENCE'S OWN SHAPE: the same sequence of torch CPU operators the reference runs per batch
(uint8 -> float / 2 -> where(1.5 -> 0), X @ V, rms_norm, Linear + ReLU, per-head Linear + softmax, Q @ P^T, clamp, BCELoss(sum),
autograd backward, fused Adam(0.9, 0.95), clamp of P: neural_admixture.py:169-176, :83-98, :288, :410-412, :187-204), forced to true
fp32 matmuls.  TEST INFRASTRUCTURE: the second, reference-shaped CPU baseline of bench.py (`cpu_baseline_reference_shaped`, what
SURVEY 8d / BASELINE.md section 3 ask for beside the fused C/OpenMP port of nadm_oracle_c.c) and one more checker, pinned against
the fixtures captured from the reference (tests/test_oracle_golden.py).  Never imported by the product."""
from typing import List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


class TorchShapedModel:
    def __init__(self, V_MC: np.ndarray, P_list: Sequence[np.ndarray], g, W1, b1, Wk: Sequence[np.ndarray], bk: Sequence[np.ndarray], lr: float):
        t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), requires_grad=True)
        self.V, self.g, self.W1, self.b1 = t(V_MC), t(g), t(W1), t(b1)
        self.Wk, self.bk = [t(w) for w in Wk], [t(b) for b in bk]
        self.P = [t(p) for p in P_list]                               # [M, k] each (the decoder's logical weight, neural_admixture.py:73-74)
        groups = [{"params": self.Wk + self.bk}, {"params": [self.W1, self.b1]}, {"params": [self.g]}, {"params": [self.V]}, {"params": self.P}]
        self.opt = torch.optim.Adam(groups, lr=lr, betas=(0.9, 0.95), fused=True)        # :187-204
        self.last_Q: List[torch.Tensor] = []

    def forward_loss(self, G_u8: torch.Tensor) -> torch.Tensor:
        X = G_u8.float() / 2
        X = torch.where(X == 1.5, 0.0, X)                              # :169-170
        Z = X @ self.V                                                 # :172
        Zn = F.rms_norm(Z, (Z.shape[1],), self.g, 1e-8)                # :135,173
        H = torch.relu(Zn @ self.W1.T + self.b1)                       # :138-140,174
        loss = 0.0
        self.last_Q = []
        for Wk, bk, P in zip(self.Wk, self.bk, self.P):
            Q = torch.softmax(H @ Wk.T + bk, dim=1)                    # :29,48,176
            R = torch.clamp(Q @ P.T, 0.0, 1.0)                         # :94-97
            loss = loss + F.binary_cross_entropy(R, X, reduction="sum")   # :288,431
            self.last_Q.append(Q.detach())
        return loss

    def step(self, G_u8: torch.Tensor) -> float:
        self.opt.zero_grad(set_to_none=True)
        loss = self.forward_loss(G_u8)
        loss.backward()                                                # :410
        self.opt.step()                                                # :411
        with torch.no_grad():
            for P in self.P:
                P.clamp_(0.0, 1.0)                                     # restrict_P, :179-185,412
        return float(loss.item())
