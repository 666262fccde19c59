"""CPU oracle for the Neural ADMIXTURE training hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy (float32) restatement of the algorithm the reference runs through
PyTorch autograd for one training step / one run.  It is the *checker* for the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``neural_admixture_amd``) never imports anything from ``oracle/``.

Parity pinning: the reference's own tests pin nothing for this path (one placeholder test,
``neural_admixture/tests/test_placeholder.py:1``), so this oracle is pinned against outputs of the
reference itself, run in the build container and committed as ``tests/golden/*.npz`` by
``tests/golden/make_golden.py`` (``tests/test_oracle_golden.py`` checks every fixture).

Every function cites the reference lines (relative to /root/reference) that it restates.
torch is used only for RNG parity (``torch.randperm`` / ``torch.manual_seed`` streams, which the
reference consumes through ``RandomSampler`` / ``DistributedSampler`` / ``nn.Linear`` init).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

F32 = np.float32
BCE_EPS = F32(1e-12)      # ATen binary_cross_entropy_backward EPSILON (behind neural_admixture.py:288,410)
LOG_CLAMP = F32(-100.0)   # ATen binary_cross_entropy clamps log terms at >= -100
RMS_EPS = 1e-8            # torch.nn.RMSNorm(eps=1e-8), neural_admixture.py:135
BETA1, BETA2, ADAM_EPS = 0.9, 0.95, 1e-8   # neural_admixture.py:204 (torch.optim.Adam defaults eps)


class precision64:
    """Context manager: run the restatement in float64 (tests use it as a rounding-free yardstick to
    tell accumulated fp32 rounding from a real discrepancy).  Not used for the golden comparisons."""

    def __enter__(self):
        g = globals()
        self._saved = (g["F32"], g["BCE_EPS"], g["LOG_CLAMP"])
        g["F32"] = np.float64
        g["BCE_EPS"] = np.float64(1e-12)
        g["LOG_CLAMP"] = np.float64(-100.0)
        return self

    def __exit__(self, *exc):
        g = globals()
        g["F32"], g["BCE_EPS"], g["LOG_CLAMP"] = self._saved
        return False


# --------------------------------------------------------------------------------------------
# 2-bit packing  (src/utils_c/pack2bit.cu:10-36 pack, :38-62 unpack)
# --------------------------------------------------------------------------------------------
def pack2bit(G: np.ndarray) -> np.ndarray:
    """uint8 [N,M] genotypes -> uint8 [N, ceil(M/4)].  SNP 4c+i lives in bits [2i, 2i+1] of byte c,
    only the two low bits of each input byte are kept, tail bits are zero (pack2bit.cu:26-31)."""
    G = np.asarray(G, dtype=np.uint8)
    N, M = G.shape
    Mp = (M + 3) // 4
    pad = np.zeros((N, Mp * 4), dtype=np.uint8)
    pad[:, :M] = G & 3
    q = pad.reshape(N, Mp, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)


def unpack2bit(Pk: np.ndarray, M: int) -> np.ndarray:
    """Inverse of :func:`pack2bit` (pack2bit.cu:52-61)."""
    Pk = np.asarray(Pk, dtype=np.uint8)
    N, Mp = Pk.shape
    out = np.empty((N, Mp, 4), dtype=np.uint8)
    for i in range(4):
        out[:, :, i] = (Pk >> (2 * i)) & 3
    return out.reshape(N, Mp * 4)[:, :M].copy()


def decode_x(G: np.ndarray) -> np.ndarray:
    """X = G/2 with missing (3 -> 1.5) replaced by 0 (neural_admixture.py:169-170)."""
    X = G.astype(F32) / F32(2)
    X[G == 3] = 0
    return X


# --------------------------------------------------------------------------------------------
# Batch order  (src/loaders.py:8-35, neural_admixture.py:283,364-366)
# --------------------------------------------------------------------------------------------
class EpochOrder:
    """Yields the per-epoch sample permutation exactly as the reference's DataLoader does.

    world<=1: ``RandomSampler(dataset, generator=g)`` with ``g = torch.Generator().manual_seed(seed)``
    created once (neural_admixture.py:283) and carried across epochs.  RandomSampler.__iter__ (torch
    2.x) draws ``torch.randperm(n, generator=g)`` for the epoch and then a second, discarded draw
    for the empty ``[:num_samples % n]`` tail, i.e. two randperm draws per epoch.

    world>1: ``DistributedSampler(shuffle=True, seed=seed)`` with ``set_epoch`` never called
    (loaders.py:27): the same ``randperm(n, manual_seed(seed+0))`` every epoch, padded by wrapping
    to a multiple of ``world`` and split ``indices[rank::world]``.
    """

    def __init__(self, n: int, seed: int, world: int = 1):
        import torch
        self.n, self.seed, self.world = n, seed, max(1, world)
        self._torch = torch
        self.g = torch.Generator().manual_seed(seed)

    def next_epoch(self) -> np.ndarray:
        torch = self._torch
        if self.world <= 1:
            perm = torch.randperm(self.n, generator=self.g)
            torch.randperm(self.n, generator=self.g)  # discarded tail draw of RandomSampler
            return perm.numpy().astype(np.int64)
        g = torch.Generator().manual_seed(self.seed + 0)
        idx = torch.randperm(self.n, generator=g).tolist()
        total = math.ceil(self.n / self.world) * self.world
        pad = total - len(idx)
        if pad > 0:
            idx += (idx * math.ceil(pad / len(idx)))[:pad]
        return np.asarray(idx, dtype=np.int64)

    def rank_indices(self, perm: np.ndarray, rank: int) -> np.ndarray:
        return perm[rank::self.world] if self.world > 1 else perm


def batches(order: np.ndarray, batch: int):
    """DataLoader(batch_size, drop_last=False): consecutive chunks, partial last batch kept (loaders.py:33)."""
    for s in range(0, len(order), batch):
        yield order[s:s + batch]


# --------------------------------------------------------------------------------------------
# Parameters
# --------------------------------------------------------------------------------------------
@dataclass
class Params:
    """All trainable tensors of ``Q_P`` (neural_admixture.py:100-150), float32 numpy.

    V  [M,C]  (:129-130)          g  [C] RMSNorm weight (:135)
    W1 [Hd,C], b1 [Hd] (:138-140) Wk[h] [k_h,Hd], bk[h] [k_h] (:29)   P[h] [M,k_h] decoder (:73-74)
    """
    V: np.ndarray
    g: np.ndarray
    W1: np.ndarray
    b1: np.ndarray
    Wk: List[np.ndarray]
    bk: List[np.ndarray]
    P: List[np.ndarray]
    ks: List[int] = field(default_factory=list)

    def tensors(self) -> Dict[str, np.ndarray]:
        d = {"V": self.V, "g": self.g, "W1": self.W1, "b1": self.b1}
        for h, k in enumerate(self.ks):
            d[f"Wk{h}"] = self.Wk[h]
            d[f"bk{h}"] = self.bk[h]
            d[f"P{h}"] = self.P[h]
        return d

    def copy(self) -> "Params":
        return Params(self.V.copy(), self.g.copy(), self.W1.copy(), self.b1.copy(),
                      [w.copy() for w in self.Wk], [b.copy() for b in self.bk],
                      [p.copy() for p in self.P], list(self.ks))


def init_mlp_weights(seed: int, C: int, Hd: int, ks: Sequence[int]):
    """Initial encoder weights exactly as the reference builds them: ``torch.manual_seed(seed)``
    (src/utils.py:107) followed by module construction in the order common_encoder Linear(C,Hd)
    -> heads Linear(Hd,k) for ascending k (neural_admixture.py:138-141, :29); default nn.Linear
    init; RMSNorm weight = 1."""
    import torch
    torch.manual_seed(seed)
    lin1 = torch.nn.Linear(C, Hd, bias=True)
    heads = [torch.nn.Linear(Hd, k, bias=True) for k in sorted(ks)]
    W1 = lin1.weight.detach().numpy().astype(F32).copy()
    b1 = lin1.bias.detach().numpy().astype(F32).copy()
    Wk = [h.weight.detach().numpy().astype(F32).copy() for h in heads]
    bk = [h.bias.detach().numpy().astype(F32).copy() for h in heads]
    return np.ones(C, dtype=F32), W1, b1, Wk, bk


def make_params(seed: int, V_MC: np.ndarray, P_init_SM: np.ndarray, Hd: int, ks: Sequence[int]) -> Params:
    """V_MC [M,C] (= RSVD output transposed, train.py:116), P_init_SM [sum(ks), M] (train.py:63,67);
    decoder h owns rows ini:end of P_init, transposed to [M,k] (neural_admixture.py:69-75)."""
    ks = sorted(int(k) for k in ks)
    C = V_MC.shape[1]
    g, W1, b1, Wk, bk = init_mlp_weights(seed, C, Hd, ks)
    P, ini = [], 0
    for k in ks:
        P.append(np.ascontiguousarray(P_init_SM[ini:ini + k].T.astype(F32)))
        ini += k
    return Params(np.ascontiguousarray(V_MC.astype(F32)), g, W1, b1, Wk, bk, P, ks)


# --------------------------------------------------------------------------------------------
# Forward / backward of one step  (neural_admixture.py:157-177, 83-98, 419-432 + autograd)
# --------------------------------------------------------------------------------------------
def softmax_rows(L: np.ndarray) -> np.ndarray:
    m = L.max(axis=1, keepdims=True)
    e = np.exp(L - m, dtype=F32)
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def mlp_forward(p: Params, Z: np.ndarray):
    """Z [b,C] -> (rinv, Zn, H, [Q_h]): RMSNorm, Linear+ReLU, per-head Linear+softmax (neural_admixture.py:173-176)."""
    C = Z.shape[1]
    ms = (Z * Z).sum(axis=1, dtype=F32) / F32(C)
    rinv = (F32(1) / np.sqrt(ms + F32(RMS_EPS), dtype=F32)).astype(F32)
    Zn = (Z * rinv[:, None] * p.g[None, :]).astype(F32)
    Hpre = (Zn @ p.W1.T + p.b1[None, :]).astype(F32)
    H = np.maximum(Hpre, F32(0))
    Qs = [softmax_rows((H @ p.Wk[h].T + p.bk[h][None, :]).astype(F32)) for h in range(len(p.ks))]
    return rinv, Zn, H, Qs


def encoder_forward(p: Params, X: np.ndarray):
    """X [b,M] -> (Z, rinv, Zn, H, [Q_h]).  neural_admixture.py:172-176."""
    Z = (X @ p.V).astype(F32)
    rinv, Zn, H, Qs = mlp_forward(p, Z)
    return Z, rinv, Zn, H, Qs


def bce_sum(R: np.ndarray, X: np.ndarray) -> float:
    """BCELoss(reduction='sum') with ATen's -100 clamp on both log terms (neural_admixture.py:288,431).
    Accumulated in float64 (the reference sums in fp32 with a cascade; agreement is ~1e-7 rel)."""
    with np.errstate(divide="ignore"):
        l1 = np.maximum(np.log(R, dtype=F32), LOG_CLAMP)
        l0 = np.maximum(np.log1p(-R, dtype=F32), LOG_CLAMP)
    return float(-(X.astype(np.float64) * l1 + (1.0 - X.astype(np.float64)) * l0).sum())


SUPERVISED_WEIGHT = 100.0   # supervised_loss_weight default, neural_admixture.py:249


def supervised_p_init(G: np.ndarray, y: np.ndarray, K: int) -> np.ndarray:
    """P init of supervised mode, [K,M]: per-class mean of the RAW uint8 codes (0,1,2 and 3 for missing; not
    halved, not clipped) -- model/train.py:82."""
    return np.vstack([G[y == k].astype(np.float32).mean(axis=0) for k in range(K)]).astype(np.float32)


def labels_from_pops(pops) -> np.ndarray:
    """Population names -> class indices in sorted-unique order (model/train.py:78-81)."""
    names = sorted(np.unique(np.asarray(list(pops))))
    lut = {a: i for i, a in enumerate(names)}
    return np.asarray([lut[a] for a in pops], dtype=np.int64)


def decoder_grads(Q: np.ndarray, P: np.ndarray, X: np.ndarray):
    """One head: (loss, dP [M,k], dQ [b,k]) of BCE(sum)(clamp(Q.P^T, 0, 1), X) -- neural_admixture.py:94-97 (clamp, gradient
    mask on the pre-clamp value with inclusive bounds) + BCE backward (r-x)/max(r(1-r),1e-12)."""
    Rraw = (Q @ P.T).astype(F32)
    R = np.clip(Rraw, F32(0), F32(1))
    loss = bce_sum(R, X)
    den = np.maximum((F32(1) - R) * R, BCE_EPS)
    dR = ((R - X) / den).astype(F32)
    dR[(Rraw < 0) | (Rraw > 1)] = 0
    return loss, (dR.T @ Q).astype(F32), (dR @ P).astype(F32)


def supervised_term(Q: np.ndarray, labels: np.ndarray):
    """(loss, dQ contribution) of 100 * CrossEntropyLoss(sum)(Q, labels) with Q -- the softmax OUTPUT -- used as logits
    (neural_admixture.py:470-473): loss = sum_i logsumexp(q_i) - q_i[y_i]; d/dq = softmax(q_i) - onehot."""
    qm = Q.max(axis=1, keepdims=True)
    e = np.exp(Q - qm, dtype=F32)
    se = e.sum(axis=1, keepdims=True, dtype=F32)
    lse = (qm + np.log(se, dtype=F32))[:, 0]
    rows = np.arange(Q.shape[0])
    loss = SUPERVISED_WEIGHT * float((lse.astype(np.float64) - Q[rows, labels].astype(np.float64)).sum())
    gce = (e / se).astype(F32)
    gce[rows, labels] -= F32(1)
    return loss, (F32(SUPERVISED_WEIGHT) * gce).astype(F32)


def mlp_backward(p: Params, Z: np.ndarray, rinv: np.ndarray, Zn: np.ndarray, H: np.ndarray, Qs, dQs):
    """Backward of mlp_forward given dL/dQ_h: returns (grads of g, W1, b1, Wk_h, bk_h; dZ [b,C])."""
    C = Z.shape[1]
    dH = np.zeros_like(H)
    grads: Dict[str, np.ndarray] = {}
    for h in range(len(p.ks)):
        Q, dQ = Qs[h], dQs[h]
        dL = (Q * (dQ - (dQ * Q).sum(axis=1, keepdims=True, dtype=F32))).astype(F32)
        grads[f"Wk{h}"] = (dL.T @ H).astype(F32)
        grads[f"bk{h}"] = dL.sum(axis=0, dtype=F32)
        dH += (dL @ p.Wk[h]).astype(F32)
    dHpre = dH * (H > 0)
    grads["W1"] = (dHpre.T @ Zn).astype(F32)
    grads["b1"] = dHpre.sum(axis=0, dtype=F32)
    dZn = (dHpre @ p.W1).astype(F32)
    t = dZn * p.g[None, :]
    grads["g"] = (dZn * Z * rinv[:, None]).sum(axis=0, dtype=F32)
    dZ = (rinv[:, None] * t - Z * (rinv ** 3)[:, None] * ((t * Z).sum(axis=1, keepdims=True, dtype=F32) / F32(C))).astype(F32)
    return grads, dZ


def step_grads(p: Params, G: np.ndarray, labels: Optional[np.ndarray] = None):
    """One forward+backward on batch G uint8 [b,M].  With ``labels`` (int [b]) the supervised term
    100 * CrossEntropyLoss(sum)(Q_0, labels) is added -- applied to the softmax OUTPUT of head 0 as if it were
    logits (neural_admixture.py:470-473: out[1][0] is probs[0]).  Returns (loss, grads dict keyed like
    Params.tensors(), aux dict with Z/Q).  Closed form of the autograd graph of
    neural_admixture.py:157-177 + :94-97 (clamp, mask on the pre-clamp value, inclusive bounds)
    + BCE backward (r-x)/max(r(1-r),1e-12)."""
    X = decode_x(G)
    Z, rinv, Zn, H, Qs = encoder_forward(p, X)
    loss = 0.0
    grads: Dict[str, np.ndarray] = {}
    dQs = []
    for h in range(len(p.ks)):
        l, dP, dQ = decoder_grads(Qs[h], p.P[h], X)
        loss += l
        grads[f"P{h}"] = dP
        if labels is not None and h == 0:
            ls, dq_sup = supervised_term(Qs[h], labels)
            loss += ls
            dQ = (dQ + dq_sup).astype(F32)
        dQs.append(dQ)
    g_small, dZ = mlp_backward(p, Z, rinv, Zn, H, Qs, dQs)
    grads.update(g_small)
    grads["V"] = (X.T @ dZ).astype(F32)
    aux = {"Z": Z, "Qs": Qs, "dZ": dZ, "H": H, "Zn": Zn}
    return loss, grads, aux


# --------------------------------------------------------------------------------------------
# Adam + clamp  (neural_admixture.py:187-204 optimizer, :411 step, :179-185/:412 restrict_P)
# --------------------------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam(betas=(0.9,0.95), eps=1e-8, fused) closed form: bias corrections in
    float64 on the host, element math in float32 (matches torch's fused CPU kernel to ~4e-7)."""

    def __init__(self, params: Params, lr: float):
        self.lr = float(lr)
        self.t = 0
        self.m = {k: np.zeros_like(v) for k, v in params.tensors().items()}
        self.v = {k: np.zeros_like(v) for k, v in params.tensors().items()}

    def step(self, params: Params, grads: Dict[str, np.ndarray]) -> None:
        self.t += 1
        bc1 = 1.0 - BETA1 ** self.t
        bc2 = 1.0 - BETA2 ** self.t
        step_size = F32(self.lr / bc1)
        bc2_sqrt = F32(math.sqrt(bc2))
        for k, w in params.tensors().items():
            g = grads[k].astype(F32)
            m, v = self.m[k], self.v[k]
            m += (g - m) * F32(1.0 - BETA1)                   # lerp(m, g, 1-beta1)
            v *= F32(BETA2)
            v += g * g * F32(1.0 - BETA2)
            denom = np.sqrt(v, dtype=F32) / bc2_sqrt + F32(ADAM_EPS)
            w -= step_size * (m / denom)
        for P in params.P:                                    # restrict_P
            np.clip(P, F32(0), F32(1), out=P)


# --------------------------------------------------------------------------------------------
# Whole run  (neural_admixture.py:324-392)
# --------------------------------------------------------------------------------------------
def train_run(G: np.ndarray, params: Params, epochs: int, batch_size: int, lr: float, seed: int,
              world: int = 1, record_orders: Optional[list] = None, labels: Optional[np.ndarray] = None):
    """Full training loop on uint8 G [N,M].  ``world>1`` emulates DDP: per-rank batch =
    batch_size//world (neural_admixture.py:287), DistributedSampler shards, gradients averaged over
    ranks (DDP mean all-reduce, :317), per-rank losses summed only for logging.
    Returns (params, Qs list [N,k], per-epoch loss sums)."""
    N = G.shape[0]
    world = max(1, world)
    b_local = batch_size // world if world > 1 else batch_size
    order = EpochOrder(N, seed, world)
    opt = Adam(params, lr)
    losses = []
    for _ in range(epochs):
        perm = order.next_epoch()
        if record_orders is not None:
            record_orders.append(perm.copy())
        acc = 0.0
        if world == 1:
            for idx in batches(perm, b_local):
                loss, grads, _ = step_grads(params, G[idx], None if labels is None else labels[idx])
                opt.step(params, grads)
                acc += loss
        else:
            shards = [list(batches(order.rank_indices(perm, r), b_local)) for r in range(world)]
            for s in range(len(shards[0])):
                gsum, l0 = None, 0.0
                for r in range(world):
                    loss, grads, _ = step_grads(params, G[shards[r][s]],
                                                None if labels is None else labels[shards[r][s]])
                    if r == 0:
                        l0 = loss
                    gsum = grads if gsum is None else {k: gsum[k] + grads[k] for k in grads}
                opt.step(params, {k: (v / F32(world)).astype(F32) for k, v in gsum.items()})
                acc += l0  # master logs its own rank's loss (neural_admixture.py:414-417)
        losses.append(acc)
    Qs = infer_q(G, params, min(N, 1024))
    return params, Qs, losses


def infer_q(G: np.ndarray, params: Params, batch: int = 1024) -> List[np.ndarray]:
    """Final Q pass: sequential batches of <=1024, encoder only (neural_admixture.py:369-383;
    also src/inference.py:71-77)."""
    outs = [[] for _ in params.ks]
    for s in range(0, G.shape[0], batch):
        _, _, _, _, Qs = encoder_forward(params, decode_x(G[s:s + batch]))
        for h, Q in enumerate(Qs):
            outs[h].append(Q)
    return [np.concatenate(o, axis=0) for o in outs]


# --------------------------------------------------------------------------------------------
# Reports  (src/utils_c/utils.pyx:8-40 loglikelihood; neural_admixture.py:532-553 Hudson Fst)
# --------------------------------------------------------------------------------------------
def loglikelihood(G: np.ndarray, P: np.ndarray, Q: np.ndarray, eps: float = 1e-6) -> float:
    """float64 sum over non-missing of g*log(r) + (2-g)*log1p(-r), r = clip(Q_i.P_j, eps, 1-eps),
    g clipped to [eps, 2-eps] (utils.pyx:24-40)."""
    P = P.astype(np.float64)
    Q = Q.astype(np.float64)
    total = 0.0
    for s in range(0, G.shape[0], 256):
        g = G[s:s + 256]
        rec = np.clip(Q[s:s + 256] @ P.T, eps, 1.0 - eps)
        gd = np.clip(g.astype(np.float64), eps, 2.0 - eps)
        term = gd * np.log(rec) + (2.0 - gd) * np.log1p(-rec)
        total += float(term[g != 3].sum())
    return total


def hudson_fst(p1: np.ndarray, p2: np.ndarray) -> float:
    """mean((p1-p2)^2) / (mean(p1(1-p2)+p2(1-p1)) + 1e-7)  (neural_admixture.py:545-550)."""
    p1 = p1.astype(F32)
    p2 = p2.astype(F32)
    num = np.mean((p1 - p2) ** 2, dtype=F32)
    den = np.mean(p1 * (1 - p2) + p2 * (1 - p1), dtype=F32) + F32(1e-7)
    return float(num / den)


# --------------------------------------------------------------------------------------------
# Synthetic genotypes: admixture model, K_true populations, 1 % missing
# --------------------------------------------------------------------------------------------
def synth_genotypes(N: int, M: int, K: int, seed: int = 1234, missing: float = 0.01) -> np.ndarray:
    rng = np.random.default_rng(seed)
    Fq = np.clip(0.5 * rng.beta(0.5, 0.5, size=(K, M)), 0.005, 0.5)
    Qt = rng.dirichlet(0.2 * np.ones(K), size=N)
    G = rng.binomial(2, Qt @ Fq).astype(np.uint8)
    if missing > 0:
        G[rng.random((N, M)) < missing] = 3
    return G
