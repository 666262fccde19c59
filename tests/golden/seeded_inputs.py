"""Inputs that are REGENERATED FROM A SEED on both sides of a fixture (r06): the full-width cases of BASELINE configs[1]
(2504 x 600k) are too big to commit, so `make_golden.py` (which runs the reference on them, in the build container) and the
tests (which run the HIP path on them, on the GPU box) both call the functions below and the fixture stores only the
reference's OUTPUTS plus a sha256 of every regenerated input.  A test that cannot reproduce the hash fails before it compares
anything.

The generator is built to give the same bytes on any host:
  * allele frequencies F [K, M] and ancestries Qt [N, K] are drawn once (beta / dirichlet of numpy's Generator) and rounded
    to grids of 2^-16 / 2^-20, so a last-bit difference of a libm call cannot change them;
  * p = Qt . F is then EXACT in float64 whatever the summation order (36-bit products, 7 terms), computed with elementwise
    operations only (no BLAS);
  * a genotype is two threshold comparisons of one raw 32-bit draw against (1-p)^2 and 1-p^2 (binomial(2, p) by inversion),
    a missing call one comparison of a raw 16-bit draw -- integer draws, IEEE elementwise arithmetic, nothing else;
  * every block of `ROWS` samples has a generator stream of its own ([seed, 1 + block]), so blocks can be made on threads.
This is SURVEY 8d's admixture model (K_true populations, U-shaped frequencies, Dirichlet(0.2) ancestries, 1 % missing) with a
different, reproducible sampler -- not the same bytes as oracle.synth_genotypes(seed).
"""
from __future__ import annotations

import hashlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROWS = 16          # samples per generator stream / work item


def model_of(N: int, M: int, K: int, seed: int):
    """(F [K,M], Qt [N,K]) float64 on their grids."""
    rng = np.random.default_rng([seed, 0])
    F = np.clip(0.5 * rng.beta(0.5, 0.5, size=(K, M)), 0.005, 0.5)
    F = np.round(F * 65536.0) / 65536.0
    Qt = rng.dirichlet(0.2 * np.ones(K), size=N)
    Qt = np.round(Qt * 1048576.0) / 1048576.0
    return F, Qt


def _block(args):
    G, F, Qt, r0, r1, seed, missing = args
    rng = np.random.default_rng([seed, 1 + r0 // ROWS])
    K, M = F.shape
    p = np.zeros((r1 - r0, M), dtype=np.float64)
    for k in range(K):                                         # exact: see the module docstring
        p += Qt[r0:r1, k:k + 1] * F[k][None, :]
    q = 1.0 - p
    t0 = q * q                                                 # P(G = 0)
    t1 = 1.0 - p * p                                           # P(G <= 1)
    u = rng.integers(0, 1 << 32, size=p.shape, dtype=np.uint32).astype(np.float64) * (1.0 / 4294967296.0)
    g = (u >= t0).astype(np.uint8)
    g += (u >= t1)
    if missing > 0:
        mz = rng.integers(0, 1 << 16, size=p.shape, dtype=np.uint16)
        g[mz < np.uint16(int(round(missing * 65536)))] = 3
    G[r0:r1] = g


def genotypes(N: int, M: int, K: int, seed: int, missing: float = 0.01, threads: int = 8) -> np.ndarray:
    """uint8 [N, M] genotype codes (0, 1, 2; 3 = missing)."""
    F, Qt = model_of(N, M, K, seed)
    G = np.empty((N, M), dtype=np.uint8)
    jobs = [(G, F, Qt, r0, min(N, r0 + ROWS), seed, missing) for r0 in range(0, N, ROWS)]
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        list(pool.map(_block, jobs))
    return G


def init_v_p(M: int, C: int, S: int, seed: int):
    """The seeded start of a trajectory fixture: V0 [M, C] ~ N(0, 1/M), P0 [S, M] ~ U(0.05, 0.95), float32 (the multibatch
    miniature's recipe at full width)."""
    rng = np.random.default_rng([seed, 7])
    V0 = (rng.standard_normal((M, C)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.05, 0.95, size=(S, M)).astype(np.float32)
    return V0, P0


def warm_v_p(N: int, M: int, K: int, C: int, seed: int):
    """A start INSIDE the regime a real run trains in -- the decoder near the true allele frequencies (what the mixture init gives),
    the projection spanning the signal directions -- that both sides can regenerate exactly: built from the generator's own F with
    elementwise float64 arithmetic only (no norms, no factorisations: an RSVD's noise-direction components are not reproducible
    between two fp32 implementations, profiles/r06_c2_vpert.txt).  V0 [M, C]: columns 0..K-1 = (F_k - 1/4) / (0.17 sqrt(M)) (unit-ish
    norm like a singular vector), the remaining columns N(0, 1/M) noise; P0 [K, M] = clip(F_k + U(-0.01, 0.01), 5e-6, 1 - 5e-6)."""
    F, _ = model_of(N, M, K, seed)
    rng = np.random.default_rng([seed, 13])
    V0 = np.empty((M, C), dtype=np.float64)
    scale = 1.0 / (0.17 * float(np.sqrt(float(M))))
    for c in range(C):
        V0[:, c] = (F[c] - 0.25) * scale if c < K else rng.standard_normal(M) / float(np.sqrt(float(M)))
    P0 = np.clip(F + (rng.random((K, M)) - 0.5) * 0.02, 5e-6, 1 - 5e-6)
    return V0.astype(np.float32), P0.astype(np.float32)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_rows(M: int, n: int, seed: int) -> np.ndarray:
    """n sorted distinct SNP indices: where a fixture keeps rows of P / V."""
    rng = np.random.default_rng([seed, 11])
    return np.sort(rng.choice(M, size=n, replace=False))
