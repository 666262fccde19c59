"""Oracle-backed CPU doubles of neural_admixture_amd.Engine / SnpShardedEngine (TEST INFRASTRUCTURE).

The product's step is one C call into HIP kernels (csrc/nadm_step.hip) and cannot run without a GPU.  What the gloo tests on CPU
cover is everything ABOVE the step -- train()'s init broadcasts, NeuralAdmixture's sharding by the DistributedSampler, the per-rank
batch loop, the final-Q gather, the flat parameter layout with its per-rank slices -- with the step itself RESTATED here on the
numpy oracle's kernels, following the plan of nadm_step line by line:

  dp   forward, backward -> message A = [all P]: sum over ranks, Adam + restrict_P with 1/world on THIS rank's slice and its own
       moments, all-gather of the parameters -> the same for every bucket of message B = [small | V]  (NADM_MODE_DP)
  snp  partial Z summed over ranks -> MLP forward -> pass 2 on the slice -> partial dQ summed -> MLP backward -> pass 3 on the
       slice -> Adam with 1/world on the slice                                                         (NADM_MODE_SNP)

The real step is held to the same reference fixtures on the GPU (tests/test_gpu_parity.py: world-2 over gloo on one device)."""
import numpy as np
import torch
import torch.distributed as dist

from neural_admixture_amd.engine import Engine
from neural_admixture_amd.snp_parallel import SnpShardedEngine
from oracle import nadm_oracle as O


def _adam_on(p_, g_, m_, v_, clamp, lr, grad_scale, t):
    bc1, bc2 = 1.0 - O.BETA1 ** t, 1.0 - O.BETA2 ** t
    g = g_ * np.float32(grad_scale)
    m_.add_((g - m_) * np.float32(1.0 - O.BETA1))
    v_.mul_(np.float32(O.BETA2)).add_(g * g * np.float32(1.0 - O.BETA2))
    den = v_.sqrt() / np.float32(np.sqrt(bc2)) + np.float32(O.ADAM_EPS)
    p_.sub_(np.float32(lr / bc1) * (m_ / den))
    if clamp:
        p_.clamp_(0.0, 1.0)


class OracleEngine(Engine):
    @classmethod
    def supports(cls, device):                              # the step below is numpy: any device that holds torch tensors will do
        return True

    def pack_from_host(self, data_u8, rows=None, chunk_rows=None):
        G = data_u8.numpy()
        self.G = np.ascontiguousarray(G if rows is None else G[np.asarray(rows)])
        self.xp = torch.zeros((self.G.shape[0], self.ld), dtype=torch.uint8)
        self.rows_are_sharded = rows is not None

    def _params(self) -> O.Params:
        L, h = self.lay, self.lay.heads
        sm = self.small.numpy()
        big = self.big.numpy()
        V = big[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C].copy()
        P = [big[L.p_off[i]: L.p_off[i] + L.M * L.kp[i]].reshape(L.M, L.kp[i])[:, :k].copy() for i, k in enumerate(L.ks)]
        Wk = [sm[h.wk_off[i]: h.wk_off[i] + k * L.Hd].reshape(k, L.Hd).copy() for i, k in enumerate(L.ks)]
        bk = [sm[h.bk_off[i]: h.bk_off[i] + k].copy() for i, k in enumerate(L.ks)]
        return O.Params(V, sm[h.g_off: h.g_off + L.C].copy(), sm[h.w1_off: h.w1_off + L.Hd * L.C].reshape(L.Hd, L.C).copy(),
                        sm[h.b1_off: h.b1_off + L.Hd].copy(), Wk, bk, P, list(L.ks))

    def _small_grad_vec(self, g):
        L, h = self.lay, self.lay.heads
        sm = np.zeros(L.n_small, dtype=np.float32)
        sm[h.g_off: h.g_off + L.C] = g["g"]
        sm[h.w1_off: h.w1_off + L.Hd * L.C] = g["W1"].reshape(-1)
        sm[h.b1_off: h.b1_off + L.Hd] = g["b1"]
        for i, k in enumerate(L.ks):
            sm[h.wk_off[i]: h.wk_off[i] + k * L.Hd] = g[f"Wk{i}"].reshape(-1)
            sm[h.bk_off[i]: h.bk_off[i] + k] = g[f"bk{i}"]
        return sm

    def forward(self, idx, b):
        self._idx = idx.numpy().astype(np.int64)[:b]
        p = self._params()
        _, _, _, _, Qs = O.encoder_forward(p, O.decode_x(self.G[self._idx]))
        L = self.lay
        Q = np.zeros((b, L.SP), dtype=np.float32)
        for i, k in enumerate(L.ks):
            Q[:, L.qoff[i]: L.qoff[i] + k] = Qs[i]
        self._Q[: b * L.SP] = torch.from_numpy(Q.reshape(-1))

    def backward(self, idx, b, with_loss=True):
        L = self.lay
        lab = None if self.labels is None else self.labels.numpy().astype(np.int64)[self._idx]
        loss, g, _ = O.step_grads(self._params(), self.G[self._idx], lab)
        flat = np.zeros(L.n_flat, dtype=np.float32)
        flat[: L.n_small] = self._small_grad_vec(g)
        big = flat[L.off_v:]
        big[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C] = g["V"]
        for i, k in enumerate(L.ks):
            big[L.p_off[i]: L.p_off[i] + L.M * L.kp[i]].reshape(L.M, L.kp[i])[:, :k] = g[f"P{i}"]
        self.gflat.copy_(torch.from_numpy(flat))
        if with_loss:
            self.loss_acc[0] += loss
            self.loss_acc[1] = loss

    def adam(self, lr, grad_scale=1.0):
        L = self.lay
        t = self.step_count + 1
        cut = L.off_v + L.clamp_from
        _adam_on(self.pflat[:cut], self.gflat[:cut], self.mflat[:cut], self.vflat[:cut], False, lr, grad_scale, t)
        _adam_on(self.pflat[cut:], self.gflat[cut:], self.mflat[cut:], self.vflat[cut:], True, lr, grad_scale, t)
        self._set_state(t, True)

    def _sync_message(self, msg_off, sl, mom_off, clamp, lr, t):
        """nadm_step's sync_message: reduce-scatter (here: a sum over the whole message) -> Adam on the own slice -> all-gather."""
        w, r = self.world, self.rank
        msg_g, msg_p = self.gflat[msg_off: msg_off + w * sl], self.pflat[msg_off: msg_off + w * sl]
        if w > 1:
            dist.all_reduce(msg_g, op=dist.ReduceOp.SUM)
        lo, hi = r * sl, (r + 1) * sl
        _adam_on(msg_p[lo:hi], msg_g[lo:hi], self.mflat[mom_off: mom_off + sl], self.vflat[mom_off: mom_off + sl], clamp, lr, 1.0 / w, t)
        if w > 1:
            dist.all_gather_into_tensor(msg_p, msg_p[lo:hi].clone())

    def train_step(self, idx, b, lr, with_loss=True):
        L = self.lay
        self.forward(idx, b)
        self.backward(idx, b, with_loss)
        if self.mode == "single":
            return self.adam(lr)
        assert self.mode == "dp"
        t = self.step_count + 1
        mom_a = L.slice_b if self.moments_sharded else L.msg_a_off + self.rank * L.slice_a
        self._sync_message(L.msg_a_off, L.slice_a, mom_a, True, lr, t)
        for j in range(L.n_buckets):                       # message B, bucket by bucket: V's SNP range j (bucket 0: small in front)
            mom = L.bkt_mom[j] if self.moments_sharded else L.bkt_off[j] + self.rank * L.bkt_slice[j]
            self._sync_message(L.bkt_off[j], L.bkt_slice[j], mom, False, lr, t)
        self._set_state(t, True)


class OracleSnpEngine(OracleEngine, SnpShardedEngine):
    """CPU double of snp_parallel.SnpShardedEngine: the slicing of data / parameters, read_loss and gather_rows are the product's;
    the step restates NADM_MODE_SNP of nadm_step with the oracle's pieces on this rank's SNP slice."""

    def __init__(self, *a, **k):
        SnpShardedEngine.__init__(self, *a, **k)

    def pack_from_host(self, data_u8, rows=None, chunk_rows=None):
        assert rows is None
        self.G = np.ascontiguousarray(data_u8.numpy()[:, self.m0:self.m1])
        self.xp = torch.zeros((self.G.shape[0], self.ld), dtype=torch.uint8)
        self.rows_are_sharded = False

    load_params = SnpShardedEngine.load_params
    read_loss = SnpShardedEngine.read_loss

    def _all_reduce(self, a: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def forward(self, idx, b):
        L = self.lay
        self._idx = idx.numpy().astype(np.int64)[:b]
        self._X = O.decode_x(self.G[self._idx])
        p = self._params()
        Z = self._all_reduce((self._X @ p.V).astype(np.float32))                 # partial Z over this rank's SNPs -> sum over ranks
        rinv, Zn, H, Qs = O.mlp_forward(p, Z)
        self._fw = (Z, rinv, Zn, H, Qs)
        Q = np.zeros((b, L.SP), dtype=np.float32)
        for i, k in enumerate(L.ks):
            Q[:, L.qoff[i]: L.qoff[i] + k] = Qs[i]
        self._Q[: b * L.SP] = torch.from_numpy(Q.reshape(-1))

    def backward(self, idx, b, with_loss=True):
        L = self.lay
        p = self._params()
        Z, rinv, Zn, H, Qs = self._fw
        big = self.gflat[L.off_v:].numpy()
        loss, dQs = 0.0, []
        for i, k in enumerate(L.ks):                                              # pass 2 on the slice: dP final and local
            l, dP, dQ = O.decoder_grads(Qs[i], p.P[i], self._X)
            loss += l
            big[L.p_off[i]: L.p_off[i] + L.M * L.kp[i]].reshape(L.M, L.kp[i])[:, :k] = dP
            if self.labels is not None and self.rank == 0 and i == 0:             # the supervised term enters the sum over ranks once
                ls, dq_sup = O.supervised_term(Qs[0], self.labels.numpy().astype(np.int64)[self._idx])
                loss += ls
                dQ = (dQ + dq_sup).astype(np.float32)
            dQs.append(dQ)
        flat = self._all_reduce(np.concatenate([d.reshape(-1) for d in dQs]))     # partial dQ -> sum over ranks
        o, dQr = 0, []
        for i, k in enumerate(L.ks):
            dQr.append(flat[o: o + b * k].reshape(b, k))
            o += b * k
        g, dZ = O.mlp_backward(p, Z, rinv, Zn, H, Qs, dQr)                        # replicated: identical on every rank
        self.gflat[: L.n_small] = torch.from_numpy(self._small_grad_vec(g))
        big[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C] = (self._X.T @ dZ).astype(np.float32)     # pass 3 on the slice
        if with_loss:
            self.loss_acc[0] += loss
            self.loss_acc[1] = loss

    def train_step(self, idx, b, lr, with_loss=True):
        self.forward(idx, b)
        self.backward(idx, b, with_loss)
        self.adam(lr, 1.0 / self.world)
