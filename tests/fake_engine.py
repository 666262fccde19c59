"""Oracle-backed test double of neural_admixture_amd.Engine (TEST INFRASTRUCTURE).

Only the kernel-calling primitives are replaced by the numpy oracle; everything else -- the flat
parameter/gradient layout, train_step_ddp's all-reduce + 1/world scaling, sharding, the final-Q
gather in NeuralAdmixture.launch_training -- is the product code, which is what the gloo tests cover."""
import numpy as np
import torch

from neural_admixture_amd.engine import Engine
from neural_admixture_amd.snp_parallel import SnpShardedEngine
from oracle import nadm_oracle as O


class OracleEngine(Engine):
    _CPU_TEST_DOUBLE = True

    def pack_from_host(self, data_u8, rows=None, chunk_rows=8192):
        G = data_u8.numpy()
        self.G = np.ascontiguousarray(G if rows is None else G[np.asarray(rows)])
        self.xp = torch.zeros((self.G.shape[0], self.ld), dtype=torch.uint8)

    def _params(self) -> O.Params:
        L, h = self.lay, self.lay.heads
        sm = self.small.numpy()
        big = self._big.numpy()
        V = big[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C].copy()
        P = [big[L.p_off[i]: L.p_off[i] + L.M * L.kp[i]].reshape(L.M, L.kp[i])[:, :k].copy() for i, k in enumerate(L.ks)]
        Wk = [sm[h.wk_off[i]: h.wk_off[i] + k * L.Hd].reshape(k, L.Hd).copy() for i, k in enumerate(L.ks)]
        bk = [sm[h.bk_off[i]: h.bk_off[i] + k].copy() for i, k in enumerate(L.ks)]
        return O.Params(V, sm[h.g_off: h.g_off + L.C].copy(), sm[h.w1_off: h.w1_off + L.Hd * L.C].reshape(L.Hd, L.C).copy(),
                        sm[h.b1_off: h.b1_off + L.Hd].copy(), Wk, bk, P, list(L.ks))

    def forward(self, idx, b):
        self._idx = idx.numpy().astype(np.int64)[:b]
        p = self._params()
        _, _, _, _, Qs = O.encoder_forward(p, O.decode_x(self.G[self._idx]))
        L = self.lay
        Q = np.zeros((b, L.SP), dtype=np.float32)
        for i, k in enumerate(L.ks):
            Q[:, L.qoff[i]: L.qoff[i] + k] = Qs[i]
        self.Q[: b * L.SP] = torch.from_numpy(Q.reshape(-1))

    def backward(self, idx, b, with_loss=True, on_grad_ready=None, p_parts=1, v_parts=1,
                 pre_adam=None):
        L, h = self.lay, self.lay.heads
        if pre_adam is not None:                        # the product's pass 2 applies the previous step's P update in its prologue
            self.adam_p_range(L.clamp_from, L.n_big, pre_adam[0], pre_adam[1], pre_adam[2])
        lab = None if self.labels is None else self.labels.numpy().astype(np.int64)[self._idx]
        loss, g, _ = O.step_grads(self._params(), self.G[self._idx], lab)
        big = np.zeros(L.n_big, dtype=np.float32)
        big[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C] = g["V"]
        for i, k in enumerate(L.ks):
            big[L.p_off[i]: L.p_off[i] + L.M * L.kp[i]].reshape(L.M, L.kp[i])[:, :k] = g[f"P{i}"]
        sm = np.zeros(L.n_small, dtype=np.float32)
        sm[h.g_off: h.g_off + L.C] = g["g"]
        sm[h.w1_off: h.w1_off + L.Hd * L.C] = g["W1"].reshape(-1)
        sm[h.b1_off: h.b1_off + L.Hd] = g["b1"]
        for i, k in enumerate(L.ks):
            sm[h.wk_off[i]: h.wk_off[i] + k * L.Hd] = g[f"Wk{i}"].reshape(-1)
            sm[h.bk_off[i]: h.bk_off[i] + k] = g[f"bk{i}"]
        self.gbig.copy_(torch.from_numpy(big))
        self.gsmall.copy_(torch.from_numpy(sm))
        if on_grad_ready is not None:                   # same message plan as the product: P pieces, then small+V pieces
            split = self._ns_pad + L.clamp_from
            mid = split + (L.n_big - L.clamp_from) // 2
            on_grad_ready(split, mid)
            on_grad_ready(mid, self._ns_pad + L.n_big)
            half = self._ns_pad + L.clamp_from // 2
            on_grad_ready(0, half)
            on_grad_ready(half, split)
        if with_loss:
            self.loss_acc[0] += loss
            self.loss_acc[1] = loss

    def adam(self, lr, grad_scale=1.0):
        self.step_count += 1
        for part in ("P", "V", "small"):
            self.adam_part(part, lr, grad_scale)

    def adam_v_small(self, lr, grad_scale, step=None):
        t = self.step_count if step is None else step
        cf = self.lay.clamp_from
        self._adam_on(self._big[:cf], self.gbig[:cf], self._mbig[:cf], self._vbig[:cf], False, lr, grad_scale, t)
        self._adam_on(self._small, self._gsmall, self._msmall, self._vsmall, False, lr, grad_scale, t)

    def adam_p_range(self, lo, hi, lr, grad_scale, step, stream=None):
        self._adam_on(self._big[lo:hi], self.gbig[lo:hi], self._mbig[lo:hi], self._vbig[lo:hi], True, lr, grad_scale, step)

    def adam_part(self, part, lr, grad_scale=1.0, stream=None):
        t = self.step_count
        bc1, bc2 = 1.0 - O.BETA1 ** t, 1.0 - O.BETA2 ** t
        cf = self.lay.clamp_from
        if part == "small":
            p_, g_, m_, v_, clamp = self.small, self.gsmall, self.msmall, self.vsmall, False
        elif part == "V":
            p_, g_, m_, v_, clamp = self._big[:cf], self.gbig[:cf], self._mbig[:cf], self._vbig[:cf], False
        else:
            p_, g_, m_, v_, clamp = self._big[cf:], self.gbig[cf:], self._mbig[cf:], self._vbig[cf:], True
        self._adam_on(p_, g_, m_, v_, clamp, lr, grad_scale, t)

    @staticmethod
    def _adam_on(p_, g_, m_, v_, clamp, lr, grad_scale, t):
        bc1, bc2 = 1.0 - O.BETA1 ** t, 1.0 - O.BETA2 ** t
        g = g_ * np.float32(grad_scale)
        m_.add_((g - m_) * np.float32(1.0 - O.BETA1))
        v_.mul_(np.float32(O.BETA2)).add_(g * g * np.float32(1.0 - O.BETA2))
        den = v_.sqrt() / np.float32(np.sqrt(bc2)) + np.float32(O.ADAM_EPS)
        p_.sub_(np.float32(lr / bc1) * (m_ / den))
        if clamp:
            p_.clamp_(0.0, 1.0)

    def infer_q(self, idx, b):
        self.forward(idx, b)
        L = self.lay
        Q = self.Q[: b * L.SP].view(b, L.SP)
        return [Q[:, L.qoff[i]: L.qoff[i] + k].clone() for i, k in enumerate(L.ks)]


class OracleSnpEngine(OracleEngine, SnpShardedEngine):
    """CPU double of snp_parallel.SnpShardedEngine: the STAGES (encode_partial, mlp_forward, decode_all, mlp_backward,
    encode_backward, adam_part) are the oracle's pieces on this rank's SNP slice; the step logic, the two all-reduces, the
    slicing of data/parameters, read_loss and gather_rows are the product code of SnpShardedEngine."""

    def __init__(self, *a, **k):
        SnpShardedEngine.__init__(self, *a, **k)

    def pack_from_host(self, data_u8, rows=None, chunk_rows=8192):
        assert rows is None
        self.G = np.ascontiguousarray(data_u8.numpy()[:, self.m0:self.m1])
        self.xp = torch.zeros((self.G.shape[0], self.ld), dtype=torch.uint8)
        self.rows_are_sharded = False

    load_params = SnpShardedEngine.load_params
    forward = SnpShardedEngine.forward
    backward = SnpShardedEngine.backward
    train_step = SnpShardedEngine.train_step
    read_loss = SnpShardedEngine.read_loss

    def sum_rows(self, src, rows, n, out):                    # (the product folds the slabs with nadm_sum_rows)
        torch.sum(src[: rows * n].view(rows, n), dim=0, out=out[:n])

    def encode_partial(self, idx, b):
        L = self.lay
        self._idx = idx.numpy().astype(np.int64)[:b]
        self._X = O.decode_x(self.G[self._idx])
        Zp = np.zeros((b, L.CP), dtype=np.float32)
        Zp[:, : L.C] = (self._X @ self._params().V).astype(np.float32)
        zp = self.zpart[: L.enc_chunks * b * L.CP].view(L.enc_chunks, b * L.CP)
        zp.zero_()
        zp[0] = torch.from_numpy(Zp.reshape(-1))            # all of this rank's partial sum in chunk 0

    def mlp_forward(self, b, z_src=None, n_chunks=None):
        L = self.lay
        src = self.zpart if z_src is None else z_src
        nch = L.enc_chunks if n_chunks is None else n_chunks
        Z = src[: nch * b * L.CP].view(nch, b, L.CP).sum(0).numpy()[:, : L.C].astype(np.float32)
        p = self._params()
        rinv, Zn, H, Qs = O.mlp_forward(p, Z)
        self._fw = (Z, rinv, Zn, H, Qs)
        Q = np.zeros((b, L.SP), dtype=np.float32)
        for i, k in enumerate(L.ks):
            Q[:, L.qoff[i]: L.qoff[i] + k] = Qs[i]
        self.Q[: b * L.SP] = torch.from_numpy(Q.reshape(-1))

    def decode_all(self, idx, b, with_loss=True, on_grad_ready=None, p_parts=1, supervised=True):
        L = self.lay
        p = self._params()
        Qs = self._fw[4]
        dq_offs, _ = L.dq_offsets(b)
        big = self.gbig.numpy()
        self.dqpart.zero_()
        loss = 0.0
        for i, k in enumerate(L.ks):
            l, dP, dQ = O.decoder_grads(Qs[i], p.P[i], self._X)
            loss += l
            big[L.p_off[i]: L.p_off[i] + L.M * L.kp[i]].reshape(L.M, L.kp[i])[:, :k] = dP
            if self.labels is not None and supervised and i == 0:
                ls, dq_sup = O.supervised_term(Qs[0], self.labels.numpy().astype(np.int64)[self._idx])
                loss += ls
                dQ = (dQ + dq_sup).astype(np.float32)
            blk = np.zeros((b, L.kp[i]), dtype=np.float32)
            blk[:, :k] = dQ
            self.dqpart[dq_offs[i]: dq_offs[i] + b * L.kp[i]] = torch.from_numpy(blk.reshape(-1))     # chunk 0 of the head's slab
        self._loss_partial = loss
        return 1

    def mlp_backward(self, b, n_loss, dq_src=None, dq_M=None, weights=True):
        L, h = self.lay, self.lay.heads
        assert dq_src is not None and dq_M == 1
        Z, rinv, Zn, H, Qs = self._fw
        dQs, o = [], 0
        for i, k in enumerate(L.ks):
            dQs.append(dq_src[o: o + b * L.kp[i]].view(b, L.kp[i]).numpy()[:, :k].astype(np.float32))
            o += b * L.kp[i]
        g, dZ = O.mlp_backward(self._params(), Z, rinv, Zn, H, Qs, dQs)
        sm = np.zeros(L.n_small, dtype=np.float32)
        sm[h.g_off: h.g_off + L.C] = g["g"]
        sm[h.w1_off: h.w1_off + L.Hd * L.C] = g["W1"].reshape(-1)
        sm[h.b1_off: h.b1_off + L.Hd] = g["b1"]
        for i, k in enumerate(L.ks):
            sm[h.wk_off[i]: h.wk_off[i] + k * L.Hd] = g[f"Wk{i}"].reshape(-1)
            sm[h.bk_off[i]: h.bk_off[i] + k] = g[f"bk{i}"]
        self.gsmall.copy_(torch.from_numpy(sm))
        self._dZ = dZ
        if n_loss > 0:
            self.loss_acc[0] += self._loss_partial
            self.loss_acc[1] = self._loss_partial

    def encode_backward(self, idx, b, on_grad_ready=None, v_parts=1):
        L = self.lay
        self.gbig.numpy()[: L.M * L.CP].reshape(L.M, L.CP)[:, : L.C] = (self._X.T @ self._dZ).astype(np.float32)

    def infer_q(self, idx, b):
        self.forward(idx, b)
        L = self.lay
        Q = self.Q[: b * L.SP].view(b, L.SP)
        return [Q[:, L.qoff[i]: L.qoff[i] + k].clone() for i, k in enumerate(L.ks)]
