"""GPU parity tests: every C-ABI kernel against the CPU oracle, on golden fixtures captured from the
reference and on seeded random cases (edge shapes included).  All marked ``gpu``.

Tolerances (float32 path): the kernels accumulate in fp32 like the oracle but in a different order,
so per-step quantities agree to ~1e-6 relative (1e-5 asserted); trajectories after a few epochs of
Adam agree to 1e-4 (Q) / 1e-4 (P) -- the reference's own bf16-vs-fp32 gap on the same fixtures is
8e-4 .. 7e-3."""
import os

import numpy as np
import pytest
import torch

from oracle import nadm_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
GOLD = G
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def mx(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def rel(a, b):
    return mx(a, b) / (float(np.abs(b).max()) + 1e-30)


def small_vec(p: O.Params):
    parts = [p.g, p.W1.reshape(-1), p.b1]
    for h in range(len(p.ks)):
        parts += [p.Wk[h].reshape(-1), p.bk[h]]
    return np.concatenate(parts).astype(np.float32)


def split_small(L, v):
    h = L.heads
    out = {"g": v[h.g_off:h.g_off + L.C], "W1": v[h.w1_off:h.w1_off + L.Hd * L.C].reshape(L.Hd, L.C), "b1": v[h.b1_off:h.b1_off + L.Hd]}
    for i, k in enumerate(L.ks):
        out[f"Wk{i}"] = v[h.wk_off[i]:h.wk_off[i] + k * L.Hd].reshape(k, L.Hd)
        out[f"bk{i}"] = v[h.bk_off[i]:h.bk_off[i] + k]
    return out


def make_engine(Gm, p: O.Params, bmax, **kw):
    import neural_admixture_amd as na
    dev = _dev()
    M, C = p.V.shape
    e = na.Engine(M, C, p.W1.shape[0], p.ks, dev, bmax, **kw)
    P_SM = np.concatenate([P.T for P in p.P], axis=0)
    e.load_params(p.V, P_SM, small_vec(p))
    e.pack_from_host(torch.from_numpy(np.ascontiguousarray(Gm)))
    return e


def engine_grads(e):
    L = e.lay
    g = split_small(L, e.gsmall.cpu().numpy())
    g["V"] = e.gV().cpu().numpy()
    for h in range(len(L.ks)):
        g[f"P{h}"] = e.gP(h).cpu().numpy()
    return g


# ------------------------------------------------------------------------------------------------
def test_pack_unpack_bit_exact():
    import ctypes as C
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.layout import ModelLayout
    dev = _dev()
    d = np.load(f"{G}/pack_layout.npz")
    rng = np.random.default_rng(0)
    cases = [d["G"], d["G_hibits"], rng.integers(0, 4, size=(7, 1), dtype=np.uint8), rng.integers(0, 4, size=(1, 1023), dtype=np.uint8),
             rng.integers(0, 4, size=(300, 4099), dtype=np.uint8)]
    for Gm in cases:
        N, M = Gm.shape
        ld = ModelLayout.row_stride(M)
        ref = O.pack2bit(Gm)
        # host packer
        src = torch.from_numpy(np.ascontiguousarray(Gm))
        out = torch.full((N, ld), 255, dtype=torch.uint8)
        check(lib.nadm_pack2bit_host(ptr(src), ptr(out), N, M, ld))
        assert np.array_equal(out.numpy()[:, :ref.shape[1]], ref) and not out.numpy()[:, ref.shape[1]:].any()
        # device packer + unpacker
        gd = src.to(dev)
        od = torch.full((N, ld), 255, dtype=torch.uint8, device=dev)
        check(lib.nadm_pack2bit(ptr(gd), ptr(od), N, M, ld, None))
        torch.cuda.synchronize()
        assert np.array_equal(od.cpu().numpy(), out.numpy())
        ud = torch.empty((N, M), dtype=torch.uint8, device=dev)
        check(lib.nadm_unpack2bit(ptr(od), ptr(ud), N, M, ld, None))
        torch.cuda.synchronize()
        assert np.array_equal(ud.cpu().numpy(), Gm & 3)


def decode_dz_image(img_u8: np.ndarray, b: int, CP: int) -> np.ndarray:
    """The operand image of dZ (include/nadm.h, nadm_dz_image; layout in csrc/nadm_common.h) back to [b, CP] values: per tile of 128
    samples 7 x 64 uint4; lane l = 16 q + 8 parity + c holds, for row group rg, 32 FP6 (E2M3) codes in dwords 6 rg .. 6 rg + 5 (code e
    at bits 6 e ..) and the E8M0 scale in byte rg of uint4 6's first dword; element e of K-block q is sample
    32 q + 8 (e >> 3) + 4 (e & 1) + ((e & 7) >> 1); a value is the sum of its eight pieces."""
    u32 = img_u8.view(np.uint32)
    ntiles = (b + 127) // 128
    out = np.zeros((ntiles * 128, 8), dtype=np.float64)
    for T in range(ntiles):
        tile = u32[T * 7 * 64 * 4:(T + 1) * 7 * 64 * 4].reshape(7, 64, 4)
        for l in range(64):
            q, c = l >> 4, l & 7
            dw = tile[:6, l, :].reshape(24)
            sc = int(tile[6, l, 0])
            for rg in range(4):
                bits = 0
                for i in range(6):
                    bits |= int(dw[6 * rg + i]) << (32 * i)
                scale = 2.0 ** (((sc >> (8 * rg)) & 255) - 127)
                for e in range(32):
                    code = (bits >> (6 * e)) & 63
                    ex, m = (code >> 3) & 3, code & 7
                    val = m / 8.0 if ex == 0 else (1 + m / 8.0) * 2.0 ** (ex - 1)
                    if code & 32:
                        val = -val
                    smp = 128 * T + 32 * q + 8 * (e >> 3) + 4 * (e & 1) + ((e & 7) >> 1)
                    out[smp, c] += val * scale
    return out[:b, :CP]


def test_pass3_propagates_a_nan_in_dz():
    """A NaN (or inf) in dZ must not vanish inside the FP6 operand image: the block that holds it gets the NaN scale, and column c
    of dV comes out NaN like the reference's X^T.dZ would -- a diverged run stays visible."""
    from neural_admixture_amd._lib import lib, check, ptr
    rng = np.random.default_rng(3)
    N, M = 200, 4096
    Gm = O.synth_genotypes(N, M, 3, seed=8, missing=0.02)
    V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, size=(5, M)).astype(np.float32)
    e = make_engine(Gm, O.make_params(2, V0, P0, 64, [5]), N)
    idx = torch.arange(N, dtype=torch.int32, device=e.device)
    dZ = rng.standard_normal((N, 8)).astype(np.float32)
    dZ[77, 3] = np.nan
    dZ[150, 6] = np.inf
    e.dZ[: N * 8] = torch.from_numpy(dZ.reshape(-1)).to(e.device)
    e.encode_backward(idx, N)                              # (dZ was written from outside: the pass rebuilds its operand image)
    torch.cuda.synchronize()
    g = e.gV().cpu().numpy()
    assert np.isnan(g[:, 3]).all() and not np.isfinite(g[:, 6]).any()
    assert np.isfinite(g[:, [0, 1, 2, 4, 5, 7]]).all()


@pytest.mark.parametrize("b", [800, 790, 37])
def test_dz_operand_image_of_pass3(b):
    """Pass 3 (C <= 8) consumes dZ as FP6 pieces with block scales.  The image nadm_mlp_bwd_image leaves behind -- built by whichever
    block of the launch completes a 32-sample group last -- and the one nadm_dz_image builds from dZ must both decode to dZ: exactly for
    elements within 2^-8 of their block's (32 samples x column) largest magnitude, to 2^-31 of that maximum otherwise."""
    from neural_admixture_amd._lib import lib, check, ptr
    rng = np.random.default_rng(b)
    N, M, Hd, ks = 800, 4096, 64, [5]
    Gm = O.synth_genotypes(N, M, 3, seed=5, missing=0.02)
    V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
    p = O.make_params(2, V0, P0, Hd, ks)
    e = make_engine(Gm, p, N)
    idx = torch.arange(b, dtype=torch.int32, device=e.device)
    for _ in range(2):                                   # twice: the group counters must have returned to zero
        e._dzimg.zero_()
        e.forward(idx, b)
        e.backward(idx, b, True)
        torch.cuda.synchronize()
        assert int(e._dzcnt.abs().sum().item()) == 0
        CP = e.lay.CP
        dZ = e.dZ.cpu().numpy()[: b * CP].reshape(b, CP).astype(np.float64)
        fused = decode_dz_image(e._dzimg.cpu().numpy(), b, CP)
        alone = torch.zeros_like(e._dzimg)
        check(lib.nadm_dz_image(ptr(e.dZ), b, CP, ptr(alone), None))
        torch.cuda.synchronize()
        assert np.array_equal(fused, decode_dz_image(alone.cpu().numpy(), b, CP))
        pad = np.zeros((((b + 31) // 32) * 32, CP))
        pad[:b] = np.abs(dZ)
        blockmax = np.repeat(pad.reshape(-1, 32, CP).max(axis=1), 32, axis=0)[:b]
        assert np.all(np.abs(fused - dZ) <= blockmax * 2.0 ** -31)
        near = np.abs(dZ) >= blockmax * 2.0 ** -8
        assert np.array_equal(fused[near], dZ[near])


@pytest.mark.parametrize("name", ["one_step_k3", "one_step_multihead", "one_step_k8_h1024", "one_step_edge",
                                  "one_step_supervised", "one_step_k7_h1024", "one_step_heads2to10", "one_step_k16_h1024", "one_step_k9"])
def test_one_step_against_reference_fixture(name):
    """Loss, every gradient and 3 Adam steps against tensors captured from the reference's autograd."""
    d = np.load(f"{G}/{name}.npz")
    ks = [int(k) for k in d["ks"]]
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), ks)
    Gm = d["G"]
    b = Gm.shape[0]
    e = make_engine(Gm, p, b)
    idx = torch.arange(b, dtype=torch.int32, device=e.device)
    if "labels" in d.files:                 # supervised term: BCE + 100 * CE(sum) on head 0's softmax output
        e.set_labels(d["labels"], ks[0], 100.0)
    # one_step_edge plants P rows at exactly 0 and 1: r == 0 / r == 1 entries divide by the 1e-12 floor (single gradient elements of
    # 5e11, SURVEY.md section 7 "edge semantics").  dP is still compared at 2e-5 of its maximum; the gradients BEHIND dQ (MLP, V) sum
    # such elements with others 12 orders of magnitude smaller, so their fp32 value depends on the summation order at the 1e-3
    # level (the fp32 oracle itself is 1e-3 from its float64 run there): 3e-3.  After Adam those entries move by lr whatever the
    # gradient's size, and an entry whose r sits on the rounding boundary of the clamp mask may or may not see the floor: P at 1e-4.
    edge = name.endswith("edge")
    gtol = 3e-3 if edge else 2e-5
    for s in range(3):
        e.forward(idx, b)
        e.backward(idx, b, True)
        torch.cuda.synchronize()
        _, last = e.read_loss()
        assert abs(last - float(d[f"loss{s}"])) / float(d[f"loss{s}"]) < 5e-6
        if s == 0:
            L = e.lay
            assert mx(e.Z.cpu().numpy()[: b * L.CP].reshape(b, L.CP)[:, :L.C], d["Z0"]) < 5e-6
            Q = e.Q.cpu().numpy()[: b * L.SP].reshape(b, L.SP)
            g = engine_grads(e)
            for h, k in enumerate(ks):
                assert mx(Q[:, L.qoff[h]:L.qoff[h] + k], d[f"Q0_{h}"]) < 2e-6
                assert rel(g[f"P{h}"], d[f"grad0_decoders_decoders_{h}_weight"]) < 2e-5
                assert rel(g[f"Wk{h}"], d[f"grad0_multihead_encoder_heads_{h}_weight"]) < gtol
                assert rel(g[f"bk{h}"], d[f"grad0_multihead_encoder_heads_{h}_bias"]) < gtol
            assert rel(g["V"], d["grad0_V"]) < gtol
            assert rel(g["g"], d["grad0_batch_norm_weight"]) < gtol
            assert rel(g["W1"], d["grad0_common_encoder_0_weight"]) < gtol
            assert rel(g["b1"], d["grad0_common_encoder_0_bias"]) < gtol
        e.adam(float(d["lr"]))
        torch.cuda.synchronize()
        if not edge:
            assert mx(e.V().cpu().numpy(), d[f"after{s}_V"]) < 5e-6
            sm = split_small(e.lay, e.small.cpu().numpy())
            assert mx(sm["W1"], d[f"after{s}_common_encoder_0_weight"]) < 5e-6
            assert mx(sm["g"], d[f"after{s}_batch_norm_weight"]) < 5e-6
        for h in range(len(ks)):
            assert mx(e.P(h).cpu().numpy(), d[f"after{s}_decoders_decoders_{h}_weight"]) < (1e-4 if edge else 5e-6)


@pytest.mark.parametrize("N,M,ks,Hd,C,seed", [
    (1, 5, [2], 8, 8, 0),            # single sample, M < 4*2
    (3, 1023, [3], 32, 8, 1),        # M = 1023: last byte partial, one chunk
    (65, 2049, [5], 64, 4, 2),       # just over a wave / chunk boundary, C = 4
    (130, 4100, [8], 128, 8, 3),     # K = 8 (SPL 8 path)
    (40, 3000, [9], 64, 8, 4),       # KP = 12
    (33, 2500, [16], 64, 12, 5),     # KP = 16, CP = 12
    (20, 1500, [20], 32, 8, 6),      # KP = 24 (SPL 2)
    (10, 900, [33], 32, 16, 7),      # KP = 48 (SPL 1)
    (900, 1300, [4], 64, 8, 8),      # b > 832: two row-blocks in pass 1
    (70, 2600, [2, 3, 4, 5, 6, 7, 8, 9, 10], 64, 8, 9),   # c3-style multi-head, SP = 68
    (150, 5000, [13], 64, 8, 10),    # KP = 16 on the bf16 matrix-pipe kernel: several sample tiles, ragged last one
    (12, 1200, [64], 32, 8, 11),     # K = 64: the widest head the ABI takes (NADM_MAX_K)
    (9, 800, [5], 2048, 8, 12),      # Hd = 2048: eight hidden units per thread in the register-resident MLP kernels
    (9, 800, [5], 2304, 8, 13),      # Hd > 2048: the generic MLP kernels
    (6, 700_000, [3], 64, 8, 14),    # 2735 dQ slab rows: above the threshold where the slab is folded by dq_prereduce_kernel first
])
def test_step_against_oracle_random_shapes(N, M, ks, Hd, C, seed):
    Gm = O.synth_genotypes(N, M, max(2, min(max(ks), 6)), seed=seed + 100, missing=0.05)
    rng = np.random.default_rng(seed)
    V0 = (rng.standard_normal((M, C)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
    p = O.make_params(seed, V0, P0, Hd, ks)
    e = make_engine(Gm, p, N)
    perm = rng.permutation(N).astype(np.int32)          # exercise the row gather
    idx = torch.from_numpy(perm).to(e.device)
    loss, grads, aux = O.step_grads(p, Gm[perm])
    with O.precision64():                                # rounding-free yardstick
        p64 = O.Params(*[a.astype(np.float64) if isinstance(a, np.ndarray) else [x.astype(np.float64) for x in a]
                         for a in (p.V, p.g, p.W1, p.b1, p.Wk, p.bk, p.P)], ks=list(p.ks))
        loss64, grads64, aux64 = O.step_grads(p64, Gm[perm])
    e.forward(idx, N)
    e.backward(idx, N, True)
    torch.cuda.synchronize()
    _, last = e.read_loss()
    L = e.lay
    assert abs(last - loss64) / abs(loss64) < 5e-6
    assert mx(e.Z.cpu().numpy()[: N * L.CP].reshape(N, L.CP)[:, :L.C], aux64["Z"]) < 1e-5

    def close(got, want64, want32, what):
        """GPU fp32 error vs the float64 yardstick must be of the same order as the fp32 oracle's own."""
        scale = float(np.abs(want64).max()) + 1e-30
        err_gpu, err_o32 = mx(got, want64) / scale, mx(want32, want64) / scale
        assert err_gpu < max(2e-5, 8 * err_o32), (what, err_gpu, err_o32)
    close(e.dZ.cpu().numpy()[: N * L.CP].reshape(N, L.CP)[:, :L.C], aux64["dZ"], aux["dZ"], "dZ")
    g = engine_grads(e)
    for k_, v in grads64.items():
        close(g[k_], v, grads[k_], k_)
    # padded columns stay exactly zero (they must never leak into the true ones)
    big = e.gbig.cpu().numpy()
    gv = big[: L.M * L.CP].reshape(L.M, L.CP)
    assert not gv[:, L.C:].any()
    for h, k in enumerate(ks):
        gp = big[L.p_off[h]: L.p_off[h] + L.M * L.kp[h]].reshape(L.M, L.kp[h])
        assert not gp[:, k:].any()


def test_step_with_mostly_missing_genotypes_and_an_all_missing_sample():
    """Missing calls are x = 0 in input AND target (neural_admixture.py:169-170): 60 % missing, one sample without a single
    call, one SNP column missing everywhere -- loss and gradients against the oracle, nothing non-finite."""
    N, M, ks, Hd = 37, 1900, [4], 64
    Gm = O.synth_genotypes(N, M, 4, seed=77, missing=0.6)
    Gm[5, :] = 3
    Gm[:, 100:108] = 3
    rng = np.random.default_rng(5)
    p = O.make_params(5, (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32), rng.uniform(0.02, 0.98, size=(4, M)).astype(np.float32), Hd, ks)
    e = make_engine(Gm, p, N)
    idx = torch.arange(N, dtype=torch.int32, device=e.device)
    loss, grads, aux = O.step_grads(p, Gm)
    e.forward(idx, N)
    e.backward(idx, N, True)
    torch.cuda.synchronize()
    assert abs(e.read_loss()[1] - loss) / loss < 5e-6
    g = engine_grads(e)
    for k_, v in grads.items():
        assert np.isfinite(g[k_]).all() and rel(g[k_], v) < 2e-5, k_
    assert mx(e.Z.cpu().numpy()[: N * 8].reshape(N, 8)[5], 0 * aux["Z"][5]) == 0          # the all-missing sample projects to exactly 0


def test_fast_and_generic_mlp_kernels_agree(request):
    """nadm_mlp_fwd / nadm_mlp_bwd pick register-resident kernels for Hd <= 2048, C <= 8 and the generic ones otherwise
    (the test hook nadm_test_force_generic_mlp -- test build only -- forces the latter): same outputs up to the summation order over the hidden dimension."""
    from conftest import in_hook_build
    if not in_hook_build(request):
        return
    from neural_admixture_amd._lib import lib
    rng = np.random.default_rng(21)
    for Hd, ks in ((1024, [8]), (1536, [2, 3, 4, 5]), (96, [11])):
        N, M, C = 37, 900, 8
        Gm = O.synth_genotypes(N, M, max(ks), seed=5)
        V0 = (rng.standard_normal((M, C)) / np.sqrt(M)).astype(np.float32)
        P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
        p = O.make_params(3, V0, P0, Hd, ks)
        outs = []
        for generic in (False, True):
            lib.nadm_test_force_generic_mlp(1 if generic else 0)
            try:
                e = make_engine(Gm, p, N)
                idx = torch.arange(N, dtype=torch.int32, device=e.device)
                e.forward(idx, N)
                e.backward(idx, N, True)
                torch.cuda.synchronize()
            finally:
                lib.nadm_test_force_generic_mlp(0)
            outs.append((e.Q.cpu().numpy().copy(), e.H.cpu().numpy().copy(), e.dZ.cpu().numpy().copy(), e.gsmall.cpu().numpy().copy(),
                         e.read_loss()[1]))
        for a, g in zip(outs[0][:4], outs[1][:4]):
            assert mx(a, g) <= 2e-5 * max(1.0, float(np.abs(g).max()))
        assert abs(outs[0][4] - outs[1][4]) <= 1e-6 * abs(outs[1][4])


@pytest.mark.parametrize("name", ["one_step_k3", "one_step_multihead", "one_step_k8_h1024", "one_step_edge",
                                  "one_step_supervised", "one_step_k7_h1024", "one_step_heads2to10", "one_step_k16_h1024", "one_step_k9"])
def test_production_step_against_reference_fixture(name):
    """The step the trainer and bench.py run -- Engine.train_step: Q operand images, Adam in the epilogues of passes 2 and 3,
    small-parameter update riding in the next pass 1 -- against the parameters and losses the reference's own
    forward / backward / optimizer.step / restrict_P produced for three steps (the unfused sequence has its own test above)."""
    d = np.load(f"{G}/{name}.npz")
    ks = [int(k) for k in d["ks"]]
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), ks)
    Gm = d["G"]
    b = Gm.shape[0]
    e = make_engine(Gm, p, b)
    idx = torch.arange(b, dtype=torch.int32, device=e.device)
    if "labels" in d.files:
        e.set_labels(d["labels"], ks[0], 100.0)
    # one_step_edge: P rows at exactly 0 / 1 with r == 0 entries -- the BCE backward divides by the 1e-12 floor there, single
    # gradients of 5e11 go through Adam, and which side of a rounding boundary r falls on decides whether an entry sees one: P
    # is compared at 1e-4 (Adam's step is lr = 2e-3 whatever the gradient's size), the other parameters are not compared
    edge = name.endswith("edge")
    for s in range(3):
        e.train_step(idx, b, float(d["lr"]), with_loss=True)
        torch.cuda.synchronize()
        _, last = e.read_loss()
        assert abs(last - float(d[f"loss{s}"])) / float(d[f"loss{s}"]) < 5e-6
        if not edge:
            assert mx(e.V().cpu().numpy(), d[f"after{s}_V"]) < 5e-6
            sm = split_small(e.lay, e.small.cpu().numpy())       # (the property applies the update owed to the next pass 1)
            assert mx(sm["W1"], d[f"after{s}_common_encoder_0_weight"]) < 5e-6
            assert mx(sm["b1"], d[f"after{s}_common_encoder_0_bias"]) < 5e-6
            assert mx(sm["g"], d[f"after{s}_batch_norm_weight"]) < 5e-6
            for h in range(len(ks)):
                assert mx(sm[f"Wk{h}"], d[f"after{s}_multihead_encoder_heads_{h}_weight"]) < 5e-6
        for h in range(len(ks)):
            assert mx(e.P(h).cpu().numpy(), d[f"after{s}_decoders_decoders_{h}_weight"]) < (1e-4 if edge else 5e-6)


@pytest.mark.parametrize("ks", [[5], [13]])
def test_pair_product_loss_and_its_exact_fallback(ks):
    """Pass 2 (P in [0, 1]) takes ONE logarithm per pair of genotypes -- log2 of the product of the four factors d / 1-d the codes
    select -- and recomputes a tile pair in the exact two-logarithms-per-genotype form when a product is 0 or underflows, i.e.
    whenever one of the reference's max(log, -100) clamps could be active.  Exact zeros (P rows of zeros under non-zero
    genotypes: d == 0, loss term 100 per allele copy), P rows of ones (1 - d == 0 up to rounding) and P rows of 1e-12 (d^2
    underflows) are planted; the loss must equal the exact form's (with_loss bit 1 selects it) to rounding and the oracle's
    where the reconstruction does not sit on a rounding boundary; gradients do not depend on the loss form at all."""
    rng = np.random.default_rng(21)
    N, M, Hd = 70, 3000, 64
    Gm = O.synth_genotypes(N, M, 4, seed=9, missing=0.03)
    Gm[:, 100:140] = rng.integers(0, 3, size=(N, 40))            # non-zero genotypes over the all-zero P rows
    V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
    P0[:, 100:140] = 0.0
    P0[:, 300:320] = 1e-12
    p = O.make_params(3, V0, P0, Hd, ks)
    loss_o, grads_o, _ = O.step_grads(p, Gm)
    idx = None
    res = {}
    for form in ("fast", "exact"):
        e = make_engine(Gm, p, N)
        idx = torch.arange(N, dtype=torch.int32, device=e.device)
        assert e.p_unit
        e.p_unit = form == "fast"                              # False: with_loss = 3, the exact form in every tile
        e.forward(idx, N)
        e.backward(idx, N, True)
        torch.cuda.synchronize()
        res[form] = (e.read_loss()[1], e.gbig.cpu().numpy().copy(), e.dZ.cpu().numpy().copy())
    assert np.isfinite(res["fast"][0])
    assert abs(res["fast"][0] - res["exact"][0]) <= 2e-6 * abs(res["exact"][0])
    assert abs(res["fast"][0] - loss_o) <= 5e-6 * abs(loss_o)
    assert np.array_equal(res["fast"][1], res["exact"][1]) and np.array_equal(res["fast"][2], res["exact"][2])
    # rows of ones: 1 - d is 0 or 6e-8 depending on the summation order, so only the two forms on the SAME reconstruction compare
    P1 = P0.copy()
    P1[:, 200:230] = 1.0
    p1 = O.make_params(3, V0, P1, Hd, ks)
    out = []
    for form in ("fast", "exact"):
        e = make_engine(Gm, p1, N)
        e.p_unit = form == "fast"
        e.forward(idx, N)
        e.backward(idx, N, True)
        torch.cuda.synchronize()
        out.append(e.read_loss()[1])
    assert np.isfinite(out[0]) and abs(out[0] - out[1]) <= 2e-6 * abs(out[1])
    # ... and called heterozygous in EVERY sample: where the reconstruction rounds above 1 both factors of a pair are d (1 - d) < 0 and
    # their product is positive -- the factor's saturation (clamp on the blend) must still send the pair to the exact form, whose
    # term is 50 per genotype there (log 1 and the -100 floor, halved)
    G2 = Gm.copy()
    G2[:, 200:230] = 1
    out = []
    for form in ("fast", "exact"):
        e = make_engine(G2, p1, N)
        e.p_unit = form == "fast"
        e.forward(idx, N)
        e.backward(idx, N, True)
        torch.cuda.synchronize()
        out.append(e.read_loss()[1])
    assert np.isfinite(out[0]) and abs(out[0] - out[1]) <= 2e-6 * abs(out[1]), out


def test_snp_subrange_launches_give_identical_gradients():
    """include/nadm.h (nadm_decode_chunk_snps): passes 2 and 3 may be launched on SNP sub-ranges [m0, m1) with m0 a multiple of
    lcm(chunk, 1024) -- pointers advanced by m0 -- and must write the bits of the whole-range launch: gradients, dQ slab rows, loss
    slots."""
    import ctypes as C
    import math
    from neural_admixture_amd._lib import lib, check, ptr
    rng = np.random.default_rng(2)
    for M, ks in ((5003, [6]), (9001, [2, 5, 11])):
        N, Hd = 90, 64
        Gm = O.synth_genotypes(N, M, 4, seed=M)
        V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
        P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
        p = O.make_params(1, V0, P0, Hd, ks)
        e = make_engine(Gm, p, N)
        L = e.lay
        idx = torch.arange(N, dtype=torch.int32, device=e.device)
        e.forward(idx, N)
        e.backward(idx, N, True)
        torch.cuda.synchronize()
        whole_g, whole_dq, whole_loss = e.gflat.clone(), e.dqpart.clone(), e.losspart.clone()
        dq_offs, _ = L.dq_offsets(N)
        loss_offs = L.loss_offsets()
        for parts in (2, 3):
            e.gflat.zero_(); e.dqpart.zero_(); e.losspart.zero_()
            gbig = e.gflat[L.off_v:]
            for h, kp in enumerate(L.kp):
                cs = int(lib.nadm_decode_chunk_snps(kp))
                align = cs * 1024 // math.gcd(cs, 1024)
                units = (M + align - 1) // align
                cuts = sorted({min(M, (units * i // parts) * align) for i in range(parts)} | {M})
                for m0, m1 in zip(cuts[:-1], cuts[1:]):
                    c0 = m0 // cs
                    check(lib.nadm_decode_bce(C.c_void_p(e.xp.data_ptr() + m0 // 4), e.ld, ptr(idx), N, m1 - m0,
                                              C.c_void_p(e.big.data_ptr() + (L.p_off[h] + m0 * kp) * 4), kp, C.c_void_p(e.Q.data_ptr() + L.qoff[h] * 4), L.SP,
                                              C.c_void_p(gbig.data_ptr() + (L.p_off[h] + m0 * kp) * 4), C.c_void_p(e.dqpart.data_ptr() + (dq_offs[h] + c0 * N * kp) * 4),
                                              C.c_void_p(e.losspart.data_ptr() + (loss_offs[h] + c0) * 4), 1, None))
            units = (M + 1023) // 1024
            cuts = sorted({min(M, (units * i // parts) * 1024) for i in range(parts)} | {M})
            dzimg = torch.empty(int(lib.nadm_dz_image_bytes(N)), dtype=torch.uint8, device=e.device)
            check(lib.nadm_dz_image(ptr(e.dZ), N, L.CP, ptr(dzimg), None))
            for m0, m1 in zip(cuts[:-1], cuts[1:]):
                check(lib.nadm_encode_bwd(C.c_void_p(e.xp.data_ptr() + m0 // 4), e.ld, ptr(idx), N, m1 - m0, ptr(e.dZ), ptr(dzimg), L.CP,
                                          C.c_void_p(gbig.data_ptr() + m0 * L.CP * 4), 0, None))
            torch.cuda.synchronize()
            assert torch.equal(e.gflat[L.off_v:], whole_g[L.off_v:])
            assert torch.equal(e.dqpart, whole_dq) and torch.equal(e.losspart[: L.n_loss], whole_loss[: L.n_loss])


def test_snp_sharded_engine_on_gpu():
    """snp_parallel.SnpShardedEngine with the real kernels: (1) on a 1-rank RCCL communicator (nadm_step, NADM_MODE_SNP: the two
    all-reduces run through RCCL) it must reproduce the plain engine (same kernels, Z and dQ folded by nadm_sum_rows instead of
    inside the MLP kernels -> 1e-6); (2) two slices of one matrix driven stage by stage on one GPU, with the two all-reduces
    done by hand, must reproduce it as well (slicing of packed columns, of V / P, kernels on a slice whose length is not a
    multiple of any chunk)."""
    from neural_admixture_amd.comm import rccl_comm, torch_comm
    from neural_admixture_amd.snp_parallel import SnpShardedEngine, snp_slices
    from neural_admixture_amd.model import init_encoder_weights
    dev = _dev()
    rng = np.random.default_rng(31)
    N, M, ks, Hd, b = 200, 9001, [3, 7], 64, 150
    Gm = O.synth_genotypes(N, M, 4, seed=8, missing=0.03)
    V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
    small = init_encoder_weights(4, 8, Hd, ks)
    p = O.make_params(4, V0, P0, Hd, ks)
    data = torch.from_numpy(Gm)
    idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(dev)

    ref = make_engine(Gm, p, b)
    for _ in range(3):
        ref.train_step(idx, b, 2e-3, True)
    torch.cuda.synchronize()
    ref_loss = ref.read_loss()[0]

    # (1) world 1, a real RCCL communicator
    comm = rccl_comm(0, 1)
    e1 = SnpShardedEngine(M, 8, Hd, ks, dev, b, comm=comm)
    e1.load_params(V0, P0, small)
    e1.pack_from_host(data)
    for _ in range(3):
        e1.train_step(idx, b, 2e-3, True)
    torch.cuda.synchronize()
    assert mx(e1.big.cpu().numpy(), ref.big.cpu().numpy()) < 2e-6 and mx(e1.small.cpu().numpy(), ref.small.cpu().numpy()) < 2e-6
    assert abs(e1.read_loss()[0] - ref_loss) < 1e-6 * abs(ref_loss)
    del e1
    comm.close()

    # (2) two slices, stages driven by hand (no process group: _all_reduce is a no-op at world... so sum explicitly)
    sl = snp_slices(M, 2)
    assert sl[0][1] == sl[1][0] and sl[1][1] == M and (sl[0][1] - sl[0][0]) % 4 == 0
    es = []
    for r in range(2):
        e = SnpShardedEngine(M, 8, Hd, ks, dev, b, comm=torch_comm(r, 2))      # (its collectives are never called: stages by hand)
        e.load_params(V0, P0, small)
        e.pack_from_host(data)
        es.append(e)
    L = es[0].lay
    for _ in range(3):
        zs = []
        for e in es:
            e.encode_partial(idx, b)
            zs.append(e.zpart[: e.lay.enc_chunks * b * L.CP].view(e.lay.enc_chunks, b * L.CP).sum(0))
        zsum = (zs[0] + zs[1]).contiguous()
        dqs = []
        for r, e in enumerate(es):
            e.mlp_forward(b, zsum, 1)
            n_loss = e.decode_all(idx, b, True, supervised=(r == 0))
            offs, _ = e.lay.dq_offsets(b)
            dqs.append(torch.cat([e.dqpart[offs[h]: offs[h] + e.lay.dec_chunks[h] * b * kp].view(e.lay.dec_chunks[h], b * kp).sum(0)
                                  for h, kp in enumerate(L.kp)]))
            e._n_loss = n_loss
        dqsum = (dqs[0] + dqs[1]).contiguous()
        for e in es:
            e.mlp_backward(b, e._n_loss, dq_src=dqsum, dq_M=1)
            e.encode_backward(idx, b)
            e.adam(2e-3, 1.0)
    torch.cuda.synchronize()
    got_V = np.concatenate([e.V().cpu().numpy() for e in es], axis=0)
    assert mx(got_V, ref.V().cpu().numpy()) < 2e-6
    for h in range(len(ks)):
        assert mx(np.concatenate([e.P(h).cpu().numpy() for e in es], axis=0), ref.P(h).cpu().numpy()) < 2e-6
    assert mx(es[0].small.cpu().numpy(), ref.small.cpu().numpy()) < 2e-6 and np.array_equal(es[0].small.cpu().numpy(), es[1].small.cpu().numpy())
    tot = sum(float(e.loss_acc.cpu()[0]) for e in es)
    assert abs(tot - ref_loss) < 1e-6 * abs(ref_loss)


def test_without_loss_gives_same_gradients():
    Gm = O.synth_genotypes(50, 2100, 4, seed=5)
    rng = np.random.default_rng(1)
    p = O.make_params(1, (rng.standard_normal((2100, 8)) / 40).astype(np.float32), rng.uniform(0.1, 0.9, (4, 2100)).astype(np.float32), 64, [4])
    e = make_engine(Gm, p, 50)
    idx = torch.arange(50, dtype=torch.int32, device=e.device)
    e.forward(idx, 50); e.backward(idx, 50, True); torch.cuda.synchronize()
    g1, s1 = e.gbig.clone(), e.gsmall.clone()
    e.forward(idx, 50); e.backward(idx, 50, False); torch.cuda.synchronize()
    assert torch.equal(g1, e.gbig) and torch.equal(s1, e.gsmall)     # bit-identical and deterministic


def _run_trajectory(Gm, p, epochs, batch, lr, seed):
    import neural_admixture_amd as na
    dev = _dev()
    N, M = Gm.shape
    tr = na.NeuralAdmixture(p.ks[0] if len(p.ks) == 1 else None, epochs, batch, lr, dev, seed, 1, True, None,
                            None if len(p.ks) == 1 else p.ks[0], None if len(p.ks) == 1 else p.ks[-1], loss_mode="always")
    P_SM = torch.from_numpy(np.concatenate([P.T for P in p.P], axis=0))
    Qs, Ps, model = tr.launch_training(P_SM, torch.from_numpy(np.ascontiguousarray(Gm)), p.W1.shape[0], p.V.shape[1],
                                       torch.from_numpy(p.V), M, N, None)
    return Qs, Ps, model, tr


def test_trajectory_multibatch_k8_vs_reference():
    """c4-shaped miniature: same init, same RandomSampler batch order, 3 epochs; Q/P/V vs the reference's fp32 run."""
    d = np.load(f"{G}/multibatch_k8.npz")
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), [int(d["K"])])
    Qs, Ps, model, tr = _run_trajectory(Gm, p, int(d["epochs"]), int(d["b"]), float(d["lr"]), int(d["seed"]))
    assert mx(Qs[0], d["hi_Q"]) < 2e-3            # stated tolerance: Q <= 2e-3 after <= 5 epochs
    assert mx(Ps[0], d["hi_P"]) < 1e-3
    assert mx(model.state_dict()["V"].numpy(), d["hi_V"]) < 2e-3
    ref = d["hi_losses"].reshape(int(d["epochs"]), -1).sum(1)
    got = np.asarray([tr.epoch_losses[e_] for e_ in range(int(d["epochs"]))])
    assert np.allclose(got, ref, rtol=2e-5)
    # and we are closer to the fp32 reference than the reference's own bf16 run is
    assert mx(Qs[0], d["hi_Q"]) < mx(d["med_Q"], d["hi_Q"])


def test_trajectory_multihead_vs_reference():
    d = np.load(f"{G}/multihead_run.npz")
    ks = [int(k) for k in d["ks"]]
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["V0"], d["P0"], int(d["Hd"]), ks)
    Qs, Ps, model, tr = _run_trajectory(Gm, p, int(d["epochs"]), int(d["b"]), float(d["lr"]), int(d["seed"]))
    for h in range(len(ks)):
        assert mx(Qs[h], d[f"hi_Q{h}"]) < 1e-3
        assert mx(Ps[h], d[f"hi_P{h}"]) < 1e-3
    got = np.asarray([tr.epoch_losses[e_] for e_ in range(int(d["epochs"]))])
    assert np.allclose(got, d["hi_losses"].reshape(int(d["epochs"]), -1).sum(1), rtol=2e-5)


@pytest.mark.parametrize("ep", [5, 25])
def test_trajectory_demo_c1_vs_reference(ep):
    """BASELINE config 1: bundled demo data, K=3, same RSVD V and GMM P_init as the reference run."""
    d = np.load(f"{G}/demo_k3.npz")
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    p = O.make_params(int(d["seed"]), d["Vt"].T, d["P_init"], int(d["Hd"]), [3])
    Qs, Ps, model, tr = _run_trajectory(Gm, p, ep, 800, float(d["lr"]), int(d["seed"]))
    assert mx(Qs[0], d[f"hi_e{ep}_Q"]) < 2e-3
    assert mx(Ps[0], d[f"hi_e{ep}_P"]) < 1e-2
    got = np.asarray([tr.epoch_losses[e_] for e_ in range(ep)])
    assert np.allclose(got, d[f"hi_e{ep}_losses"], rtol=5e-5)
    sd = model.state_dict()
    assert set(sd) == {"V", "batch_norm.weight", "common_encoder.0.weight", "common_encoder.0.bias",
                       "multihead_encoder.heads.0.weight", "multihead_encoder.heads.0.bias", "decoders.decoders.0.weight"}
    if ep == 5:
        from neural_admixture_amd.report import loglikelihood_packed
        ll = loglikelihood_packed(tr.engine, torch.from_numpy(Gm), Ps[0], Qs[0])
        assert abs(ll - float(d["hi_e5_loglik"])) / abs(float(d["hi_e5_loglik"])) < 1e-4


def _end_of_run_vs_reference(Gm, p, d, batch, lr, seed):
    """A run of the PRODUCTION path (NeuralAdmixture.launch_training on the HIP engine: fused epilogues, deferred small update,
    prefetched epoch orders) of the fixture's length, held to SURVEY 8c's end-of-run bounds against the reference's fp32 run
    (tests/test_oracle_golden.check_end_of_run: mean |dQ| <= 1e-2, max |dP| <= 1e-2, per-epoch loss <= 1e-3 rel, log-likelihood
    <= 1e-4 rel, and closer to it than the reference's own bf16 run is -- on average AND in the worst sample)."""
    from neural_admixture_amd.report import loglikelihood_packed
    from test_oracle_golden import check_end_of_run
    ep = int(d["epochs"])
    Qs, Ps, model, tr = _run_trajectory(Gm, p, ep, batch, lr, seed)
    ll = loglikelihood_packed(tr.engine, torch.from_numpy(Gm), Ps[0], Qs[0])
    dq, ref_dq = np.abs(Qs[0] - d["hi_Q"]), np.abs(d["med_Q"] - d["hi_Q"])
    print(f"end of run ({ep} epochs): max |dQ| {dq.max():.3e} (reference's bf16 run {ref_dq.max():.3e}), mean {dq.mean():.3e} ({ref_dq.mean():.3e})")
    # r06: the production path is held to the yardstick itself, worst sample included (r04-r05 allowed 2 x): measured 5.82e-2 against the
    # reference's 6.03e-2 on the 250-epoch demo, 1.14e-2 against 4.80e-2 on the 60-epoch miniature -- and the kernels are bit-reproducible
    check_end_of_run(Qs[0], Ps[0], [tr.epoch_losses[e_] for e_ in range(ep)], ll, d, worst_sample_factor=1.0)


def test_default_horizon_demo_250_epochs_vs_reference():
    """BASELINE's second metric is the wall-clock of a DEFAULT run = 250 epochs (entry.py:27, neural_admixture.py:365-366): the
    bundled demo, K=3, from the reference's own RSVD V and GMM P_init, against the reference's 250-epoch fp32 run."""
    dm, d = np.load(f"{G}/demo_k3.npz"), np.load(f"{G}/demo_k3_e250.npz")
    Gm = O.unpack2bit(dm["G_packed"], int(dm["M"]))
    p = O.make_params(int(dm["seed"]), dm["Vt"].T, dm["P_init"], int(dm["Hd"]), [3])
    _end_of_run_vs_reference(Gm, p, d, 800, float(dm["lr"]), int(dm["seed"]))


def test_long_horizon_multibatch_60_epochs_vs_reference():
    """180 steps (60 epochs of 400 + 400 + 200 rows, the sampler's own orders) of the K=8 miniature."""
    m, d = np.load(f"{G}/multibatch_k8.npz"), np.load(f"{G}/multibatch_k8_e60.npz")
    Gm = O.unpack2bit(m["G_packed"], int(m["M"]))
    p = O.make_params(int(m["seed"]), m["V0"], m["P0"], int(m["Hd"]), [int(m["K"])])
    _end_of_run_vs_reference(Gm, p, d, int(m["b"]), float(m["lr"]), int(m["seed"]))


@pytest.mark.parametrize("b,M,ks", [(800, 500_000, [8]),                    # configs[3] / the bench workload
                                    (800, 600_000, [7]),                    # configs[1]: 1000-Genomes scale, single head K=7
                                    (104, 600_000, [7]),                    # ... and its partial last batch (2504 = 3 * 800 + 104)
                                    (64, 1_000_000, [16]),                  # configs[4] width: M = 1M, K = 16 (two k slots in pass 2)
                                    (800, 1_000_000, [16]),                 # configs[4] at its full per-GPU batch (r05; ~25 GB of temporaries)
                                    (104, 600_000, list(range(2, 11)))])    # configs[2]: heads K = 2..10 over the 1000-Genomes width
def test_full_width_against_torch_fp32_on_device(b, M, ks):
    """BASELINE-scale width: the three passes against a plain torch fp32 / fp64 computation on the same GPU from the
    unpacked matrix (independent code path: every configs[] width meets something other than itself), plus linearity."""
    dev = _dev()
    import neural_admixture_amd as na
    from neural_admixture_amd._lib import lib, check, ptr
    import ctypes as C
    Cc, Hd, K = 8, 1024, max(ks)
    e = na.Engine(M, Cc, Hd, ks, dev, b)
    g = torch.Generator(device="cpu").manual_seed(0)
    Qt = torch.distributions.Dirichlet(torch.full((K,), 0.2)).sample((b,)).float().to(dev)
    Fq = (0.5 * torch.rand(K, M, generator=g)).clamp(0.005, 0.5).to(dev)
    xp = torch.empty((b, e.ld), dtype=torch.uint8, device=dev)
    check(lib.nadm_synth_packed(ptr(xp), b, 0, M, e.ld, ptr(Qt), ptr(Fq), K, 0.01, 1234, None))
    e.set_packed(xp)
    Gd = torch.empty((b, M), dtype=torch.uint8, device=dev)
    check(lib.nadm_unpack2bit(ptr(xp), ptr(Gd), b, M, e.ld, None))
    cnt = torch.bincount(Gd.reshape(-1).to(torch.int64), minlength=4).cpu().numpy() / (b * M)
    assert 0.005 < cnt[3] < 0.02 and cnt[0] > 0.3 and cnt[1] > 0.05          # generator sanity
    X = torch.where(Gd == 3, torch.zeros((), device=dev), Gd.float() / 2)
    del Gd
    V = (torch.randn(M, Cc, generator=g) / M ** 0.5).numpy()
    P = torch.rand(sum(ks), M, generator=g).mul(0.9).add(0.05).numpy()
    small = na.model.init_encoder_weights(42, Cc, Hd, ks)
    e.load_params(V, P, small)
    idx = torch.arange(b, dtype=torch.int32, device=dev)
    e.forward(idx, b)
    e.backward(idx, b, True)
    torch.cuda.synchronize()
    Vd = e.V().contiguous()
    Z_ref = X.double() @ Vd.double()
    assert (e.Z[: b * Cc].view(b, Cc).double() - Z_ref).abs().max().item() < 1e-5 * Z_ref.abs().max().item() + 1e-6
    loss_ref = 0.0
    for h, k in enumerate(ks):
        Pd = e.P(h).contiguous()
        Q = e.Q[: b * e.lay.SP].view(b, e.lay.SP)[:, e.lay.qoff[h]: e.lay.qoff[h] + k].contiguous()
        Rraw = Q @ Pd.T
        R = Rraw.clamp(0, 1)
        loss_ref += torch.nn.functional.binary_cross_entropy(R, X, reduction="sum").double().item()
        dR = (R - X) / ((1 - R) * R).clamp_min(1e-12) * ((Rraw >= 0) & (Rraw <= 1))
        dP_ref = (dR.double().T @ Q.double())
        assert (e.gP(h).double() - dP_ref).abs().max().item() < 2e-5 * dP_ref.abs().max().item(), h
        del Rraw, R, dR, dP_ref
    assert abs(e.read_loss()[1] - loss_ref) / loss_ref < 2e-5
    dZ = e.dZ[: b * Cc].view(b, Cc)
    dV_ref = X.double().T @ dZ.double()
    assert (e.gV().double() - dV_ref).abs().max().item() < 2e-5 * dV_ref.abs().max().item()
    # linearity of pass 3 in dZ: dV(2*dZ) == 2*dV(dZ) exactly (power-of-two scaling is exact in fp32)
    g1 = e.gV().clone()
    e.dZ.mul_(2.0)
    e.encode_backward(idx, b)                              # (dZ was edited: the pass rebuilds its operand image)
    torch.cuda.synchronize()
    assert torch.equal(e.gV(), 2 * g1)


def test_train_boundary_on_demo_from_bed(tmp_path, caplog):
    """The drop-in boundary itself: train(...) with the reference's positional signature on BASELINE config 1
    (demo BED, K=3, 5 epochs), fed by the BED -> packed reader (no uint8 [N,M] matrix), same RSVD V as the
    reference run; GMM init, training, final Q, log-likelihood report; outputs written in the reference's formats."""
    import logging
    import neural_admixture_amd as na
    from neural_admixture_amd.io import read_bed_packed, write_outputs, save_model
    dev = _dev()
    d = np.load(f"{G}/demo_k3.npz")
    d["bed_bytes"].tofile(tmp_path / "demo.bed")
    (tmp_path / "demo.fam").write_text("\n".join(["s"] * int(d["N"])) + "\n")
    data = read_bed_packed(str(tmp_path / "demo.bed"))
    with caplog.at_level(logging.INFO):
        Ps, Qs, model = na.train(5, 800, float(d["lr"]), 3, int(d["seed"]), data, dev, 1, int(d["Hd"]), True, d["Vt"], None, None, None, 8)
    assert Ps[0].shape == (int(d["M"]), 3) and Qs[0].shape == (int(d["N"]), 3) and Ps[0].dtype == np.float32
    assert mx(Qs[0], d["hi_e5_Q"]) < 2e-3 and mx(Ps[0], d["hi_e5_P"]) < 1e-2
    ll = [float(r.getMessage().split(":")[1].strip().rstrip(".")) for r in caplog.records if "Log-likelihood" in r.getMessage()]
    assert len(ll) == 1 and abs(ll[0] - float(d["hi_e5_loglik"])) / abs(float(d["hi_e5_loglik"])) < 1e-4
    save_model(model, "demo_run", str(tmp_path))
    write_outputs(Qs, "demo_run", 3, None, None, tmp_path, Ps)
    sd = torch.load(tmp_path / "demo_run.pt", weights_only=True)
    assert "V" in sd and not any(k.startswith("decoders") for k in sd)          # src/main.py:41
    assert np.loadtxt(tmp_path / "demo_run.3.Q").shape == (int(d["N"]), 3)
    # infer path: reload the saved encoder and recompute Q from raw genotypes (src/inference.py:54-77)
    m2 = na.Q_P(int(d["Hd"]), 8, ks_list=[3], is_train=False).load_state_dict(sd, device=dev)
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    probs, _ = m2(torch.from_numpy(Gm))
    assert mx(probs[0].cpu().numpy(), Qs[0]) < 1e-6


def test_train_boundary_supervised_vs_reference():
    """Supervised mode through train(..., pops=[names]) against the reference's own train() on the same inputs
    (tests/golden/supervised_k4.npz): label mapping, raw-code class-mean P init, BCE + 100*CE.  Step 0 is a
    rounding-level pin; later steps are compared at the reference's own fp32-vs-bf16 self-distance (the init
    saturates most of R, see tests/test_oracle_golden.py::test_supervised_run)."""
    import neural_admixture_amd as na
    from neural_admixture_amd.model import NeuralAdmixture
    dev = _dev()
    d = np.load(f"{G}/supervised_k4.npz")
    N, M, K, Hd = int(d["N"]), int(d["M"]), int(d["K"]), int(d["Hd"])
    Gm = O.unpack2bit(d["G_packed"], M)
    pops = [str(a) for a in d["pops"]]
    data = torch.from_numpy(Gm)
    # step-level: the trainer with loss on every step
    from neural_admixture_amd.train import supervised_init
    y, P0 = supervised_init(Gm, pops, K)
    assert np.array_equal(y, O.labels_from_pops(pops)) and mx(P0, O.supervised_p_init(Gm, y, K)) < 1e-6
    tr = NeuralAdmixture(K, 1, int(d["b"]), float(d["lr"]), dev, int(d["seed"]), 1, True, None, None, None, loss_mode="always")
    tr.launch_training(torch.from_numpy(P0), data, Hd, 8, torch.from_numpy(np.ascontiguousarray(d["Vt"].T)), M, N,
                       torch.from_numpy(y))
    ref, med = d["hi_losses"], d["med_losses"]
    assert abs(tr.epoch_losses[0] - ref[:3].sum()) / ref[:3].sum() < max(3 * abs(med[:3].sum() - ref[:3].sum()) / ref[:3].sum(), 5e-3)
    # one step exactly
    e = tr.engine_cls(M, 8, Hd, [K], dev, N)
    from neural_admixture_amd.model import init_encoder_weights
    e.load_params(np.ascontiguousarray(d["Vt"].T), P0, init_encoder_weights(int(d["seed"]), 8, Hd, [K]))
    e.pack_from_host(data)
    e.set_labels(y, K)
    g = torch.Generator().manual_seed(int(d["seed"]))
    order = np.asarray(list(iter(torch.utils.data.RandomSampler(range(N), generator=g))), dtype=np.int32)[: int(d["b"])]
    idx = torch.from_numpy(order).to(dev)
    e.forward(idx, len(order)); e.backward(idx, len(order), True)
    assert abs(e.read_loss()[1] - ref[0]) / ref[0] < 5e-6
    # the boundary call itself
    Ps, Qs, model = na.train(int(d["epochs"]), int(d["b"]), float(d["lr"]), K, int(d["seed"]), data, dev, 1, Hd, True,
                             d["Vt"], pops, None, None, 8)
    # Tolerance of the end-of-run Q: the supervised start (class means of the RAW codes, values up to 3, train.py:82) saturates
    # most of R at the clamp, where one rounding decides whether an element sees the 1e-12 floor (gradients of 1e3 .. 1e12 into
    # Adam).  The reference's own fp32 and bf16 runs of this fixture therefore end 0.135 apart in Q (max; 0.058 mean) after its
    # 9 steps although their step-0 losses agree to 4e-6 -- there is no tighter yardstick than that self-distance for any two
    # fp32 summation orders.  Asserted: within twice the reference's distance from itself (max) and within it on average; the
    # rounding-level pins are the step-0 loss above (5e-6) and one_step_supervised (gradients 2e-5, test_one_step_...).
    selfQ, selfQ_mean = mx(d["med_Q"], d["hi_Q"]), float(np.abs(d["med_Q"] - d["hi_Q"]).mean())
    assert Ps[0].shape == (M, K) and Qs[0].shape == (N, K)
    assert mx(Qs[0], d["hi_Q"]) < 2 * selfQ, (mx(Qs[0], d["hi_Q"]), selfQ)
    assert float(np.abs(Qs[0] - d["hi_Q"]).mean()) < selfQ_mean, (float(np.abs(Qs[0] - d["hi_Q"]).mean()), selfQ_mean)
    assert float(Ps[0].min()) >= 0.0 and float(Ps[0].max()) <= 1.0
    with pytest.raises(AssertionError):                      # train.py:79
        na.train(1, 100, 2e-3, K + 1, 1, data, dev, 1, Hd, True, d["Vt"], pops, None, None, 8)


def test_rsvd_on_the_gpu_with_fewer_samples_than_sketch_columns():
    """svd.RSVD's GPU path takes the SVD of the wide factor from the Cholesky factor of its Gram matrix; with N < k' = 20 samples that
    matrix is singular and the path falls back to the reference's own host SVD -- same subspace as the all-host path."""
    from neural_admixture_amd.svd import RSVD
    dev = _dev()
    Gm = O.synth_genotypes(12, 3000, 3, seed=8, missing=0.01)
    Vg = RSVD(Gm, 12, 3000, 8, 5, device=dev)
    Vh = RSVD(Gm, 12, 3000, 8, 5, device=None)
    assert Vg.shape == Vh.shape == (8, 3000) and np.isfinite(Vg).all()
    for r in range(4):                                     # the leading directions (the trailing ones of a rank-12 matrix are noise)
        assert abs(float(np.dot(Vg[r], Vh[r]))) / (np.linalg.norm(Vg[r]) * np.linalg.norm(Vh[r])) > 0.999


def test_pca_projection_on_gpu_counts_missing_as_one_and_a_half():
    """train.pca_project_gpu (nadm_pca_project) = (G/2) @ V.T on the raw codes, missing (3) -> 1.5, as the reference's
    init-time projection (train.py:49-55); from a uint8 matrix and from PackedGenotypes, several row chunks."""
    from neural_admixture_amd.train import pca_project_gpu
    from neural_admixture_amd.io import PackedGenotypes
    from neural_admixture_amd.layout import ModelLayout
    dev = _dev()
    rng = np.random.default_rng(3)
    for N, M, C in ((300, 5003, 8), (37, 129, 5)):
        Gm = O.synth_genotypes(N, M, 4, seed=9, missing=0.05)
        assert (Gm == 3).any()
        V = (rng.standard_normal((C, M)) / np.sqrt(M)).astype(np.float32)
        ref = (Gm.astype(np.float64) / 2) @ V.T.astype(np.float64)
        a = pca_project_gpu(torch.from_numpy(Gm), V, dev, chunk_rows=128)
        ld = ModelLayout.row_stride(M)
        pk = np.zeros((N, ld), dtype=np.uint8)
        pk[:, :(M + 3) // 4] = O.pack2bit(Gm)
        b2 = pca_project_gpu(PackedGenotypes(torch.from_numpy(pk), N, M), V, dev, chunk_rows=1000)
        assert a.shape == (N, C) and mx(a, ref) < 2e-6 * max(1.0, float(np.abs(ref).max())) and np.array_equal(a, b2)


def test_rsvd_through_hip_kernels_matches_reference_vt():
    """svd.RSVD with both tall-skinny products on the pass-1 / pass-3 kernels (nadm_pca_project / _t, raw codes incl.
    missing = 3) against the V the reference's own RSVD produced on the demo matrix (tests/golden/demo_k3.npz), from
    the packed matrix and from the uint8 matrix; plus the two products alone against float64."""
    from neural_admixture_amd.svd import RSVD, _Rows
    from neural_admixture_amd.io import PackedGenotypes
    from neural_admixture_amd.layout import ModelLayout
    dev = _dev()
    d = np.load(f"{G}/demo_k3.npz")
    N, M = int(d["N"]), int(d["M"])
    Gm = O.unpack2bit(d["G_packed"], M)
    ld = ModelLayout.row_stride(M)
    pk = np.zeros((N, ld), dtype=np.uint8)
    pk[:, :(M + 3) // 4] = d["G_packed"]
    for data in (PackedGenotypes(torch.from_numpy(pk), N, M), torch.from_numpy(Gm)):
        Vt = RSVD(data, N, M, 8, int(d["seed"]), device=dev)
        assert Vt.shape == (8, M) and mx(Vt, d["Vt"]) < 5e-6
    rng = np.random.default_rng(1)
    Gs = O.synth_genotypes(333, 4099, 3, seed=2, missing=0.04)
    rows = _Rows(torch.from_numpy(Gs), dev)
    B = rng.standard_normal((4099, 20)).astype(np.float32)
    QT = rng.standard_normal((20, 333)).astype(np.float32)
    A64 = Gs.astype(np.float64)
    r1, r2 = A64 @ B.astype(np.float64), QT.astype(np.float64) @ A64
    assert mx(rows.a_times(B, rows=100), r1) < 2e-6 * np.abs(r1).max()
    assert mx(rows.qt_times(np.ascontiguousarray(QT.T)), r2) < 2e-6 * np.abs(r2).max()


def test_loglikelihood_kernel_matches_float64_oracle():
    """nadm_loglik (float64 reduction over the resident packed matrix) against the oracle's restatement of the
    reference's Cython loop (utils.pyx:15-40) and against the value the reference printed on the demo run."""
    from neural_admixture_amd.report import loglikelihood_hip
    from neural_admixture_amd.layout import ModelLayout
    dev = _dev()
    rng = np.random.default_rng(4)
    for N, M, K in ((70, 1500, 3), (33, 4099, 7), (20, 700, 12), (5, 3, 1), (1101, 300, 5)):      # last: every row slice of the grid has rows
        Gm = O.synth_genotypes(N, M, max(K, 2), seed=K, missing=0.05)
        P = rng.uniform(0, 1, size=(M, K)).astype(np.float32)
        P[rng.uniform(size=P.shape) < 0.05] = 0.0                     # clamped entries -> rec hits eps
        Q = rng.dirichlet(np.ones(K), size=N).astype(np.float32)
        ld = ModelLayout.row_stride(M)
        pk = np.zeros((N, ld), dtype=np.uint8)
        pk[:, :(M + 3) // 4] = O.pack2bit(Gm)
        got = loglikelihood_hip(torch.from_numpy(pk).to(dev), M, P, Q)
        ref = O.loglikelihood(Gm, P, Q)
        assert abs(got - ref) <= 1e-11 * abs(ref)
    d = np.load(f"{G}/demo_k3.npz")
    N, M = int(d["N"]), int(d["M"])
    ld = ModelLayout.row_stride(M)
    pk = np.zeros((N, ld), dtype=np.uint8)
    pk[:, :(M + 3) // 4] = d["G_packed"]
    got = loglikelihood_hip(torch.from_numpy(pk).to(dev), M, d["hi_e5_P"], d["hi_e5_Q"])
    assert abs(got - float(d["hi_e5_loglik"])) <= 1e-10 * abs(float(d["hi_e5_loglik"]))


def test_bed_to_packed_on_device_equals_the_host_converter(tmp_path):
    """nadm_bed_to_packed_dev (word-level 4x4 transposes of 2-bit fields, LDS tile, integer-atomic counts, on-device flip
    decision) against nadm_bed_to_packed on ragged shapes, with and without the allele flip, and on the demo BED."""
    import ctypes as C
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.layout import ModelLayout
    from neural_admixture_amd.io import read_bed_packed
    dev = _dev()
    rng = np.random.default_rng(17)
    for N, M, p_alt in ((1, 1, 0.2), (5, 3, 0.9), (131, 517, 0.15), (130, 2049, 0.8), (1000, 4100, 0.3),
                        (70_001, 67, 0.85)):                                # > 65535 rows with the flip: row-chunked launch
        nb = (N + 3) // 4
        # PLINK codes: 0 = hom A1 (-> 2), 1 = missing (-> 3), 2 = het (-> 1), 3 = hom A2 (-> 0); p_alt steers the mean across 1
        codes = rng.choice(4, size=(M, nb * 4), p=[p_alt * 0.9, 0.03, 0.07, 0.9 - p_alt * 0.9]).astype(np.uint8)
        bed = (codes[:, 0::4] | (codes[:, 1::4] << 2) | (codes[:, 2::4] << 4) | (codes[:, 3::4] << 6)).astype(np.uint8)
        bed = np.ascontiguousarray(bed)
        ld = ModelLayout.row_stride(M)
        ref = torch.full((N, ld), 255, dtype=torch.uint8)
        c4, fl = (C.c_int64 * 4)(), C.c_int32(0)
        check(lib.nadm_bed_to_packed(C.c_void_p(bed.ctypes.data), N, M, ptr(ref), ld, c4, 1, C.byref(fl)))
        bd = torch.from_numpy(bed).to(dev)
        out = torch.full((N, ld), 255, dtype=torch.uint8, device=dev)
        cnt = torch.full((4,), 7, dtype=torch.int64, device=dev)
        flp = torch.full((1,), 9, dtype=torch.int32, device=dev)
        check(lib.nadm_bed_to_packed_dev(ptr(bd), N, M, ptr(out), ld, ptr(cnt), 1, ptr(flp), None))
        torch.cuda.synchronize()
        assert [int(v) for v in cnt.cpu()] == [int(c4[i]) for i in range(4)]
        assert int(flp.cpu()[0]) == fl.value
        assert np.array_equal(out.cpu().numpy(), ref.numpy())
    d = np.load(f"{G}/demo_k3.npz")
    d["bed_bytes"].tofile(tmp_path / "demo.bed")
    (tmp_path / "demo.fam").write_text("\n".join(["s"] * int(d["N"])) + "\n")
    a, b2 = read_bed_packed(str(tmp_path / "demo.bed")), read_bed_packed(str(tmp_path / "demo.bed"), dev, keep_on_device=True)
    assert b2.packed.device.type == "cuda" and a.flipped == b2.flipped and np.array_equal(a.packed.numpy(), b2.packed.cpu().numpy())


def test_pack_from_host_in_many_chunks_and_with_a_row_selection():
    """Engine.pack_from_host with the rows cut into many chunks (two pinned buffers alternate, the copies run on a stream of their own
    while the next chunk is packed) and with a shard's row selection: the packed rows of the oracle, whatever the chunking."""
    import neural_admixture_amd as na
    dev = _dev()
    N, M = 1003, 2301
    Gm = O.synth_genotypes(N, M, 3, seed=9, missing=0.05)
    want = O.pack2bit(Gm)
    e = na.Engine(M, 8, 16, [3], dev, 64)
    for cr in (None, 7, 64, 500, 2000):
        e.pack_from_host(torch.from_numpy(Gm), chunk_rows=cr)
        assert np.array_equal(e.xp.cpu().numpy()[:, : want.shape[1]], want), cr
        assert not e.xp.cpu().numpy()[:, want.shape[1]:].any()
    rows = np.random.default_rng(2).permutation(N)[:333]
    e.pack_from_host(torch.from_numpy(Gm), rows=rows, chunk_rows=50)
    assert np.array_equal(e.xp.cpu().numpy()[:, : want.shape[1]], want[rows]) and e.rows_are_sharded


def test_bed_file_through_the_pinned_ring_equals_the_host_reader(tmp_path):
    """io.read_bed_packed on a file above the ring's threshold (64 MB: 16 MB pieces read straight into two pinned buffers, each copied
    to HBM while the next is read, a ragged last piece) against the host reader on the same file: the same packed rows, flip decision
    and counts -- and below the threshold (the whole-file path) as well."""
    from neural_admixture_amd.io import read_bed_packed
    dev = _dev()
    rng = np.random.default_rng(12)
    for N, M in ((4099, 70_001), (803, 1501)):
        nb = (N + 3) // 4
        raw = rng.integers(0, 256, size=M * nb, dtype=np.uint8)
        raw &= ~(raw & ~(raw >> 1) & 0x55)                              # no missing calls
        raw[:nb] |= 0x03 * (rng.integers(0, 2, size=nb, dtype=np.uint8))   # (a SNP with plenty of homozygous-alt codes all the same)
        with open(tmp_path / "x.bed", "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            raw.tofile(f)
        with open(tmp_path / "x.fam", "w") as f:
            f.write("".join(f"f{i} i{i} 0 0 0 -9\n" for i in range(N)))
        assert (M * nb >= (64 << 20)) == (N == 4099)
        a = read_bed_packed(str(tmp_path / "x.bed"), dev, keep_on_device=True)
        b = read_bed_packed(str(tmp_path / "x.bed"))
        assert (a.N, a.M, a.flipped) == (b.N, b.M, b.flipped)
        assert torch.equal(a.packed.cpu(), b.packed)


def test_cli_train_and_infer_demo(tmp_path):
    """`python -m neural_admixture_amd train|infer` on the demo BED: RSVD (GPU, from packed) + GMM init + training +
    outputs in the reference's file formats; infer reproduces Q from the saved encoder."""
    from neural_admixture_amd import cli
    _dev()
    d = np.load(f"{G}/demo_k3.npz")
    d["bed_bytes"].tofile(tmp_path / "demo.bed")
    (tmp_path / "demo.fam").write_text("\n".join(["s"] * int(d["N"])) + "\n")
    out = tmp_path / "out"
    assert cli.main(["train", "--epochs", "5", "--k", "3", "--name", "run", "--data_path", str(tmp_path / "demo.bed"),
                     "--save_dir", str(out), "--seed", "42", "--num_gpus", "1", "--threads", "1"]) == 0
    Q = np.loadtxt(out / "run.3.Q")
    P = np.loadtxt(out / "run.3.P")
    assert Q.shape == (int(d["N"]), 3) and P.shape == (int(d["M"]), 3)
    assert mx(Q, d["hi_e5_Q"]) < 5e-3 and mx(P, d["hi_e5_P"]) < 2e-2        # RSVD runs on the GPU here: V differs at 1e-6
    assert (out / "run.pt").exists() and (out / "run_config.json").exists()
    assert cli.main(["infer", "--name", "run", "--save_dir", str(out), "--out_name", "again", "--data_path", str(tmp_path / "demo.bed")]) == 0
    assert mx(np.loadtxt(out / "again.3.Q"), Q) < 1e-6


def test_cli_supervised_run_from_bed_and_pops_file(tmp_path):
    """`train --pops_path`: BED written from the supervised fixture's matrix (inverse of the reader's [2,3,1,0] recode),
    population names from a text file (src/utils.py:28-33); the result must equal the boundary call with the same inputs."""
    import neural_admixture_amd as na
    from neural_admixture_amd import cli
    from neural_admixture_amd.io import read_bed_packed
    from neural_admixture_amd.svd import RSVD
    dev = _dev()
    d = np.load(f"{G}/supervised_k4.npz")
    N, M, K = int(d["N"]), int(d["M"]), int(d["K"])
    Gm = O.unpack2bit(d["G_packed"], M)
    assert Gm[Gm != 3].mean() < 1.0 and Gm.mean() < 1.0                   # no allele flip in the reader
    inv = np.array([3, 2, 0, 1], dtype=np.uint8)                           # genotype code -> PLINK 2-bit code
    codes = inv[Gm.T]                                                      # SNP-major [M, N]
    nb = (N + 3) // 4
    pad = np.zeros((M, nb * 4), dtype=np.uint8)
    pad[:, :N] = codes
    bed = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
    (tmp_path / "s.bed").write_bytes(bytes([0x6C, 0x1B, 0x01]) + bed.tobytes())
    (tmp_path / "s.fam").write_text("\n".join(["s"] * N) + "\n")
    (tmp_path / "pops.txt").write_text("\n".join(str(a) for a in d["pops"]) + "\n")
    data = read_bed_packed(str(tmp_path / "s.bed"))
    assert np.array_equal(data.unpack_rows(0, N), Gm)
    out = tmp_path / "out"
    assert cli.main(["train", "--epochs", "3", "--k", str(K), "--name", "sup", "--data_path", str(tmp_path / "s.bed"), "--save_dir", str(out),
                     "--seed", "13", "--batch_size", "100", "--hidden_size", "128", "--pops_path", str(tmp_path / "pops.txt")]) == 0
    Q = np.loadtxt(out / "sup.4.Q", dtype=np.float32)
    V = RSVD(data, N, M, 8, 13)
    Ps, Qs, _ = na.train(3, 100, 2e-3, K, 13, data, dev, 1, 128, True, V, [str(a) for a in d["pops"]], None, None, 8)
    assert np.array_equal(Q, Qs[0])


@pytest.mark.parametrize("buckets,p3_whole,second_comm", [(1, False, False), (4, False, False), (4, True, False), (3, False, True), (8, False, False)])
def test_ddp_step_on_rccl_world1_equals_plain_step(buckets, p3_whole, second_comm):
    """The sample-sharded step (nadm_step, NADM_MODE_DP) on a ONE-rank RCCL communicator: reduce-scatter / all-gather through RCCL,
    message A on the side stream, message B bucket by bucket on the second one -- pass 3 launched range by range (or whole), the next
    pass 1 in the same ranges on streams of their own -- Adam as launches on the (whole-buffer) slices, optionally a second
    communicator for message A: must leave the bits of the single-GPU step with its fused epilogues: parameters, moments, losses."""
    from neural_admixture_amd.comm import rccl_comm
    dev = _dev()
    comm = rccl_comm(0, 1)
    comm_a = rccl_comm(0, 1) if second_comm else None
    kw = dict(mode="dp", comm=comm, n_buckets=buckets, p3_whole=p3_whole, comm_a=comm_a, debug=True)
    Gm = O.synth_genotypes(70, 2300, 4, seed=11)
    rng = np.random.default_rng(3)
    p = O.make_params(3, (rng.standard_normal((2300, 8)) / 48).astype(np.float32), rng.uniform(0.1, 0.9, (5, 2300)).astype(np.float32), 64, [5])
    e1, e2 = make_engine(Gm, p, 70), make_engine(Gm, p, 70, **kw)
    assert e2.lay.n_buckets == min(buckets, 2)                   # 2300 SNPs hold two ranges
    idx = torch.arange(70, dtype=torch.int32, device=dev)
    for _ in range(3):
        e1.train_step(idx, 70, 2e-3, True)
        e2.train_step(idx, 70, 2e-3, True)
    torch.cuda.synchronize()
    assert torch.equal(e1.big, e2.big) and torch.equal(e1.small, e2.small)
    assert e1.read_loss() == e2.read_loss()
    # several rounds of pass-2 blocks, multi-head (two pass-2 streams, ONE P message), K > 8 (two k slots), K > 16 (generic kernel),
    # C > 8 (the VALU kernels of passes 1 and 3, no batch copy: the ranges are gathered out of the resident matrix)
    for M2, ks2, C2 in ((300_000, [5], 8), (40_000, [2, 3, 4], 8), (6_000, [13], 8), (3_000, [20], 8), (9_000, [4], 12)):
        Gw = O.synth_genotypes(12, M2, 3, seed=5)
        pw = O.make_params(2, (rng.standard_normal((M2, C2)) / 500).astype(np.float32),
                           rng.uniform(0.1, 0.9, (sum(ks2), M2)).astype(np.float32), 64, ks2)
        ea, eb = make_engine(Gw, pw, 12), make_engine(Gw, pw, 12, **kw)
        assert eb.lay.n_buckets == min(buckets, (M2 + 2047) // 2048)
        ix = torch.arange(12, dtype=torch.int32, device=dev)
        for s_ in range(3):
            ea.train_step(ix, 12, 2e-3, True)
            eb.train_step(ix, 12, 2e-3, True)
            if s_ == 1:                                          # a look in between: the accessors settle both messages first
                assert torch.equal(ea.P(0), eb.P(0)) and torch.equal(ea.V(), eb.V())
        torch.cuda.synchronize()
        assert torch.equal(ea.big, eb.big) and torch.equal(ea.small, eb.small)
        assert torch.equal(ea.mbig, eb.mbig) and torch.equal(ea.vbig, eb.vbig) and torch.equal(ea.msmall, eb.msmall)
        assert ea.read_loss() == eb.read_loss()
        qa, qb = ea.infer_q(ix, 12), eb.infer_q(ix, 12)          # the encoder-only pass runs in the same parts
        assert all(torch.equal(x, y) for x, y in zip(qa, qb))
        del eb
    del e2
    comm.close()
    if comm_a is not None:
        comm_a.close()


@pytest.mark.parametrize("buckets", [1, 2])
def test_emulated_world_updates_only_rank_0s_slices(buckets):
    """nadm_comm_emulated(W) (bench.py --emulate-world): rank 0 of W ranks, no-op collectives.  After a step the parameters inside
    rank 0's slice of message A and of every bucket of message B equal the 1-rank step's with grad_scale 1/W (Adam is nearly
    scale-invariant: compare against an engine stepping with the same scale), every other parameter is untouched, and the moments
    are slice-sized."""
    from neural_admixture_amd.comm import emulated_comm
    dev = _dev()
    W = 4
    Gm = O.synth_genotypes(40, 5001, 4, seed=2)
    rng = np.random.default_rng(4)
    p = O.make_params(3, (rng.standard_normal((5001, 8)) / 55).astype(np.float32), rng.uniform(0.1, 0.9, (7, 5001)).astype(np.float32), 64, [7])
    comm = emulated_comm(W)
    e = make_engine(Gm, p, 40, mode="dp", comm=comm, n_buckets=buckets)
    ref = make_engine(Gm, p, 40)
    L = e.lay
    assert L.n_buckets == buckets
    assert e.mflat.numel() == L.slice_b + L.slice_a and L.n_flat == W * (L.slice_b + L.slice_a)
    before = e.pflat.clone()
    idx = torch.arange(40, dtype=torch.int32, device=dev)
    e.train_step(idx, 40, 2e-3, True)
    e.sync()
    ref.forward(idx, 40); ref.backward(idx, 40, True); ref.adam(2e-3, 1.0 / W)
    torch.cuda.synchronize()
    assert e.read_loss() == ref.read_loss()
    after = e.pflat
    # the reference engine has the world-1 layout (no gaps; [small | pad | V] at the same offsets): compare region by region
    assert L.off_v == ref.lay.off_v
    for j in range(L.n_buckets):
        lo, sl, hi = L.bkt_off[j], L.bkt_slice[j], L.bkt_off[j + 1]
        assert torch.equal(after[lo: lo + sl], ref.pflat[lo: lo + sl])                      # rank 0's slice of the bucket
        assert torch.equal(after[lo + sl: hi], before[lo + sl: hi])                        # the other ranks' slices
        assert torch.equal(e.mflat[L.bkt_mom[j]: L.bkt_mom[j] + sl], ref.mflat[lo: lo + sl])
    pa = ref.pflat[ref.lay.msg_a_off: ref.lay.msg_a_off + L.slice_a]
    assert torch.equal(after[L.msg_a_off: L.msg_a_off + L.slice_a], pa)
    assert torch.equal(after[L.msg_a_off + L.slice_a:], before[L.msg_a_off + L.slice_a:])
    del e
    comm.close()


# ---------------------------------------------------------------------------------------------------------------------
# world 2 with the REAL engine: two processes share cuda:0, gloo carries the (device) tensors.  RCCL refuses two ranks on
# one GPU, and the test boxes have one; the driver's N>1 bench runs are the RCCL measurement.  What this pins is the
# product's world>1 logic with the HIP kernels underneath: per-rank batches, message plan, 1/world, deferred P piece,
# final-Q gather -- against the DDP emulation captured from the reference (tests/golden/ddp_w2.npz).
def _w2_gpu_worker(rank, world, port, out_path, parallelism):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na_
    from oracle import nadm_oracle as O_
    d = np.load(os.path.join(GOLD, "ddp_w2.npz"))
    G = O_.unpack2bit(d["G_packed"], int(d["M"]))
    dev = torch.device("cuda:0")
    tr = na_.NeuralAdmixture(int(d["K"]), int(d["epochs"]), int(d["batch"]), float(d["lr"]), dev, int(d["seed"]),
                              world, rank == 0, None, None, None, loss_mode="always", parallelism=parallelism)
    Qs, Ps, model = tr.launch_training(torch.from_numpy(d["P0"]), torch.from_numpy(G), int(d["Hd"]), 8, torch.from_numpy(d["V0"]),
                                       int(d["M"]), int(d["N"]), None)
    assert type(tr.engine).__module__.startswith("neural_admixture_amd")            # the HIP engine, not a stand-in
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].cpu().numpy(),
                 losses=np.asarray([tr.epoch_losses[e] for e in range(int(d["epochs"]))]))
    else:
        assert Qs == [] and Ps == []
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("parallelism", ["dp", "snp"])
def test_world2_real_engine_on_one_gpu_matches_reference_ddp(tmp_path, parallelism):
    import torch.multiprocessing as mp
    _dev()
    port = 33500 + (os.getpid() % 2000) + (7 if parallelism == "snp" else 0)
    out = str(tmp_path / f"w2_{parallelism}.npz")
    mp.spawn(_w2_gpu_worker, args=(2, port, out, parallelism), nprocs=2, join=True)
    r = np.load(out)
    d = np.load(os.path.join(GOLD, "ddp_w2.npz"))
    assert np.abs(r["Q"] - d["Q"]).max() < 1e-4
    assert np.abs(r["P"] - d["P"]).max() < 1e-5
    assert np.abs(r["V"] - d["V"]).max() < 1e-4
    if parallelism == "dp":
        assert np.allclose(r["losses"], d["losses_rank0"].reshape(int(d["epochs"]), -1).sum(1), rtol=1e-5)


def _wN_gpu_worker(rank, world, port, out_path, buckets, second_comm, N, M, K, Hd, batch, epochs, seed, parallelism="dp"):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na_
    from test_ddp_gloo import _wN_inputs
    na_.NeuralAdmixture.dp_buckets = buckets
    na_.NeuralAdmixture.dp_second_comm = bool(second_comm)
    G, V0, P0 = _wN_inputs(N, M, K)
    dev = torch.device("cuda:0")
    tr = na_.NeuralAdmixture(K, epochs, batch, 2e-3, dev, seed, world, rank == 0, None, None, None, loss_mode="always", parallelism=parallelism)
    assert tr.batch_size == batch // world                   # neural_admixture.py:287
    Qs, Ps, model = tr.launch_training(torch.from_numpy(P0), torch.from_numpy(G), Hd, 8, torch.from_numpy(V0), M, N, None)
    e = tr.engine
    assert type(e).__module__.startswith("neural_admixture_amd")
    if parallelism == "dp":
        assert e.comm.kind == "torch" and e.moments_sharded
        assert e.lay.n_buckets == min(buckets, (M + 2047) // 2048) and (e.comm_a is not None) == bool(second_comm)
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].cpu().numpy(),
                 losses=np.asarray([tr.epoch_losses[ep] for ep in range(epochs)]))
    dist.barrier()
    dist.destroy_process_group()


def test_torch_transport_names_the_null_stream_as_the_default_stream():
    """A zero stream handle (the step was called on the default stream: what Engine.train_step passes unless the caller set another
    current stream) must map to torch's default stream.  ExternalStream(0) is a POOL stream -- torch takes the zero for "no pointer" --
    and a collective issued there is ordered against nothing the step launched: 4 and 8 ranks sharing a GPU applied stale or
    half-summed gradients to message B that way (2 ranks passed by timing)."""
    from neural_admixture_amd.comm import _TorchTransport
    dev = torch.device("cuda", 0)
    s0 = _TorchTransport.torch_stream(0, dev)
    assert s0.cuda_stream == 0 and s0 == torch.cuda.default_stream(dev)
    assert _TorchTransport.torch_stream(None, dev).cuda_stream == 0
    side = torch.cuda.Stream(dev)
    assert _TorchTransport.torch_stream(side.cuda_stream, dev).cuda_stream == side.cuda_stream
    print("ExternalStream(0).cuda_stream =", torch.cuda.ExternalStream(0, device=dev).cuda_stream)


@pytest.mark.parametrize("world,buckets,second_comm,K,parallelism", [(8, 1, False, 3, "dp"), (4, 3, True, 3, "dp"), (4, 1, False, 3, "dp"),
                                                                     (8, 1, False, 16, "dp"), (4, 1, False, 3, "snp"), (8, 1, False, 9, "snp")])
def test_worldN_real_engine_on_one_gpu_at_the_reference_batch_semantics(tmp_path, world, buckets, second_comm, K, parallelism):
    """configs[3]'s 8-GPU FORM with the real HIP engine: --batch_size 800 over W ranks = 800 // W rows per rank and step (100 at W = 8,
    neural_admixture.py:287), N = 1003 not a multiple of W (the DistributedSampler wraps), a ragged second step, every message cut into W
    slices with sharded moments (and, second case, message B in three buckets + message A on a transport of its own) -- W processes
    share cuda:0, gloo callbacks carry the collectives -- against the oracle's DDP emulation of the same run (pinned by ddp_w2 / ddp_w4
    from the reference).  K = 16: configs[4]'s model (the two-k-slot variant of pass 2) in its 8-rank form.  "snp": the SNP-sharded mode
    at 4 and 8 ranks (every rank all rows of its SNP range, two small all-reduces per step; same mathematics, so the same expected values;
    its per-rank loss is a partial sum and is not compared).  RCCL refuses several ranks on one device; tests/test_multi_gpu_rccl.py is the
    same over RCCL where GPUs exist."""
    import sys
    import torch.multiprocessing as mp
    _dev()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ddp_gloo import _wN_inputs
    N, M, Hd, batch, epochs, seed = 1003, 6200, 32, 800, 2, 5
    port = 36500 + (os.getpid() % 2000) + 11 * world + K + (5 if parallelism == "snp" else 0)
    out = str(tmp_path / f"w{world}.npz")
    mp.spawn(_wN_gpu_worker, args=(world, port, out, buckets, second_comm, N, M, K, Hd, batch, epochs, seed, parallelism), nprocs=world, join=True)
    r = np.load(out)
    G_, V0, P0 = _wN_inputs(N, M, K)
    p = O.make_params(seed, V0.copy(), P0.copy(), Hd, [K])
    p, Qs, losses = O.train_run(G_, p, epochs, batch, 2e-3, seed, world=world)
    assert np.abs(r["Q"] - Qs[0]).max() < 1e-4
    assert np.abs(r["P"] - p.P[0]).max() < 1e-5
    assert np.abs(r["V"] - p.V).max() < 1e-4
    if parallelism == "dp":
        assert np.allclose(r["losses"], losses, rtol=1e-5)


def _w2_train_worker(rank, world, port, out_path):
    """Both ranks call the drop-in boundary train(...): master-only GMM init, barrier, dist.broadcast(P_init / V)
    (model/train.py:86-113), sharded training on the HIP engine, master-only outputs."""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_admixture_amd as na_
    from oracle import nadm_oracle as O_
    d = np.load(os.path.join(GOLD, "ddp_w2.npz"))
    Gm = O_.unpack2bit(d["G_packed"], int(d["M"]))
    V_CM = np.ascontiguousarray(d["V0"].T)
    if rank != 0:
        V_CM = np.zeros_like(V_CM)                          # only the broadcast can give rank 1 the master's V
    dev = torch.device("cuda:0")
    Ps, Qs, model = na_.train(int(d["epochs"]), int(d["batch"]), float(d["lr"]), int(d["K"]), int(d["seed"]), torch.from_numpy(Gm), dev,
                              world, int(d["Hd"]), rank == 0, V_CM, None, None, None, 8)
    assert type(model.engine).__module__.startswith("neural_admixture_amd")
    if rank == 0:
        np.savez(out_path, Q=Qs[0], P=Ps[0], V=model.state_dict()["V"].cpu().numpy())
    else:
        assert Ps == [] and Qs == []
    ref = model.engine.small.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, model.engine.small)             # same parameters on both ranks at the end
    dist.barrier()
    dist.destroy_process_group()


def test_world2_train_boundary_runs_the_init_broadcasts_on_the_hip_engine(tmp_path):
    import torch.multiprocessing as mp
    from neural_admixture_amd.train import gmm_p_init
    _dev()
    port = 35500 + (os.getpid() % 2000)
    out = str(tmp_path / "w2_train.npz")
    mp.spawn(_w2_train_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    d = np.load(os.path.join(GOLD, "ddp_w2.npz"))
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    P_init = gmm_p_init(Gm, np.ascontiguousarray(d["V0"].T), int(d["K"]), None, None, 8, int(d["seed"]), None)   # host projection + sklearn
    p = O.make_params(int(d["seed"]), d["V0"].copy(), P_init.astype(np.float32), int(d["Hd"]), [int(d["K"])])
    p, Qs, _ = O.train_run(Gm, p, int(d["epochs"]), int(d["batch"]), float(d["lr"]), int(d["seed"]), world=2)
    assert np.abs(r["Q"] - Qs[0]).max() < 2e-4 and np.abs(r["P"] - p.P[0]).max() < 2e-4 and np.abs(r["V"] - p.V).max() < 2e-4


def test_cli_train_num_gpus_2_through_its_spawn_worker(tmp_path):
    """`train --num_gpus 2`: the parent reads the BED and runs the RSVD, then mp.spawn's one worker per rank (entry.py:186-190)
    with the packed matrix in shared memory; the workers set up the process group, call train() and the master writes the
    outputs.  On the one-GPU test box the ranks share cuda:0 over gloo (--share_gpu); the result must equal the world-2
    boundary call made directly."""
    from neural_admixture_amd import cli
    _dev()
    d = np.load(f"{G}/demo_k3.npz")
    d["bed_bytes"].tofile(tmp_path / "demo.bed")
    (tmp_path / "demo.fam").write_text("\n".join(["s"] * int(d["N"])) + "\n")
    out = tmp_path / "out"
    os.environ["MASTER_PORT"] = str(37500 + os.getpid() % 2000)
    try:
        assert cli.main(["train", "--epochs", "5", "--k", "3", "--name", "run2", "--data_path", str(tmp_path / "demo.bed"), "--save_dir", str(out),
                         "--seed", "42", "--num_gpus", "2", "--share_gpu", "--batch_size", "64", "--threads", "1"]) == 0
        Q2 = np.loadtxt(out / "run2.3.Q")
        P2 = np.loadtxt(out / "run2.3.P")
        assert Q2.shape == (int(d["N"]), 3) and P2.shape == (int(d["M"]), 3)
        assert (out / "run2.pt").exists() and (out / "run2_config.json").exists()
        assert np.abs(Q2.sum(axis=1) - 1).max() < 1e-5 and P2.min() >= 0 and P2.max() <= 1
        # the same run on one rank with the same GLOBAL batch sees the same samples per step only if the shards line up; what
        # must hold whatever the order: the two-rank run is a valid training run that ends close to the single-rank one
        os.environ["MASTER_PORT"] = str(37600 + os.getpid() % 2000)
        assert cli.main(["train", "--epochs", "5", "--k", "3", "--name", "run1", "--data_path", str(tmp_path / "demo.bed"), "--save_dir", str(out),
                         "--seed", "42", "--num_gpus", "1", "--batch_size", "64", "--threads", "1"]) == 0
        Q1 = np.loadtxt(out / "run1.3.Q")
        assert np.abs(Q1 - Q2).mean() < 0.05
    finally:
        os.environ.pop("MASTER_PORT", None)


def test_rows_at_offsets_beyond_4_gib_in_a_resident_matrix_of_configs4_size():
    """BASELINE configs[4]: 500k samples x 1M SNPs stay packed in HBM -- 125 GB, which ONE MI355X holds.  Rows of such a
    matrix start at byte offsets far beyond 2^32; a step that gathers rows scattered over the whole allocation (first
    row, the rows either side of the 2^31- and 2^32-byte marks, the last row) must give bit-identical results to the same
    rows compacted into a small matrix.  Only the gathered rows are written; the rest of the allocation is never read."""
    import neural_admixture_amd as na
    from neural_admixture_amd._lib import lib, check, ptr
    dev = _dev()
    N, M, K, b = 500_000, 1_000_000, 16, 64
    e_big, e_small = na.Engine(M, 8, 128, [K], dev, b), na.Engine(M, 8, 128, [K], dev, b)
    free, _total = torch.cuda.mem_get_info()
    if free < N * e_big.ld + (8 << 30):
        pytest.skip("needs 125 GB of free HBM")
    ld = e_big.ld
    marks = [0, 1, (1 << 31) // ld, (1 << 31) // ld + 1, (1 << 32) // ld, (1 << 32) // ld + 1, (1 << 36) // ld + 1, N - 2, N - 1]
    rng = np.random.default_rng(8)
    rows = np.unique(np.concatenate([marks, rng.integers(0, N, size=b - len(marks))]))
    while rows.size < b:
        rows = np.unique(np.concatenate([rows, rng.integers(0, N, size=b - rows.size)]))
    rows = rng.permutation(rows).astype(np.int64)                       # gather order is not sorted either
    assert int(rows.max()) * ld > (1 << 36)
    g = torch.Generator(device="cpu").manual_seed(1)
    Qt = torch.distributions.Dirichlet(torch.full((K,), 0.3)).sample((b,)).float().to(dev)
    Fq = (0.5 * torch.rand(K, M, generator=g)).clamp(0.005, 0.5).to(dev)
    big = torch.empty((N, ld), dtype=torch.uint8, device=dev)
    small = torch.empty((b, ld), dtype=torch.uint8, device=dev)
    for j, r in enumerate(rows):
        check(lib.nadm_synth_packed(ptr(big[int(r):]), 1, int(r), M, ld, ptr(Qt[j:]), ptr(Fq), K, 0.01, 99, None))
        check(lib.nadm_synth_packed(ptr(small[j:]), 1, int(r), M, ld, ptr(Qt[j:]), ptr(Fq), K, 0.01, 99, None))
    torch.cuda.synchronize()
    assert torch.equal(big[torch.from_numpy(rows).to(dev)], small)
    V = (rng.standard_normal((M, 8)) / 1000).astype(np.float32)
    P = rng.uniform(0.05, 0.95, (K, M)).astype(np.float32)
    from neural_admixture_amd.model import init_encoder_weights
    sm = init_encoder_weights(5, 8, 128, [K])
    for e, xp in ((e_big, big), (e_small, small)):
        e.set_packed(xp)
        e.load_params(V, P, sm)
    ib = torch.from_numpy(rows.astype(np.int32)).to(dev)
    isml = torch.arange(b, dtype=torch.int32, device=dev)
    for _ in range(2):
        e_big.train_step(ib, b, 2e-3, True)
        e_small.train_step(isml, b, 2e-3, True)
    torch.cuda.synchronize()
    assert torch.equal(e_big.big, e_small.big) and torch.equal(e_big.small, e_small.small)
    assert e_big.read_loss() == e_small.read_loss()
    qa, qb = e_big.infer_q(ib, b)[0], e_small.infer_q(isml, b)[0]
    assert torch.equal(qa, qb) and abs(float(qa.sum()) - b) < 1e-3


def test_mixture_means_on_the_gpu_equal_the_library_fit():
    """train.gmm_p_init on a GPU device fits the mixture with _gmm_em (float64 device ops) for large N; fit="sklearn" selects the
    library fit the reference calls (train.py:61).  Same means -> same P init."""
    from neural_admixture_amd._gmm_em import fit_means as em
    from neural_admixture_amd._gmm_fit import fit_means as sk
    from neural_admixture_amd.train import gmm_p_init
    dev = _dev()
    rng = np.random.default_rng(4)
    for N, k, seed in ((4000, 8, 42), (900, 3, 7)):
        cent = rng.standard_normal((k, 8))
        X = (rng.dirichlet(np.full(k, 0.4), N) @ cent + 0.25 * rng.standard_normal((N, 8))).astype(np.float32).astype(np.float64)
        assert np.abs(em(X, k, seed, dev) - sk(X, k, seed)).max() < 1e-9
    Gm = O.synth_genotypes(300, 4000, 3, seed=2)
    V = np.linalg.svd(Gm.astype(np.float32), full_matrices=False)[2][:8].astype(np.float32)
    res = {}
    for how in ("em", "sklearn", "native", "auto"):
        res[how] = gmm_p_init(Gm, V, None, 2, 4, 8, 42, dev, fit=how)
    a, b = res["em"], res["sklearn"]
    assert a.shape == b.shape == (9, 4000) and np.abs(a - b).max() < 1e-6
    assert np.array_equal(res["auto"], res["native"]) and np.abs(res["native"] - b).max() < 1e-6      # N = 300: "auto" is the host restatement (r05)


def test_mixture_fit_with_the_sums_on_the_device_equals_the_host_form_and_the_library():
    """csrc/nadm_gmm_dev.hip (the EM of the reference's GaussianMixture call, model/train.py:61-66, with its sums over the samples in
    HIP kernels) against csrc/nadm_gmm.cpp (host threads) on the same seeding draws: same winner, same iteration count, means to 1e-9;
    against scikit-learn itself on a small case; the library's error for a collapsed input; what "auto" picks."""
    from neural_admixture_amd import gmm
    from neural_admixture_amd._gmm_fit import fit_means as sk
    from neural_admixture_amd.train import gmm_p_init
    _dev()
    rng = np.random.default_rng(11)
    st = torch.cuda.current_stream().cuda_stream
    for N, k, seed, spread in ((30_000, 8, 42, 0.05), (21_001, 5, 3, 0.25), (2504, 7, 42, 0.25), (700, 16, 1, 0.3), (64, 1, 5, 0.3)):
        cent = rng.standard_normal((k, 8))
        X = (rng.dirichlet(np.full(k, 0.3), N) @ cent + spread * rng.standard_normal((N, 8))).astype(np.float32).astype(np.float64)
        host = gmm.fit_means(X, k, seed)
        h = dict(gmm.fit_means.last)
        dev_m = gmm.fit_means(X, k, seed, stream=st)
        d = dict(gmm.fit_means.last)
        assert d["device"] and not h["device"]
        assert d["n_iter"] == h["n_iter"] and abs(d["lower_bound"] - h["lower_bound"]) < 1e-10
        assert np.abs(dev_m - host).max() < 1e-9, (N, k)
        if N <= 2504:
            assert np.abs(dev_m - sk(X, k, seed)).max() < 1e-9
        again = gmm.fit_means(X, k, seed, stream=st)
        assert np.array_equal(again, dev_m)                                            # fixed summation order: the same bits every time
    Xc = np.zeros((500, 8))                                                            # every sample the same point: no covariance is positive definite
    Xc[:, 0] = 1.0
    with pytest.raises(ValueError, match="ill-defined empirical covariance"):
        gmm.fit_means(Xc, 3, 0, reg_covar=0.0, stream=st)
    with pytest.raises(RuntimeError, match="d must be 8"):
        gmm.fit_means(np.zeros((50, 4)), 2, 0, stream=st)
    assert gmm.device_form_applies(100_000, 8, 8) and not gmm.device_form_applies(2504, 8, 7)
    assert not gmm.device_form_applies(100_000, 8, 17) and not gmm.device_form_applies(100_000, 6, 8)
    Gm = O.synth_genotypes(300, 4000, 3, seed=2)
    V = np.linalg.svd(Gm.astype(np.float32), full_matrices=False)[2][:8].astype(np.float32)
    a = gmm_p_init(Gm, V, 3, None, None, 8, 42, torch.device("cuda:0"), fit="device")
    b = gmm_p_init(Gm, V, 3, None, None, 8, 42, torch.device("cuda:0"), fit="native")
    assert a.shape == b.shape and np.abs(a - b).max() < 1e-7


@pytest.mark.parametrize("ks,b,M", [([8], 800, 6200), ([5], 333, 3000), ([12], 400, 2301), ([3, 9], 800, 5000)])
def test_pass2_in_sample_slices_gives_the_unsliced_gradients_and_is_reproducible(request, ks, b, M):
    """nadm_decode_bce_sliced (the batch's sample tiles dealt to S blocks per SNP chunk, the partial dP sums added by the block counted
    last): against the S = 1 kernel on the same inputs -- dQ-driven gradients and dP equal to rounding (the sum over the slices has an
    order of its own), the loss value to 1e-6 -- for S = 2, 3, 4 and the library's own choice; the same bits call after call; counters
    back at zero; the fused step (Adam in the last block's epilogue) against the unsliced step after three steps; ragged batches.
    (Forcing S needs the test build of the library: conftest.in_hook_build.)"""
    from conftest import in_hook_build
    if not in_hook_build(request):
        return
    from neural_admixture_amd._lib import lib
    N = max(b + 40, 200)
    Gm = O.synth_genotypes(N, M, max(ks), seed=31)
    rng = np.random.default_rng(8)
    Ps = [rng.uniform(0.02, 0.98, (k, M)).astype(np.float32) for k in ks]
    Ps[0][:, ::7] = 0.0                                                    # clamped entries
    p = O.make_params(5, (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32), np.concatenate(Ps, 0), 64, ks)
    idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(_dev())

    def run(force):
        lib.nadm_test_force_slices(force)
        try:
            e = make_engine(Gm, p, b)
            want = [int(lib.nadm_decode_slices(b, M, kp)) for kp in e.lay.kp]
            e.forward(idx, b)
            e.backward(idx, b)
            torch.cuda.synchronize()
            g = engine_grads(e)
            loss = e.read_loss(reset=True)[0]
            e.forward(idx, b)
            e.backward(idx, b)
            g2 = engine_grads(e)
            assert all(np.array_equal(g[k_], g2[k_]) for k_ in g), "not reproducible"
            assert e._p2_cnt is None or int(e._p2_cnt.abs().sum().item()) == 0
            e.read_loss(reset=True)
            for bb in (b, b - 37, b):                                       # the step: Adam in the epilogue of the block that is counted last
                e.train_step(idx[:bb], bb, 2e-3, True)
            torch.cuda.synchronize()
            assert e._p2_cnt is None or int(e._p2_cnt.abs().sum().item()) == 0
            return g, loss, e.pflat.clone(), want
        finally:
            lib.nadm_test_force_slices(0)

    g1, loss1, p1, want1 = run(1)
    assert want1 == [1] * len(ks)
    for force in (2, 3, 4, 0):
        g, loss, pf, want = run(force)
        if force:
            assert all(w == min(force, (b + 63) // 64) or w <= force for w in want) and max(want) > 1
        for k_ in g1:
            scale = np.abs(g1[k_]).max() + 1e-30
            assert np.abs(g[k_] - g1[k_]).max() <= 2e-6 * scale, (force, k_)
        assert abs(loss - loss1) <= 1e-6 * abs(loss1)
        assert (pf - p1).abs().max().item() <= 2e-5               # three Adam steps on gradients equal to rounding (a flipped sign of a ~0 gradient moves lr)


@pytest.mark.parametrize("ks,b,M,C", [([8], 1100, 3000, 8), ([3], 1283, 5003, 8), ([5], 4200, 1100, 4)])
def test_pass3_in_sample_slices_gives_the_unsliced_gradient_and_is_reproducible(request, ks, b, M, C):
    """nadm_encode_bwd_sliced (r06: the batch's 128-sample tiles dealt to S blocks per 512-SNP chunk, the partial dV sums added in slice
    order by the block counted last): against the S = 1 kernel on the same dZ -- dV equal to rounding (the sum over the slices has an order
    of its own) -- for S = 2, 3, 5 and the library's own choice; the same bits call after call; counters back at zero; the fused step
    (Adam on V in the last block's epilogue, the MLP weight-gradient side blocks once) equal to the unsliced step after three steps to the
    size of an Adam step on a rounding-level gradient; ragged batches.  (Forcing S needs the test build: conftest.in_hook_build.)"""
    from conftest import in_hook_build
    if not in_hook_build(request):
        return
    from neural_admixture_amd._lib import lib
    N = b + 60
    Gm = O.synth_genotypes(N, M, max(ks), seed=17, missing=0.03)
    rng = np.random.default_rng(9)
    p = O.make_params(5, (rng.standard_normal((M, C)) / np.sqrt(M)).astype(np.float32), rng.uniform(0.02, 0.98, (sum(ks), M)).astype(np.float32), 64, ks)
    idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(_dev())

    def run(force):
        lib.nadm_test_force_p3_slices(force)
        try:
            e = make_engine(Gm, p, b)
            want = int(lib.nadm_encode_slices(b, M, e.lay.CP))
            e.forward(idx, b)
            e.backward(idx, b)
            torch.cuda.synchronize()
            gV = e.gV().cpu().numpy().copy()
            e.forward(idx, b)
            e.backward(idx, b)
            assert np.array_equal(gV, e.gV().cpu().numpy()), "not reproducible"
            assert e._p3_cnt is None or int(e._p3_cnt.abs().sum().item()) == 0
            for bb in (b, b - 137, b):
                e.train_step(idx[:bb], bb, 2e-3, True)
            e.sync()
            torch.cuda.synchronize()
            assert e._p3_cnt is None or int(e._p3_cnt.abs().sum().item()) == 0
            return gV, e.pflat.clone(), want
        finally:
            lib.nadm_test_force_p3_slices(0)

    g1, p1, want1 = run(1)
    assert want1 == 1
    for force in (2, 3, 5, 0):
        g, pf, want = run(force)
        tiles = (b + 127) // 128
        assert (want == min(force, tiles) or 1 < want <= force) if force else want == int(lib.nadm_encode_slices(b, M, 8 if C > 4 else 4))
        assert np.abs(g - g1).max() <= 2e-6 * (np.abs(g1).max() + 1e-30), force
        assert (pf - p1).abs().max().item() <= 2e-5


@pytest.mark.parametrize("K", [5, 13, 20])
def test_pass2_gather_byproduct_and_pass3_on_the_compact_copy(K):
    """nadm_decode_bce_gather = nadm_decode_bce + the batch's rows written back to back: same gradients / loss bit for bit,
    the copy equals the gathered rows on every byte column that holds SNPs, and pass 3 on (copy, 0..b-1) gives the bits of
    pass 3 on (resident matrix, idx).  K = 5 / 13: the bf16 kernel writes the copy itself; K = 20: separate gather kernel.
    M = 2301 leaves a ragged last byte; the SNP sub-range form (pointer + m0/4) is covered as well."""
    import ctypes as C
    import neural_admixture_amd as na
    from neural_admixture_amd._lib import lib, check, ptr
    dev = _dev()
    N, M, b = 90, 2301, 37
    Gm = O.synth_genotypes(N, M, 4, seed=21)
    rng = np.random.default_rng(6)
    p = O.make_params(3, (rng.standard_normal((M, 8)) / 48).astype(np.float32), rng.uniform(0.05, 0.95, (K, M)).astype(np.float32), 64, [K])
    e = make_engine(Gm, p, b)
    idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(dev)
    e.forward(idx, b)
    L = e.lay
    kp = L.kp[0]
    nch = int(lib.nadm_decode_chunks(M, kp))
    outs = []
    for gather in (False, True):
        dP = torch.zeros(M * kp, dtype=torch.float32, device=dev)
        dq = torch.zeros(nch * b * kp, dtype=torch.float32, device=dev)
        ls = torch.zeros(nch, dtype=torch.float32, device=dev)
        xg = torch.full((int(lib.nadm_batch_copy_bytes(b, M)),), 0xEE, dtype=torch.uint8, device=dev)
        args = (ptr(e.xp), e.ld, ptr(idx), b, M, C.c_void_p(e.big.data_ptr() + L.p_off[0] * 4), kp, ptr(e.Q), L.SP, ptr(dP), ptr(dq), ptr(ls), 1)
        if gather:
            check(lib.nadm_decode_bce_gather(*args, ptr(xg), None))
        else:
            check(lib.nadm_decode_bce(*args, None))
        torch.cuda.synchronize()
        outs.append((dP, dq, ls, xg))
    for a_, b_ in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a_, b_)
    nbytes = (M + 3) // 4
    want = e.xp[idx.long()]                                     # the copy holds the model's input: missing calls (code 3) are 0
    miss = want & (want >> 1) & 0x55
    want = want & ~(miss * 3)
    # ... tiled by pass 3's chunks: byte column c of batch row i at (c // 128) * b * 128 + i * 128 + c % 128 (include/nadm.h)
    got = outs[1][3].view(-1, b, 128).permute(1, 0, 2).reshape(b, -1)
    assert torch.equal(got[:, :nbytes], want[:, :nbytes])
    assert bool((outs[0][3] == 0xEE).all())                      # the plain entry point leaves xg alone
    # pass 3: resident matrix + idx vs compact copy + iota
    dZ = torch.from_numpy(rng.standard_normal((b, L.CP)).astype(np.float32)).to(dev)
    iota = torch.arange(b, dtype=torch.int32, device=dev)
    dv = []
    dzimg = torch.empty(int(lib.nadm_dz_image_bytes(b)), dtype=torch.uint8, device=dev)
    check(lib.nadm_dz_image(ptr(dZ), b, L.CP, ptr(dzimg), None))
    for src, rows, fl in ((e.xp, idx, 0), (outs[1][3], iota, 1)):
        o = torch.zeros(M * L.CP, dtype=torch.float32, device=dev)
        check(lib.nadm_encode_bwd(ptr(src), e.ld, ptr(rows), b, M, ptr(dZ), ptr(dzimg), L.CP, ptr(o), fl, None))
        dv.append(o)
    torch.cuda.synchronize()
    assert torch.equal(dv[0], dv[1])
    # whole steps: the production step (pass 3 on the copy, fused epilogues) vs the plain phases with pass 3 gathering from the resident
    # matrix (K <= 16: the copy is dropped by hand), two steps, bit-identical state
    e1, e2 = make_engine(Gm, p, b), make_engine(Gm, p, b)
    for _ in range(2):
        e1.train_step(idx, b, 2e-3, True)
        e2.forward(idx, b)
        n_loss = e2.decode_all(idx, b, True)
        e2._xg_key = None
        e2.mlp_backward(b, n_loss)
        e2.encode_backward(idx, b)
        e2.adam(2e-3)
    torch.cuda.synchronize()
    assert torch.equal(e1.big, e2.big) and torch.equal(e1.small, e2.small) and e1.read_loss() == e2.read_loss()


@pytest.mark.parametrize("ks,C", [([5], 8), ([13], 8), ([20], 8), ([2, 3, 4], 8), ([6], 12)])
def test_adam_in_the_epilogues_of_passes_2_and_3_equals_the_separate_launches(ks, C):
    """Single-GPU step: nadm_decode_bce_step / nadm_encode_bwd_step apply Adam (+ clamp of P) to the rows whose gradient the
    block has just completed; the step must leave parameters and Adam moments bit-identical to passes + nadm_adam.  Covers
    the matrix-core kernels (K <= 8, K 9..16), the variants that run the update as a second kernel (K = 20, C = 12), and
    several heads; M = 2301 is ragged, b = 37 leaves a partial sample tile."""
    dev = _dev()
    N, M, b = 90, 2301, 37
    Gm = O.synth_genotypes(N, M, 4, seed=31)
    rng = np.random.default_rng(9)
    p = O.make_params(3, (rng.standard_normal((M, C)) / 48).astype(np.float32), rng.uniform(0.0, 1.0, (sum(ks), M)).astype(np.float32), 64, ks)
    e1, e2 = make_engine(Gm, p, b), make_engine(Gm, p, b)
    for s in range(4):
        idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(dev)
        e1.train_step(idx, b, 2e-3, s % 2 == 0)
        e2.forward(idx, b); e2.backward(idx, b, s % 2 == 0); e2.adam(2e-3)      # passes + nadm_adam
    torch.cuda.synchronize()
    assert e1.step_count == e2.step_count == 4
    assert torch.equal(e1.big, e2.big) and torch.equal(e1.mbig, e2.mbig) and torch.equal(e1.vbig, e2.vbig)
    assert torch.equal(e1.small, e2.small) and e1.read_loss() == e2.read_loss()
    assert float(e1.P(0).min()) >= 0.0 and float(e1.P(0).max()) <= 1.0


def test_production_step_equals_the_plain_phases_on_random_shapes():
    """nadm_step (fused epilogues, operand images, the small update riding in the next pass 1, two pass-2 streams for several heads,
    the fallbacks for C > 8 / K > 16 / wide hidden layers) against forward -> backward -> adam on a twin engine: 30 random shapes,
    three steps each with the loss value on and off, a batch that shrinks on the way -- parameters, moments and loss sums bit for bit.
    The sample-sharded sequence (NADM_MODE_DP, one rank, no-op transport: gradients written out, Adam as launches on the slices, message
    A on its side stream) must leave the same bits too."""
    from neural_admixture_amd.comm import emulated_comm
    dev = _dev()
    one_rank = emulated_comm(1)
    rng = np.random.default_rng(2024)
    shapes = [(1, 5, [2], 8, 8), (2, 1023, [3], 32, 8), (33, 2500, [16], 64, 12), (20, 1500, [20], 32, 8), (10, 900, [33], 32, 16),
              (9, 800, [5], 2304, 8), (70, 2600, [2, 3, 4, 5, 6, 7, 8, 9, 10], 64, 8), (900, 1300, [4], 64, 8), (40, 3000, [9, 20], 64, 4)]
    while len(shapes) < 30:
        nh = int(rng.integers(1, 4))
        ks = sorted(set(int(k) for k in rng.integers(2, 25, size=nh)))
        shapes.append((int(rng.integers(1, 300)), int(rng.integers(4, 9000)), ks, int(rng.choice([8, 32, 64, 96, 256, 1024])), int(rng.choice([4, 8, 8, 8, 12]))))
    for N, M, ks, Hd, C in shapes:
        Gm = O.synth_genotypes(N, M, max(2, min(max(ks), 6)), seed=N + M, missing=0.03)
        V0 = (rng.standard_normal((M, C)) / np.sqrt(M)).astype(np.float32)
        P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
        p = O.make_params(N, V0, P0, Hd, ks)
        e1, e2 = make_engine(Gm, p, N), make_engine(Gm, p, N)
        e3 = make_engine(Gm, p, N, mode="dp", comm=one_rank)         # the sample-sharded sequence of launches, one rank: the same bits
        for s_ in range(3):
            b = N if s_ != 1 else max(1, N - N // 3)                 # the middle step on a shorter batch
            idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(dev)
            e1.train_step(idx, b, 2e-3, s_ != 1)
            e2.forward(idx, b); e2.backward(idx, b, s_ != 1); e2.adam(2e-3)
            e3.train_step(idx, b, 2e-3, s_ != 1)
        torch.cuda.synchronize()
        what = (N, M, ks, Hd, C)
        loss2 = e2.read_loss()
        for e in (e1, e3):
            assert e.step_count == e2.step_count == 3, what
            assert torch.equal(e.big, e2.big) and torch.equal(e.small, e2.small), what
            assert torch.equal(e.mbig, e2.mbig) and torch.equal(e.vbig, e2.vbig) and torch.equal(e.msmall, e2.msmall), what
            assert e.read_loss() == loss2, what
        del e1, e2, e3
    one_rank.close()


@pytest.mark.parametrize("M,K", [(500_000, 8), (600_000, 7)])
def test_full_size_properties_of_the_step(M, K):
    """BASELINE configs[3] width (M = 500k, b = 800, K = 8) and configs[1] width (M = 600k, K = 7), properties that do not need
    an oracle of that size:
    (1) equivariance: permuting the batch permutes Z / Q rows bit for bit (a row's sums do not depend on its tile position)
        and leaves the loss and dP / dV unchanged up to the summation order over samples;
    (2) additivity: loss, dP and dV of the batch = the sums over its two halves (BCE(sum) is a sum over samples);
    (3) every Q row is a distribution per head; P stays in [0, 1] after an update; pad columns stay zero."""
    import neural_admixture_amd as na
    from neural_admixture_amd._lib import lib, check, ptr
    dev = _dev()
    b = 800
    e = na.Engine(M, 8, 1024, [K], dev, b)
    g = torch.Generator(device="cpu").manual_seed(3)
    Qt = torch.distributions.Dirichlet(torch.full((K,), 0.3)).sample((b,)).float().to(dev)
    Fq = (0.5 * torch.rand(K, M, generator=g)).clamp(0.005, 0.5).to(dev)
    xp = torch.empty((b, e.ld), dtype=torch.uint8, device=dev)
    check(lib.nadm_synth_packed(ptr(xp), b, 0, M, e.ld, ptr(Qt), ptr(Fq), K, 0.02, 77, None))
    e.set_packed(xp)
    e.load_params((torch.randn(M, 8, generator=g) / M ** 0.5).numpy(), torch.rand(K, M, generator=g).mul(0.98).add(0.01).numpy(),
                  na.model.init_encoder_weights(7, 8, 1024, [K]))
    L = e.lay

    def run(idx_t, n):
        e.forward(idx_t, n)
        Z = e.Z[: n * L.CP].view(n, L.CP).clone()
        Q = e.Q[: n * L.SP].view(n, L.SP).clone()
        e.backward(idx_t, n, True)
        torch.cuda.synchronize()
        return Z, Q, e.read_loss()[1], e.gP(0).clone(), e.gV().clone()

    ident = torch.arange(b, dtype=torch.int32, device=dev)
    Z0, Q0, l0, dP0, dV0 = run(ident, b)
    perm = torch.randperm(b, generator=g).to(torch.int32).to(dev)
    Z1, Q1, l1, dP1, dV1 = run(perm, b)
    assert torch.equal(Z1, Z0[perm.long()]) and torch.equal(Q1, Q0[perm.long()])
    assert abs(l1 - l0) / abs(l0) < 1e-6
    sP, sV = float(dP0.abs().max()), float(dV0.abs().max())
    assert float((dP1 - dP0).abs().max()) < 2e-5 * sP and float((dV1 - dV0).abs().max()) < 2e-5 * sV
    h = b // 2
    _, _, la, dPa, dVa = run(ident[:h].contiguous(), h)
    _, _, lb, dPb, dVb = run(ident[h:].contiguous(), b - h)
    assert abs((la + lb) - l0) / abs(l0) < 1e-6
    assert float((dPa + dPb - dP0).abs().max()) < 2e-5 * sP and float((dVa + dVb - dV0).abs().max()) < 2e-5 * sV
    assert float((Q0[:, :K].sum(dim=1) - 1).abs().max()) < 1e-5 and float(Q0.min()) >= 0.0
    assert not bool(e.gbig[: M * L.CP].view(M, L.CP)[:, L.C:].any())
    e.adam(2e-3)
    torch.cuda.synchronize()
    assert float(e.P(0).min()) >= 0.0 and float(e.P(0).max()) <= 1.0


def test_cli_train_from_vcf_equals_the_boundary_call(tmp_path):
    """`train --data_path x.vcf.gz`: the supervised fixture's matrix written as a VCF (GT 0/0, 0/1, 1|1, ./.), read by
    io.read_vcf_packed (nadm_vcf_parse_gt); the run must equal train() on the matrix itself."""
    import gzip
    import neural_admixture_amd as na
    from neural_admixture_amd import cli
    from neural_admixture_amd.io import read_vcf_packed
    from neural_admixture_amd.svd import RSVD
    dev = _dev()
    d = np.load(f"{G}/supervised_k4.npz")
    N, M, K = int(d["N"]), int(d["M"]), int(d["K"])
    Gm = O.unpack2bit(d["G_packed"], M)
    assert Gm.mean() < 1.0                                                   # no allele flip in the reader
    gt = np.array(["0/0", "0/1", "1|1", "./."])
    with gzip.open(tmp_path / "x.vcf.gz", "wt") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"s{i}" for i in range(N)) + "\n")
        for j in range(M):
            f.write(f"1\t{j + 1}\t.\tA\tG\t.\t.\t.\tGT\t" + "\t".join(gt[Gm[:, j]]) + "\n")
    data = read_vcf_packed(str(tmp_path / "x.vcf.gz"))
    assert np.array_equal(data.unpack_rows(0, N), Gm)
    out = tmp_path / "out"
    assert cli.main(["train", "--epochs", "3", "--k", str(K), "--name", "v", "--data_path", str(tmp_path / "x.vcf.gz"), "--save_dir", str(out),
                     "--seed", "13", "--batch_size", "100", "--hidden_size", "128"]) == 0
    Q = np.loadtxt(out / f"v.{K}.Q", dtype=np.float32)
    V = RSVD(data, N, M, 8, 13)
    Ps, Qs, _ = na.train(3, 100, 2e-3, K, 13, data, dev, 1, 128, True, V, None, None, None, 8)
    assert np.array_equal(Q, Qs[0])


@pytest.mark.gpu
def test_prefetched_epoch_orders_on_the_device_are_the_sampler_sequence():
    """model._EpochOrders on cuda: drawn into pinned memory one epoch ahead, copied on a side stream, narrowed on the device --
    the orders the kernels read are epoch_order()'s (= RandomSampler's), buffers are not overwritten while an epoch that reads
    them is still queued, and the generator ends in the same state."""
    from neural_admixture_amd.model import epoch_order, _EpochOrders
    dev = torch.device("cuda:0")
    n, epochs = 100_000, 6
    g1, g2 = torch.Generator().manual_seed(11), torch.Generator().manual_seed(11)
    orders = _EpochOrders(g2, n, dev)
    sums = []
    busy = torch.zeros(1 << 22, device=dev)
    for e in range(epochs):
        got = orders.take(e, prefetch=e + 1 < epochs)
        for _ in range(20):
            busy.add_(1.0)                                    # stands in for the epoch's steps: keeps the stream behind the host
        sums.append((got.to(torch.int64) * torch.arange(n, device=dev)).sum())     # reads the order late on the compute stream
        orders.epoch_queued()
    torch.cuda.synchronize()
    for e in range(epochs):
        want = epoch_order(g1, n).to(torch.int64)
        assert int(sums[e].item()) == int((want * torch.arange(n)).sum().item()), e
    assert torch.equal(g1.get_state(), g2.get_state())


@pytest.mark.gpu
@pytest.mark.parametrize("ks", [[5], [3], [8], [13], [2, 3, 9], [20, 4]])
def test_q_operand_images_from_the_mlp_forward_give_the_same_pass2(ks):
    """nadm_mlp_fwd_images + nadm_decode_bce_images: the MLP forward writes Q as the bf16 operand images every pass-2 block
    would otherwise build itself; gradients, dQ slabs and loss must be bit-identical to the path without images.  Batches of
    100, then 37, then 70 rows through the same engine: a shorter batch must not see the longer one's rows in its last tile.
    K = 20 has no image (fp32 kernel) and rides along with a K = 4 head that has one."""
    dev = _dev()
    N, M = 130, 2301
    Gm = O.synth_genotypes(N, M, 4, seed=41)
    rng = np.random.default_rng(12)
    p = O.make_params(3, (rng.standard_normal((M, 8)) / 48).astype(np.float32), rng.uniform(0.02, 0.98, (sum(ks), M)).astype(np.float32), 64, ks)
    from unfused_step import unfused_step
    e1, e2 = make_engine(Gm, p, 100), make_engine(Gm, p, 100)
    assert e1.qimg is not None
    for b in (100, 37, 70):
        idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(dev)
        for e in (e1, e2):
            e.forward(idx, b)
            if e is e2:
                e.Q = e.Q                                      # "Q written from outside": drops the images, pass 2 splits Q itself
            assert e._qimg_b == (b if e is e1 else -1)
            e.backward(idx, b, True)
        torch.cuda.synchronize()
        assert torch.equal(e1.Q, e2.Q)
        assert torch.equal(e1.gbig, e2.gbig), b
        assert torch.equal(e1.gsmall, e2.gsmall), b
        assert e1.read_loss() == e2.read_loss()
        e1.train_step(idx, b, 2e-3, False)
        unfused_step(e2, idx, b, 2e-3, False)                  # (nadm_mlp_fwd without images)
    torch.cuda.synchronize()
    assert torch.equal(e1.big, e2.big) and torch.equal(e1.small, e2.small)


@pytest.mark.gpu
@pytest.mark.parametrize("ks,Hd,M,b", [([8], 1024, 60_000, 800), ([5], 64, 2301, 37), ([2, 3, 4], 96, 2301, 37)])
def test_small_parameter_update_riding_in_the_next_pass1_equals_the_immediate_one(ks, Hd, M, b):
    """nadm_step leaves the sum of the weight-gradient partials + Adam on the small parameters to side blocks of the NEXT step's
    pass 1 (nadm_encode_fwd_small).  Parameters, moments, gradients and losses stay bit-identical to the immediate
    nadm_small_grads launch (tests/unfused_step.py); reading eng.small (or the moments / gradient) between two steps applies the
    owed update first; an encoder-only call (infer_q), a plain forward / backward pair and load_params in between are all served."""
    from unfused_step import unfused_step
    dev = _dev()
    N = max(b + 20, 90)
    Gm = O.synth_genotypes(N, M, 4, seed=78)
    rng = np.random.default_rng(6)
    p = O.make_params(3, (rng.standard_normal((M, 8)) / 48).astype(np.float32), rng.uniform(0.05, 0.95, (sum(ks), M)).astype(np.float32), Hd, ks)
    e1, e2 = make_engine(Gm, p, b), make_engine(Gm, p, b)
    for s in range(12):
        idx = torch.from_numpy(rng.permutation(N)[:b].astype(np.int32)).to(dev)
        e1.train_step(idx, b, 2e-3, s % 3 == 0)
        unfused_step(e2, idx, b, 2e-3, s % 3 == 0)
        if s == 4:                                            # a look in between: applies the update, the next pass 1 then has none to do
            assert torch.equal(e1.small, e2.small)
            assert torch.equal(e1.msmall, e2.msmall) and torch.equal(e1.gsmall, e2.gsmall)
        if s == 7:                                            # encoder-only pass in between (final-Q style): consumes the update
            q1, q2 = e1.infer_q(idx, b), e2.infer_q(idx, b)
            assert all(torch.equal(a, c) for a, c in zip(q1, q2))
        if s == 9:                                            # a plain forward / backward pair in between
            for e in (e1, e2):
                e.forward(idx, b)
                e.backward(idx, b, True)
            assert torch.equal(e1.gsmall, e2.gsmall) and torch.equal(e1.gbig, e2.gbig)
            e1.read_loss(); e2.read_loss()
    torch.cuda.synchronize()
    assert e1.read_loss() == e2.read_loss()
    assert torch.equal(e1.small, e2.small) and torch.equal(e1.msmall, e2.msmall) and torch.equal(e1.vsmall, e2.vsmall)
    assert torch.equal(e1.big, e2.big) and torch.equal(e1.mbig, e2.mbig)
    # load_params with an update still owed: the update must not leak into the new parameters
    e1.train_step(idx, b, 2e-3, False)
    for e in (e1, e2):
        e.load_params(p.V, np.concatenate([P.T for P in p.P], axis=0), small_vec(p))
    assert torch.equal(e1.small, e2.small) and float(e1.msmall.abs().max()) == 0.0 and e1.step_count == 0


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["torch_extension", "ctypes"])
def test_pack2bit_module_runs_the_reference_call_sequence(form):
    """The reference's own sequence around its native module, through neural_admixture_amd.pack2bit: allocate
    ``packed_data [N, (M + 3) // 4]`` and pack the whole matrix once (model/train.py:121,126), then per batch gather packed rows
    (DataLoader over the packed tensor, src/loaders.py:62-72), allocate ``unpacked_step [b, M]`` and unpack
    (model/neural_admixture.py:404-406; the final-Q pass does the same with sequential batches of <= 1024, :374-378).  Bit-exact
    against the layout fixture and the oracle's pack rule, caller-owned buffers, shape errors as RuntimeError."""
    import types
    from neural_admixture_amd import pack2bit as p2b
    dev = _dev()
    if form == "torch_extension":       # the reference's own form: a module built with torch.utils.cpp_extension (model/train.py:122-125)
        if p2b.extension is None:       # not shipped with this checkout: build it the way __graft_entry__.build() does (~20 s, host code only)
            import importlib
            import __graft_entry__ as ge
            ge.build_pack2bit_extension(os.path.join(ROOT, "neural-admixture_amd", "csrc"))
            p2b = importlib.reload(p2b)
        assert p2b.extension is not None, "csrc/ext/_pack2bit.so missing: __graft_entry__.build() builds it"
        assert p2b.pack2bit_cpu_to_gpu is p2b.extension.pack2bit_cpu_to_gpu          # ... and it is what the package hands out
        pack2bit = p2b.extension
    else:
        pack2bit = types.SimpleNamespace(pack2bit_cpu_to_gpu=p2b._pack2bit_cpu_to_gpu_ctypes, unpack2bit_gpu_to_gpu=p2b._unpack2bit_gpu_to_gpu_ctypes)
    d = np.load(f"{G}/pack_layout.npz")
    rng = np.random.default_rng(5)
    for Gm in (d["G"], d["G_hibits"], rng.integers(0, 4, size=(2311, 4099), dtype=np.uint8), rng.integers(0, 256, size=(9000, 37), dtype=np.uint8)):
        N, M = Gm.shape
        data = torch.from_numpy(np.ascontiguousarray(Gm))
        packed_data = torch.empty((N, (M + 3) // 4), dtype=torch.uint8, device=dev)             # train.py:121
        assert pack2bit.pack2bit_cpu_to_gpu(data, packed_data) is None                          # train.py:126
        assert np.array_equal(packed_data.cpu().numpy(), O.pack2bit(Gm))
        if Gm is d["G"]:
            assert np.array_equal(packed_data.cpu().numpy(), d["packed"])
        gen = torch.Generator().manual_seed(3)
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(packed_data), batch_size=min(N, 800),
                                             sampler=torch.utils.data.RandomSampler(range(N), generator=gen), drop_last=False)
        seen = 0
        gen2 = torch.Generator().manual_seed(3)
        order = torch.randperm(N, generator=gen2).numpy()
        for (x_step,) in loader:
            unpacked_step = torch.empty((x_step.shape[0], M), dtype=torch.uint8, device=dev)   # neural_admixture.py:405
            assert pack2bit.unpack2bit_gpu_to_gpu(x_step, unpacked_step) is None               # :406
            rows = order[seen: seen + x_step.shape[0]]
            assert np.array_equal(unpacked_step.cpu().numpy(), Gm[rows] & 3)
            seen += x_step.shape[0]
        assert seen == N
    with pytest.raises(RuntimeError, match="Output tensor column dimension mismatch"):
        pack2bit.pack2bit_cpu_to_gpu(torch.zeros((4, 9), dtype=torch.uint8), torch.empty((4, 2), dtype=torch.uint8, device=dev))
    with pytest.raises(RuntimeError, match="Input tensor row dimension mismatch"):
        pack2bit.unpack2bit_gpu_to_gpu(torch.empty((3, 3), dtype=torch.uint8, device=dev), torch.empty((4, 9), dtype=torch.uint8, device=dev))
