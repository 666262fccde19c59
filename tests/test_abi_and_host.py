"""CPU-only checks: the C-ABI library loads and exports every symbol include/nadm.h declares,
argument validation fails loudly without touching a GPU, the host packer is bit-exact, and the
host-side mirror (layout, initial weights, samplers, signatures, output formats) matches the reference."""
import ctypes as C
import inspect
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import nadm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    from neural_admixture_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nadm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hooks = set(re.findall(r"\b(nadm_[a-z0-9_]+)\s*\(", "".join(re.findall(r"#ifdef NADM_TEST_HOOKS.*?#endif", hdr, flags=re.S))))
    hdr = re.sub(r"#ifdef NADM_TEST_HOOKS.*?#endif", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nadm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 17 and hooks == {"nadm_test_force_slices", "nadm_test_force_generic_mlp", "nadm_test_force_p3_slices"}
    from conftest import HOOK_LIB
    product = os.path.join(ROOT, "neural-admixture_amd", "csrc", "libnadm.so")
    raw, test_build = C.CDLL(product), C.CDLL(HOOK_LIB)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in nadm.h but not exported"
        assert hasattr(test_build, name)
    for name in hooks:                                   # the laboratory is not in the product
        assert not hasattr(raw, name) and hasattr(test_build, name)
    assert declared == set(_lib.EXPORTS)               # the Python binding covers the whole header
    assert _lib.lib.nadm_abi_version() == 14


def test_argument_validation_without_gpu():
    from neural_admixture_amd._lib import lib, check
    assert [lib.nadm_pad_k(k) for k in (1, 3, 4, 7, 8, 9, 13, 16, 17, 24, 25, 33, 48, 49, 64)] == \
        [4, 4, 4, 8, 8, 12, 16, 16, 24, 24, 32, 48, 48, 64, 64]
    assert lib.nadm_pad_k(0) < 0 and lib.nadm_pad_k(65) < 0
    with pytest.raises(RuntimeError, match="null pointer"):
        check(lib.nadm_encode_fwd(None, 16, None, 1, 4, None, 8, None, None), "encode_fwd")
    with pytest.raises(RuntimeError, match="1-based"):
        buf = torch.zeros(8)
        p = C.c_void_p(buf.data_ptr())
        check(lib.nadm_adam(p, p, p, p, 8, 8, 1e-3, 0, 1.0, None), "adam")
    from neural_admixture_amd._lib import Heads
    h = Heads()
    ks = (C.c_int32 * 2)(5, 3)
    with pytest.raises(RuntimeError, match="ascending"):
        check(lib.nadm_heads_init(C.byref(h), 8, 64, ks, 2), "heads_init")
    ks = (C.c_int32 * 1)(65)
    with pytest.raises(RuntimeError):
        check(lib.nadm_heads_init(C.byref(h), 8, 64, ks, 1), "heads_init")


def test_argument_validation_of_the_round1_additions(tmp_path):
    """Error paths of the entry points added after the core path: every one fails with a message before any launch."""
    from neural_admixture_amd._lib import lib, check
    buf = torch.zeros(64)
    p = C.c_void_p(buf.data_ptr())
    with pytest.raises(RuntimeError, match="K must be in 1..16"):
        check(lib.nadm_loglik(p, 16, 4, 8, p, p, 17, 17, 1e-6, p, None), "loglik")
    with pytest.raises(RuntimeError, match="q_stride"):
        check(lib.nadm_loglik(p, 16, 4, 8, p, p, 4, 3, 1e-6, p, None), "loglik")
    with pytest.raises(RuntimeError, match="CP <= 8"):
        check(lib.nadm_pca_project(p, 16, p, 4, 8, p, 12, p, None), "pca_project")
    with pytest.raises(RuntimeError, match="CP <= 8"):
        check(lib.nadm_pca_project_t(p, 16, p, 4, 8, p, p, 12, p, None), "pca_project_t")
    with pytest.raises(RuntimeError, match="operand image of dZ"):
        check(lib.nadm_encode_bwd(p, 16, p, 4, 8, p, None, 8, p, 0, None), "encode_bwd")
    with pytest.raises(RuntimeError, match="0 < CP <= 8"):
        check(lib.nadm_dz_image(p, 4, 12, p, None), "dz_image")
    assert lib.nadm_dz_image_bytes(800) == 7 * 7 * 64 * 16 and lib.nadm_dz_image_bytes(128) == 7 * 64 * 16
    assert lib.nadm_batch_copy_bytes(800, 500000) == 977 * 800 * 128 and lib.nadm_batch_copy_bytes(37, 2301) == 5 * 37 * 128
    with pytest.raises(RuntimeError, match="number of classes"):
        check(lib.nadm_supervised_ce(p, 8, 3, 4, p, None, 4, 5, 100.0, p, p, None), "supervised_ce")
    with pytest.raises(RuntimeError, match="k <= kp <= SP"):
        check(lib.nadm_supervised_ce(p, 8, 5, 4, p, None, 4, 5, 100.0, p, p, None), "supervised_ce")
    with pytest.raises(RuntimeError, match="null pointer"):
        check(lib.nadm_bed_to_packed_dev(None, 4, 8, p, 16, p, 1, p, None), "bed_to_packed_dev")
    with pytest.raises(RuntimeError, match="multiple of 16"):
        check(lib.nadm_bed_to_packed_dev(p, 4, 8, p, 8, p, 1, p, None), "bed_to_packed_dev")
    with pytest.raises(RuntimeError, match="cannot open"):
        check(lib.nadm_savetxt_f32(str(tmp_path / "no_such_dir" / "x.txt").encode(), p, 2, 2, 2), "savetxt")
    with pytest.raises(RuntimeError, match="bad shape"):
        check(lib.nadm_savetxt_f32(str(tmp_path / "x.txt").encode(), p, 2, 4, 2), "savetxt")
    with pytest.raises(RuntimeError, match="null pointer"):
        check(lib.nadm_mlp_bwd_weights(None, 4, p, p, p, p, p, p, p, None), "mlp_bwd_weights")
    # chunk bookkeeping: slabs of pass 2 are 256 SNPs for the matrix-pipe kernels, consistent with nadm_decode_chunks
    for kp in (4, 8, 12, 16):
        assert lib.nadm_decode_chunk_snps(kp) == 256
        assert lib.nadm_decode_chunks(1000, kp) == 4 and lib.nadm_decode_chunks(1024, kp) == 4 and lib.nadm_decode_chunks(1025, kp) == 5
    for kp in (24, 48):
        c = lib.nadm_decode_chunk_snps(kp)
        assert c > 0 and lib.nadm_decode_chunks(10 * c + 1, kp) == 11
    assert lib.nadm_loglik_blocks(1) == 8 and lib.nadm_loglik_blocks(1025) == 16          # 8 row slices x 1024-SNP blocks


def test_pass2_sample_slice_rule_and_slab_sizes():
    """nadm_decode_slices is a function of (b, M, kp) alone -- every path to pass 2 takes the same cut, so the rounding of the sum
    over the slices is the same everywhere: never for the BASELINE shapes (the S = 1 kernel is their launch), never for fewer than 8
    tiles or 131k+ SNPs or the generic kernel (kp > 16), no empty slice, at most 8; nadm_decode_slices_max bounds every shorter batch
    (it sizes the slab); the force hook of the tests overrides the rule and is capped by the tile count."""
    from neural_admixture_amd._lib import lib
    for b, M in ((800, 500_000), (800, 600_000), (800, 1_000_000), (100, 500_000), (800, 131_072), (400, 50_000), (448, 100_000)):
        assert lib.nadm_decode_slices(b, M, 8) == 1, (b, M)
    assert lib.nadm_decode_slices(800, 50_000, 24) == 1 and lib.nadm_decode_slab_floats(50_000, 24, 4) == 0
    assert lib.nadm_decode_slices(800, 100_000, 8) == 3 and lib.nadm_decode_slices(800, 50_000, 8) == 3
    assert lib.nadm_decode_slices(6400, 62_500, 8) == 5
    for M in (2301, 6200, 25_000, 62_500, 100_000, 130_000):
        for bmax in (449, 800, 1603, 6400):
            mx = lib.nadm_decode_slices_max(bmax, M, 16)
            assert 1 <= mx <= 8
            for b in range(1, bmax + 1, 37):
                s_ = lib.nadm_decode_slices(b, M, 16)
                tiles = (b + 63) // 64
                assert 1 <= s_ <= mx and (s_ == 1 or (s_ - 1) * ((tiles + s_ - 1) // s_) < tiles)      # no empty slice
    chunks = lib.nadm_decode_chunks(50_000, 8)
    assert lib.nadm_decode_slab_floats(50_000, 8, 3) == 3 * chunks * (256 * 8 + 4) and lib.nadm_decode_slab_floats(50_000, 8, 1) == 0


def test_pass3_sample_slice_rule_and_slab_sizes():
    """nadm_encode_slices (r06) is a function of (b, M, CP) alone: 1 for every single-GPU BASELINE shape (the S = 1 kernel is their launch), for
    short batches and for the fp32 kernels (CP > 8); the SNP-sharded rank of configs[3] on 8 GPUs (6400 rows x 62.5k SNPs) is cut; no empty
    slice, at most 8; nadm_encode_slices_max bounds every shorter batch (it sizes the slab)."""
    from neural_admixture_amd._lib import lib
    for b, M in ((800, 500_000), (800, 600_000), (800, 1_000_000), (100, 500_000), (104, 600_000), (800, 62_500), (896, 4000), (6400, 500_000)):
        assert lib.nadm_encode_slices(b, M, 8) == 1, (b, M)
    assert lib.nadm_encode_slices(6400, 62_500, 8) == 2 and lib.nadm_encode_slices(6400, 62_500, 12) == 1       # measured: 77 -> 57 us; more slices lose
    assert lib.nadm_encode_slices(3200, 125_000, 8) == 1 and lib.nadm_encode_slices(4096, 90_000, 4) == 2          # 245 chunks x 25 tiles: any cut loses
    for M in (1100, 3000, 62_500, 125_000, 190_000):
        for bmax in (1025, 1600, 6400):
            mx = lib.nadm_encode_slices_max(bmax, M, 8)
            assert 1 <= mx <= 8
            for b in range(1, bmax + 1, 97):
                s_ = lib.nadm_encode_slices(b, M, 8)
                tiles = (b + 127) // 128
                assert 1 <= s_ <= mx and (s_ == 1 or (s_ - 1) * ((tiles + s_ - 1) // s_) < tiles)
    chunks = lib.nadm_encode_bwd_chunks(62_500)
    assert chunks == 123 and lib.nadm_encode_slab_floats(62_500, 8, 2) == 2 * chunks * 512 * 8 and lib.nadm_encode_slab_floats(62_500, 8, 1) == 0


def test_force_hook_of_the_test_build_overrides_the_slice_rule(request):
    from conftest import in_hook_build
    if not in_hook_build(request):
        return
    from neural_admixture_amd._lib import lib
    try:
        lib.nadm_test_force_slices(4)
        assert lib.nadm_decode_slices(800, 500_000, 8) == 4 and lib.nadm_decode_slices(100, 500_000, 8) == 2 and lib.nadm_decode_slices(800, 500_000, 32) == 1
        lib.nadm_test_force_slices(1)
        assert lib.nadm_decode_slices(800, 50_000, 8) == 1
    finally:
        lib.nadm_test_force_slices(0)
    assert lib.nadm_decode_slices(800, 50_000, 8) == 3


def test_engine_refuses_cpu_device():
    import neural_admixture_amd as na
    with pytest.raises(RuntimeError, match="GPU"):
        na.Engine(100, 8, 16, [3], torch.device("cpu"), 10)
    with pytest.raises(RuntimeError, match="GPU"):
        na.train(1, 8, 1e-3, 3, 0, torch.zeros((4, 100), dtype=torch.uint8), torch.device("cpu"), 0, 16, True, np.zeros((8, 100), np.float32), None)


@pytest.mark.parametrize("shape", [(5, 11), (3, 9), (0, 7), (4, 0), (300, 4099), (1, 1)])
def test_host_packer_bit_exact(shape):
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.layout import ModelLayout
    rng = np.random.default_rng(sum(shape))
    Gm = rng.integers(0, 256, size=shape, dtype=np.uint8)         # high bits must be masked
    N, M = shape
    ld = max(16, ModelLayout.row_stride(M))
    src = torch.from_numpy(np.ascontiguousarray(Gm).reshape(N, M)) if N * M else torch.zeros((N, M), dtype=torch.uint8)
    out = torch.full((max(N, 1), ld), 255, dtype=torch.uint8)
    if N and M:
        check(lib.nadm_pack2bit_host(ptr(src), ptr(out), N, M, ld))
        ref = O.pack2bit(Gm)
        assert np.array_equal(out.numpy()[:N, :ref.shape[1]], ref)
        assert not out.numpy()[:N, ref.shape[1]:].any()


def test_pack_layout_golden():
    from neural_admixture_amd._lib import lib, check, ptr
    d = np.load(f"{G}/pack_layout.npz")
    for g, pk in ((d["G"], d["packed"]), (d["G_hibits"], d["packed_hibits"])):
        out = torch.zeros((g.shape[0], 16), dtype=torch.uint8)
        check(lib.nadm_pack2bit_host(ptr(torch.from_numpy(np.ascontiguousarray(g))), ptr(out), g.shape[0], g.shape[1], 16))
        assert np.array_equal(out.numpy()[:, :pk.shape[1]], pk)


def test_layout_offsets_and_padding():
    from neural_admixture_amd.layout import ModelLayout
    L = ModelLayout(1000, 8, 64, [4, 2, 3])             # unsorted input is sorted like NeuralEncoder does (:27)
    assert L.ks == [2, 3, 4] and L.kp == [4, 4, 4] and L.qoff == [0, 4, 8] and L.SP == 12
    assert L.n_small == 8 + 64 * 8 + 64 + sum(k * 64 + k for k in (2, 3, 4))
    assert L.p_off == [8000, 12000, 16000] and L.n_big == 20000 and L.clamp_from == 8000
    assert ModelLayout.row_stride(1000) == 256 and ModelLayout.row_stride(8451) == 2176      # ceil(M/4) rounded up to 128 bytes
    offs, tot = L.dq_offsets(10)
    assert offs[0] == 0 and tot == sum(c * 10 * kp for c, kp in zip(L.dec_chunks, L.kp))


@pytest.mark.parametrize("M,ks,world", [(1000, [3], 1), (1000, [3], 2), (509, [2, 3, 4], 3), (8451, [3], 8), (500_000, [8], 8), (77, [20], 5)])
def test_flat_layout_cuts_both_messages_into_world_equal_slices(M, ks, world):
    """nadm_flat_layout (include/nadm.h): [small | pad | V | gap | all P | gap]; message B = [0, world * slice_b) holds the small
    parameters and V, message A = [msg_a_off, n_flat) every head's P; slices are 16-byte multiples; world = 1 has no gaps."""
    from neural_admixture_amd.layout import ModelLayout
    L = ModelLayout(M, 8, 64, ks, world)
    assert L.off_v % 64 == 0 and L.off_v % (4 * world) == 0 and L.n_small <= L.off_v < L.n_small + 64 * 4 * world
    assert L.slice_b % 4 == 0 and L.slice_a % 4 == 0
    assert L.n_buckets == 1 and L.bkt_off == [0, L.msg_a_off] and L.bkt_slice == [L.slice_b] and L.bkt_m0 == [0, M] and L.bkt_mom == [0]
    assert L.msg_a_off == world * L.slice_b and L.n_flat == L.msg_a_off + world * L.slice_a
    assert L.msg_a_off >= L.off_v + M * L.CP                               # V ends inside message B
    assert L.off_v + L.p_off[0] == L.msg_a_off                             # the first P starts message A
    end_p = L.off_v + L.p_off[-1] + M * L.kp[-1]
    assert end_p <= L.n_flat and L.n_flat - end_p < 4 * world              # the trailing gap is smaller than one rounding unit per rank
    assert L.msg_a_off - (L.off_v + M * L.CP) < 4 * world
    for h in range(len(ks) - 1):
        assert L.p_off[h + 1] == L.p_off[h] + M * L.kp[h]                  # heads back to back
    if world == 1:
        assert L.n_flat == L.off_v + M * L.CP + sum(M * kp for kp in L.kp) and L.clamp_from == M * L.CP
    L1 = ModelLayout(M, 8, 64, ks, 1)
    assert L.n_small == L1.n_small and (L.off_v == L1.off_v or 64 % (4 * world))   # V starts where it always does unless the slices need a wider pad


@pytest.mark.parametrize("M,C_,world,nb", [(500_000, 8, 8, 4), (500_000, 8, 1, 4), (1_000_000, 8, 8, 8), (600_000, 8, 3, 4), (20_000, 3, 2, 4),
                                             (8451, 8, 2, 4), (509, 8, 4, 2), (100_000, 8, 5, 3), (4096, 8, 2, 8)])
def test_flat_layout_cuts_message_b_into_range_major_buckets(M, C_, world, nb):
    """r05: message B = [small | V] as SNP-range buckets -- every bucket `world` contiguous slices (16-byte multiples), boundaries on
    every pass's chunk (2048 SNPs), bucket 0 carries the small parameters, the buckets tile [0, msg_a_off) without holes, V stays ONE
    contiguous [M, CP] array, and everything outside message B is where the one-bucket layout has it."""
    from neural_admixture_amd.layout import ModelLayout
    L = ModelLayout(M, C_, 64, [3], world, nb)
    L1 = ModelLayout(M, C_, 64, [3], world, 1)
    n = L.n_buckets
    assert 1 <= n <= nb and (n == nb or M < nb * 2048 * world)              # fewer only where M does not hold that many ranges
    for f in ("n_flat", "off_v", "msg_a_off", "slice_a", "slice_b", "p_off", "n_small"):
        assert getattr(L, f) == getattr(L1, f), f                          # the cut changes nothing else
    assert L.bkt_off[0] == 0 and L.bkt_off[n] == L.msg_a_off and L.bkt_m0[0] == 0 and L.bkt_m0[n] == M
    mom = 0
    for j in range(n):
        lo, hi = L.bkt_off[j], L.bkt_off[j + 1]
        assert hi > lo and (hi - lo) == world * L.bkt_slice[j] and L.bkt_slice[j] % 4 == 0
        assert L.bkt_mom[j] == mom
        mom += L.bkt_slice[j]
        assert L.bkt_m0[j] < L.bkt_m0[j + 1]
        if j > 0:
            assert L.bkt_m0[j] % 2048 == 0 and lo == L.off_v + L.bkt_m0[j] * L.CP     # a range starts on a chunk of every pass, at its V rows
    assert mom == L.slice_b
    sizes = [L.bkt_m0[j + 1] - L.bkt_m0[j] for j in range(n)]
    assert max(sizes) - min(sizes) <= 2 * 2048 * world                     # ranges of about equal length


def test_plan_and_transport_entry_points_validate_their_arguments():
    """No GPU needed: nadm_comm_emulated builds rank 0 of W; nadm_plan_create refuses incomplete descriptors, nadm_step a NULL plan."""
    import ctypes as C
    from neural_admixture_amd._lib import lib, CommStruct, PlanDesc
    from neural_admixture_amd.comm import emulated_comm, torch_comm
    c = emulated_comm(8)
    assert (c.rank, c.world, c.kind) == (0, 8, "emulated")
    st = c.handle.contents
    assert st.rank == 0 and st.world == 8 and st.reduce_scatter(None, None, 4, None) == 0 and st.all_gather(None, None, 4, None) == 0
    c.close()
    t = torch_comm(1, 2)
    assert t.handle.contents.rank == 1 and t.handle.contents.world == 2 and t.transport.buffers == []
    assert lib.nadm_comm_emulated(0, C.byref(C.POINTER(CommStruct)())) != 0
    plan = C.c_void_p()
    d = PlanDesc()
    assert lib.nadm_plan_create(C.byref(d), C.byref(plan)) != 0 and b"M, bmax" in lib.nadm_last_error()
    d.mode, d.M, d.ld, d.bmax = 7, 100, 32, 10
    assert lib.nadm_plan_create(C.byref(d), C.byref(plan)) != 0 and b"unknown mode" in lib.nadm_last_error()
    d.mode = 0
    assert lib.nadm_plan_create(C.byref(d), C.byref(plan)) != 0 and b"head table" in lib.nadm_last_error()
    assert lib.nadm_step(None, None, 1, 1e-3, 1, None) != 0 and b"null pointer" in lib.nadm_last_error()
    assert lib.nadm_plan_step_count(None) == -1


def test_a_step_that_fails_part_way_poisons_its_plan():
    """nadm_step advances the Adam step count, clears hand-offs and forks streams before launches that can fail.  Nothing
    is unwound; the plan is marked instead and every later call on it fails fast and says why.  Here (no GPU) the first launch is
    refused; argument errors are caught before anything is touched and do not poison."""
    import ctypes as C
    from neural_admixture_amd._lib import lib, PlanDesc
    from neural_admixture_amd.layout import ModelLayout
    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU: the test relies on the launch being refused")
    L = ModelLayout(4096, 8, 64, [3])
    d = PlanDesc()
    d.mode, d.bmax, d.M, d.ld, d.heads = 0, 16, L.M, ModelLayout.row_stride(L.M), L.heads
    buf = np.zeros(1 << 20, dtype=np.float32)                 # stands in for every buffer: nothing is ever launched on it
    for n in ("params", "grads", "m", "v", "zpart", "Z", "rinv", "Zn", "H", "Q", "dL", "dHpre", "dgp", "dZ", "dqpart", "losspart", "small_part",
              "qimg", "dzimg", "dzcnt", "xg", "loss_acc", "xp"):
        setattr(d, n, buf.ctypes.data)
    plan = C.c_void_p()
    assert lib.nadm_plan_create(C.byref(d), C.byref(plan)) == 0
    idx = np.arange(16, dtype=np.int32)
    assert lib.nadm_step(plan, idx.ctypes.data, 99, 1e-3, 1, None) != 0 and b"batch size" in lib.nadm_last_error()
    assert lib.nadm_plan_poisoned(plan) == 0                  # refused before anything was touched
    assert lib.nadm_step(plan, idx.ctypes.data, 16, 1e-3, 1, None) != 0 and b"launch failed" in lib.nadm_last_error()
    assert lib.nadm_plan_poisoned(plan) == 1
    for call in (lambda: lib.nadm_step(plan, idx.ctypes.data, 16, 1e-3, 1, None), lambda: lib.nadm_plan_flush(plan, None),
                 lambda: lib.nadm_plan_infer(plan, idx.ctypes.data, 16, None)):
        assert call() == 6 and b"failed part-way" in lib.nadm_last_error()
    lib.nadm_plan_destroy(plan)
    d.reserved = 1
    assert lib.nadm_plan_create(C.byref(d), C.byref(plan)) != 0 and b"reserved" in lib.nadm_last_error()


def test_initial_weights_match_reference_rng_stream():
    from neural_admixture_amd.model import init_encoder_weights
    d = np.load(f"{G}/one_step_multihead.npz")
    v = init_encoder_weights(int(d["seed"]), 8, int(d["Hd"]), [2, 3, 4])
    Hd = int(d["Hd"])
    assert np.all(v[:8] == 1)
    assert np.array_equal(v[8:8 + Hd * 8].reshape(Hd, 8), d["init_common_encoder_0_weight"])
    assert np.array_equal(v[8 + Hd * 8: 8 + Hd * 9], d["init_common_encoder_0_bias"])
    o = 8 + Hd * 9
    for h, k in enumerate((2, 3, 4)):
        assert np.array_equal(v[o:o + k * Hd].reshape(k, Hd), d[f"init_multihead_encoder_heads_{h}_weight"])
        o += k * Hd
        assert np.array_equal(v[o:o + k], d[f"init_multihead_encoder_heads_{h}_bias"])
        o += k


def test_train_signature_matches_reference():
    import neural_admixture_amd as na
    params = inspect.signature(na.train).parameters
    names = list(params)
    assert names[:15] == ["epochs", "batch_size", "learning_rate", "K", "seed", "data", "device", "num_gpus", "hidden_size",
                          "master", "V", "pops", "min_k", "max_k", "n_components"]     # train.py:19-21
    for extra in names[15:]:                                # additions must be keyword-only with a default
        assert params[extra].kind is inspect.Parameter.KEYWORD_ONLY and params[extra].default is not inspect.Parameter.empty
    names = list(inspect.signature(na.NeuralAdmixture.__init__).parameters)[1:13]
    assert names == ["k", "epochs", "batch_size", "learning_rate", "device", "seed", "num_gpus", "master", "pack2bit",
                     "min_k", "max_k", "supervised_loss_weight"]                        # neural_admixture.py:248-249
    names = list(inspect.signature(na.NeuralAdmixture.launch_training).parameters)[1:]
    assert names == ["P", "data", "hidden_size", "num_features", "V", "M", "N", "pops"]  # :324-325


def test_gmm_init_matches_reference_demo():
    """P_init = clip(GMM means @ V) in the PCA subspace, same sklearn call as train.py:49-63."""
    from neural_admixture_amd.train import gmm_p_init
    d = np.load(f"{G}/demo_k3.npz")
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    P = gmm_p_init(Gm, d["Vt"], 3, None, None, 8, int(d["seed"]))
    assert np.abs(P - d["P_init"]).max() < 1e-6


def test_output_writers(tmp_path):
    from neural_admixture_amd.io import write_outputs
    from neural_admixture_amd.model import Q_P
    Q = np.random.default_rng(0).random((5, 3)).astype(np.float32)
    P = np.random.default_rng(1).random((7, 3)).astype(np.float32)
    write_outputs([Q], "run", 3, None, None, tmp_path, [P])
    assert np.array_equal(np.loadtxt(tmp_path / "run.3.Q", dtype=np.float32), Q)
    assert np.array_equal(np.loadtxt(tmp_path / "run.3.P", dtype=np.float32), P)
    first = open(tmp_path / "run.3.Q").readline().split(" ")
    assert len(first) == 3 and re.fullmatch(r"\d\.\d{18}e[+-]\d{2}", first[0])          # savetxt default '%.18e'
    Q_P(1024, 8, ks_list=[3]).save_config("run", str(tmp_path))
    assert json.load(open(tmp_path / "run_config.json")) == {"ks": [3], "num_features": 8, "hidden_size": 1024, "activation": "relu"}


def _bed_bytes(G):
    """Inverse of the reference's read_bed table [2,3,1,0] (utils.pyx:52): genotype code -> PLINK 2-bit code."""
    inv = np.array([3, 2, 0, 1], dtype=np.uint8)           # code 0 -> 0b11, 1 -> 0b10, 2 -> 0b00, 3 (missing) -> 0b01
    N, M = G.shape
    nb = (N + 3) // 4
    bed = np.zeros((M, nb), dtype=np.uint8)
    for i in range(N):
        bed[:, i // 4] |= (inv[G[i]] << (2 * (i % 4))).astype(np.uint8)
    return bed.reshape(-1)


@pytest.mark.parametrize("N,M,flip", [(105, 8451, False), (7, 5, False), (13, 1030, True), (4, 256, True), (1, 3, False)])
def test_bed_to_packed_matches_reader_semantics(N, M, flip):
    """BED -> packed without the uint8 detour == read_bed (table [2,3,1,0]) + minor-allele flip + pack2bit."""
    import ctypes as C
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.layout import ModelLayout
    rng = np.random.default_rng(N * 1000 + M)
    p = [0.15, 0.2, 0.6, 0.05] if flip else [0.7, 0.18, 0.1, 0.02]
    Gm = rng.choice(4, size=(N, M), p=p).astype(np.uint8)
    bed = _bed_bytes(Gm)
    ld = ModelLayout.row_stride(M)
    out = torch.full((N, ld), 255, dtype=torch.uint8)
    counts = (C.c_int64 * 4)()
    flipped = C.c_int32(0)
    check(lib.nadm_bed_to_packed(C.c_void_p(bed.ctypes.data), N, M, ptr(out), ld, counts, 1, C.byref(flipped)))
    assert [counts[i] for i in range(4)] == [int((Gm == c).sum()) for c in range(4)]
    do_flip = Gm.mean() >= 1
    assert bool(flipped.value) == do_flip
    want = Gm.copy()
    if do_flip:                                            # 2 - G with missing left at 3 (pack2bit masks the wrapped 255 back to 3)
        want = np.where(Gm == 3, 3, 2 - Gm.astype(np.int16)).astype(np.uint8)
    ref = O.pack2bit(want)
    assert np.array_equal(out.numpy()[:, :ref.shape[1]], ref)
    assert not out.numpy()[:, ref.shape[1]:].any()


def test_read_bed_packed_demo_fixture(tmp_path):
    from neural_admixture_amd.io import read_bed_packed
    d = np.load(f"{G}/demo_k3.npz")
    d["bed_bytes"].tofile(tmp_path / "demo.bed")
    (tmp_path / "demo.fam").write_text("\n".join(["s"] * int(d["N"])) + "\n")
    pg = read_bed_packed(str(tmp_path / "demo.bed"))
    assert (pg.N, pg.M) == (int(d["N"]), int(d["M"]))
    assert np.array_equal(pg.packed.numpy()[:, :d["G_packed"].shape[1]], d["G_packed"])
    assert np.array_equal(pg.unpack_rows(3, 9), O.unpack2bit(d["G_packed"], int(d["M"]))[3:9])


def test_chunked_draws_continue_the_generators_stream():
    """svd._omega_to_device draws Omega chunk by chunk into a small ring (r05): the concatenation must be the one array the reference
    draws in a single call (src/svd.py:47-48) -- PCG64's buffered 32-bit half carries over between calls."""
    for M, kp, chunk in ((8451, 20, 1000), (70_001, 20, 32768), (5, 20, 32768), (40_000, 23, 4097)):
        whole = np.random.default_rng(42).standard_normal(size=(M, kp), dtype=np.float32)
        rng = np.random.default_rng(42)
        out = np.empty((M, kp), dtype=np.float32)
        buf = np.empty((min(chunk, M), kp), dtype=np.float32)
        for s in range(0, M, chunk):
            e = min(M, s + chunk)
            rng.standard_normal(dtype=np.float32, out=buf[: e - s])
            out[s:e] = buf[: e - s]
        assert np.array_equal(out, whole)


def test_rsvd_matches_reference_on_demo():
    """8(f)-2: same Omega stream, QR/SVD steps and sign flip as src/svd.py:39-83 (host path, no GPU)."""
    from neural_admixture_amd.svd import RSVD
    from neural_admixture_amd.io import PackedGenotypes
    d = np.load(f"{G}/demo_k3.npz")
    Gm = O.unpack2bit(d["G_packed"], int(d["M"]))
    V = RSVD(Gm, Gm.shape[0], Gm.shape[1], 8, int(d["seed"]), device=None)
    assert V.shape == d["Vt"].shape and np.abs(V - d["Vt"]).max() < 1e-5
    ld = d["G_packed"].shape[1] + (-d["G_packed"].shape[1]) % 16
    pk = np.zeros((Gm.shape[0], ld), dtype=np.uint8)
    pk[:, :d["G_packed"].shape[1]] = d["G_packed"]
    V2 = RSVD(PackedGenotypes(torch.from_numpy(pk), Gm.shape[0], Gm.shape[1]), Gm.shape[0], Gm.shape[1], 8, int(d["seed"]), device=None)
    assert np.abs(V2 - d["Vt"]).max() < 1e-5


def test_cli_parsers_have_the_reference_flags():
    from neural_admixture_amd.cli import parse_train_args, parse_infer_args
    a = parse_train_args(["--save_dir", "o", "--data_path", "x.bed", "--name", "n", "--k", "3"])
    assert (a.epochs, a.batch_size, a.learning_rate, a.seed, a.hidden_size, a.n_components) == (250, 800, 2e-3, 42, 1024, 8)   # entry.py:27-43
    b = parse_infer_args(["--out_name", "o", "--save_dir", "s", "--data_path", "x.bed", "--name", "n"])
    assert b.batch_size == 1024 and b.seed == 42                                                                           # entry.py:57-65
    assert a.parallelism == "dp" and a.pops_path == "" and a.supervised_loss_weight == 100
    c = parse_train_args(["--save_dir", "o", "--data_path", "x.bed", "--name", "n", "--k", "3", "--parallelism", "snp", "--pops_path", "p.txt"])
    assert c.parallelism == "snp" and c.pops_path == "p.txt"


def test_supervised_host_logic_against_oracle_and_reference():
    """Supervised mode on the CPU test double: train.supervised_init (label mapping + class-mean P init) and the
    trainer's label plumbing (labels follow the sampled rows) reproduce the oracle's supervised run, which
    tests/test_oracle_golden.py pins against the reference's own train()."""
    from neural_admixture_amd.model import NeuralAdmixture
    from neural_admixture_amd.train import supervised_init
    from tests.fake_engine import OracleEngine
    from oracle import nadm_oracle as O
    d = np.load(os.path.join(G, "supervised_k4.npz"))
    N, M, K, Hd = int(d["N"]), int(d["M"]), int(d["K"]), int(d["Hd"])
    Gm = O.unpack2bit(d["G_packed"], M)
    pops = [str(a) for a in d["pops"]]
    y, P0 = supervised_init(Gm, pops, K)
    assert np.array_equal(y, O.labels_from_pops(pops))
    assert np.abs(P0 - O.supervised_p_init(Gm, y, K)).max() < 1e-6
    with pytest.raises(AssertionError):
        supervised_init(Gm, pops, K + 1)
    V = np.ascontiguousarray(d["Vt"].T)
    # The fixture's own run is chaotic after its first step (class-mean P init > 1 saturates R), and the two numpy / torch
    # paths round differently per host CPU.  What is checked here is the host logic -- labels follow the sampled rows, the
    # supervised term enters every step -- so the steps are made tiny: every batch then sees (almost) the initial model and
    # the epoch loss, which depends on which label goes with which row, must agree to 1e-5.
    lr = 1e-7
    tr = NeuralAdmixture(K, 1, int(d["b"]), lr, torch.device("cpu"), int(d["seed"]), 0, True, None, None, None,
                         loss_mode="always")
    tr.engine_cls = OracleEngine
    # one BLAS thread: the supervised trajectory is chaotic after its first step (class-mean P init > 1 saturates R), so the
    # two numpy paths must also agree in their summation order, which a wide BLAS pool (128 threads on the GPU boxes) breaks
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        Qs, Ps, _ = tr.launch_training(torch.from_numpy(P0), torch.from_numpy(Gm), Hd, 8, torch.from_numpy(V), M, N, torch.from_numpy(y))
        p = O.make_params(int(d["seed"]), V, P0, Hd, [K])
        p, Qo, losses = O.train_run(Gm, p, 1, int(d["b"]), lr, int(d["seed"]), labels=y)
    assert abs(tr.epoch_losses[0] - losses[0]) / losses[0] < 1e-5
    assert np.abs(Qs[0] - Qo[0]).max() < 1e-4 and np.abs(Ps[0] - p.P[0]).max() < 1e-4
    # ... and it is sensitive to the plumbing: the same run with the labels rolled by one row gives another loss
    tr2 = NeuralAdmixture(K, 1, int(d["b"]), lr, torch.device("cpu"), int(d["seed"]), 0, True, None, None, None, loss_mode="always")
    tr2.engine_cls = OracleEngine
    with threadpool_limits(limits=1):
        tr2.launch_training(torch.from_numpy(P0), torch.from_numpy(Gm), Hd, 8, torch.from_numpy(V), M, N, torch.from_numpy(np.roll(y, 1)))
    assert abs(tr2.epoch_losses[0] - losses[0]) / losses[0] > 1e-4
    assert abs(d["hi_losses"][0] - 126760.66) < 1.0         # the fixture this is anchored on


def test_native_savetxt_matches_numpy_bytes(tmp_path):
    """io.savetxt / nadm_savetxt_f32 writes exactly what np.savetxt(delimiter=' ') writes (the .Q/.P format,
    src/utils.py:56-66), including zeros, negative zero, denormals and the largest floats."""
    from neural_admixture_amd.io import savetxt, write_outputs
    rng = np.random.default_rng(0)
    for shape in ((1000, 7), (3, 1), (0, 4), (5000, 3)):
        A = (rng.standard_normal(shape) * 10.0 ** rng.integers(-38, 38, size=shape)).astype(np.float32)
        if A.size > 6:
            A.flat[:7] = [0.0, -0.0, 1.0, 5e-6, np.finfo(np.float32).max, np.finfo(np.float32).tiny, 1e-45]
        np.savetxt(tmp_path / "ref.txt", A, delimiter=' ')
        savetxt(tmp_path / "nat.txt", A)
        assert (tmp_path / "ref.txt").read_bytes() == (tmp_path / "nat.txt").read_bytes()
    Q = rng.dirichlet(np.ones(3), size=50).astype(np.float32)
    P = rng.uniform(size=(200, 3)).astype(np.float32)
    write_outputs([Q], "run", 3, None, None, tmp_path, [P])
    np.savetxt(tmp_path / "q.txt", Q, delimiter=' ')
    assert (tmp_path / "run.3.Q").read_bytes() == (tmp_path / "q.txt").read_bytes()
    assert np.array_equal(np.loadtxt(tmp_path / "run.3.P", dtype=np.float32), P)
    savetxt(tmp_path / "f64.txt", Q.astype(np.float64))                 # other dtypes: numpy path
    assert np.allclose(np.loadtxt(tmp_path / "f64.txt"), Q)


def test_library_fit_never_raises_a_thread_pool(tmp_path):
    """r05 crash: the CLI exports OPENBLAS_NUM_THREADS / OMP_NUM_THREADS = --threads (default 1) like the reference's (entry.py:138-146);
    scipy's OpenBLAS, first loaded by the library fit AFTER that, comes up with one thread, and "limit every pool to 4" RAISED it -- the
    fit then segfaults inside trtrs.  _gmm_fit.fit_means only ever lowers.  In a child process: the libraries must load in that order."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "np.linalg.qr(np.random.rand(300, 20))\n"                         # numpy's OpenBLAS is up with the host's thread count
        "os.environ['OPENBLAS_NUM_THREADS'] = '1'; os.environ['OMP_NUM_THREADS'] = '1'\n"
        f"sys.path.insert(0, {repr(os.path.join(ROOT, 'neural-admixture_amd'))})\n"
        "import _gmm_fit\n"                                               # (stand-alone module: numpy + sklearn only)
        "X = np.random.default_rng(0).standard_normal((3000, 8))\n"
        "m = _gmm_fit.fit_means(X, 6, 1)\n"
        "print('means', m.shape)\n")
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "means (6, 8)" in r.stdout, (r.returncode, r.stderr[-500:])


def test_parallel_gmm_fits_equal_sequential_ones():
    """Multi-head init: one sklearn GMM per K (train.py:65-67) fitted in concurrent child processes (_gmm_fit.py) gives
    exactly the means of the in-process sequential fits."""
    from neural_admixture_amd.train import _gmm_means_parallel
    from neural_admixture_amd._gmm_fit import fit_means
    rng = np.random.default_rng(0)
    cent = rng.standard_normal((4, 8)) * 3
    X = np.concatenate([c + rng.standard_normal((60, 8)) for c in cent]).astype("float64")
    ks = [2, 3, 4]
    par = _gmm_means_parallel(X, ks, 7)
    assert par is not None
    for k, m in zip(ks, par):
        assert m.shape == (k, 8) and np.array_equal(m, fit_means(X, k, 7))


def test_host_converters_property_based():
    """hypothesis: for ANY small shape and code distribution the host packer and the BED converter agree with the oracle's
    restatement (ragged N and M, tail bits, pad bytes, the flip rule with missing calls kept at 3)."""
    import ctypes as C
    from hypothesis import given, settings, strategies as st
    from neural_admixture_amd._lib import lib, check, ptr
    from neural_admixture_amd.layout import ModelLayout

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 23), st.integers(1, 300), st.integers(0, 2 ** 31 - 1), st.floats(0.05, 0.9))
    def prop(N, M, seed, p0):
        rng = np.random.default_rng(seed)
        rest = (1 - p0) / 3
        Gm = rng.choice(4, size=(N, M), p=[p0, rest, rest, rest]).astype(np.uint8)
        ld = ModelLayout.row_stride(M)
        out = torch.full((N, ld), 255, dtype=torch.uint8)
        check(lib.nadm_pack2bit_host(ptr(torch.from_numpy(Gm)), ptr(out), N, M, ld))
        ref = O.pack2bit(Gm)
        assert np.array_equal(out.numpy()[:, :ref.shape[1]], ref) and not out.numpy()[:, ref.shape[1]:].any()
        assert np.array_equal(O.unpack2bit(out.numpy(), M), Gm)
        bed = _bed_bytes(Gm)
        out2 = torch.full((N, ld), 255, dtype=torch.uint8)
        counts, flipped = (C.c_int64 * 4)(), C.c_int32(0)
        check(lib.nadm_bed_to_packed(C.c_void_p(bed.ctypes.data), N, M, ptr(out2), ld, counts, 1, C.byref(flipped)))
        want = np.where(Gm == 3, 3, 2 - Gm.astype(np.int16)).astype(np.uint8) if Gm.mean() >= 1 else Gm
        assert bool(flipped.value) == bool(Gm.mean() >= 1)
        assert np.array_equal(O.unpack2bit(out2.numpy(), M), want) and not out2.numpy()[:, (M + 3) // 4:].any()
        assert [counts[i] for i in range(4)] == [int((Gm == c).sum()) for c in range(4)]
    prop()


def test_device_em_restatement_gives_the_library_mixture_means():
    """_gmm_em.fit_means (float64 tensor ops; used when the device is a GPU) against sklearn's GaussianMixture with the
    reference's arguments (train.py:61): same k-means++ picks, same EM, same restart selection -> means equal to 1e-10."""
    from neural_admixture_amd._gmm_em import fit_means as em
    from neural_admixture_amd._gmm_fit import fit_means as sk
    nt = torch.get_num_threads()
    torch.set_num_threads(1)                               # tiny ops: a wide CPU thread pool only adds overhead
    try:
        rng = np.random.default_rng(0)
        for N, k, seed, sep in ((600, 3, 42, 3.0), (1500, 7, 42, 1.0), (200, 2, 3, 0.1)):
            cent = rng.standard_normal((k, 8)) * sep
            X = (rng.dirichlet(np.full(k, 0.5), N) @ cent + 0.3 * rng.standard_normal((N, 8))).astype(np.float32).astype(np.float64)
            assert np.abs(em(X, k, seed) - sk(X, k, seed)).max() < 1e-10
        with pytest.raises(ValueError):
            em(np.zeros((2, 8)), 3, 0)
    finally:
        torch.set_num_threads(nt)


def _mixture_cases():
    rng = np.random.default_rng(0)
    for N, k, seed, sep in ((600, 3, 42, 3.0), (1500, 7, 42, 1.0), (200, 2, 3, 0.1), (2504, 7, 42, 0.7), (2504, 10, 5, 0.5), (105, 3, 42, 2.0)):
        cent = rng.standard_normal((k, 8)) * sep
        X = (rng.dirichlet(np.full(k, 0.5), N) @ cent + 0.3 * rng.standard_normal((N, 8))).astype(np.float32).astype(np.float64)
        yield X, k, seed


def test_seeding_picks_are_the_librarys():
    """gmm.kmeanspp_picks: the k-means++ seed rows of five consecutive restarts drawn from ONE numpy RandomState -- the stream, the
    greedy local trials and the searches in the cumulative distances -- against sklearn.cluster.kmeans_plusplus itself."""
    from sklearn.cluster import kmeans_plusplus
    from sklearn.utils import check_random_state
    from neural_admixture_amd.gmm import kmeanspp_picks
    for X, k, seed in _mixture_cases():
        rs_lib, rs_own = check_random_state(seed), np.random.RandomState(seed)
        for _ in range(5):
            assert np.array_equal(kmeans_plusplus(X, k, random_state=rs_lib)[1], kmeanspp_picks(X, k, rs_own))
        assert rs_lib.randint(1 << 30) == rs_own.randint(1 << 30)           # ... and the streams are in the same place afterwards


def test_host_em_restatement_gives_the_library_mixture_means():
    """gmm.fit_means (numpy seeding + csrc/nadm_gmm.cpp, the default decoder init up to 20000 samples since r05) against sklearn's
    GaussianMixture with the reference's arguments (train.py:61): means equal to 1e-10, the library's own error for too few samples
    and for a collapsed component.  Restarts that reach the SAME optimum (objectives equal to ~1e-15) are ordered by rounding noise in
    the library; there the means must agree as a set, at the level the stopping rule (tol = 1e-4 on the objective) leaves."""
    from neural_admixture_amd.gmm import fit_means as native
    from neural_admixture_amd._gmm_fit import fit_means as sk
    for X, k, seed in _mixture_cases():
        assert np.abs(native(X, k, seed) - sk(X, k, seed)).max() < 1e-10
    with pytest.raises(ValueError, match="n_samples >= n_components"):
        native(np.zeros((2, 8)), 3, 0)
    with pytest.raises(ValueError, match="ill-defined empirical covariance"):
        native(np.ones((50, 8)), 2, 0, reg_covar=0.0)
    # well-separated clusters: every restart ends in the same optimum, the winner is a matter of rounding
    rng = np.random.default_rng(1)
    cent = rng.standard_normal((5, 8)) * 6
    X = cent[rng.integers(0, 5, 1200)] + 0.5 * rng.standard_normal((1200, 8))
    a, b = native(X, 5, 9), sk(X, 5, 9)
    order = [int(np.argmin(np.abs(b - a[i]).sum(1))) for i in range(5)]
    assert sorted(order) == list(range(5)) and np.abs(a - b[order]).max() < 1e-3


def test_epoch_order_is_the_random_sampler_sequence():
    """model.epoch_order replaces iterating torch's RandomSampler (loaders.py:29-31): same indices, same generator state
    after every epoch (the sampler's discarded second draw included)."""
    from torch.utils.data import RandomSampler
    from neural_admixture_amd.model import epoch_order
    for n in (1, 7, 600, 2504):
        g1, g2 = torch.Generator().manual_seed(42), torch.Generator().manual_seed(42)
        sampler = RandomSampler(range(n), generator=g1)
        for _ in range(4):
            want = np.asarray(list(iter(sampler)), dtype=np.int32)
            got = epoch_order(g2, n)
            assert got.dtype == torch.int32 and np.array_equal(got.numpy(), want)
            assert torch.equal(g1.get_state(), g2.get_state())


def test_prefetched_epoch_orders_are_the_same_sequence():
    """model._EpochOrders (the trainer's one-epoch-ahead order buffer) hands out epoch_order()'s sequence and leaves the
    generator where epoch_order() leaves it (host path; the device path is checked in test_gpu_parity)."""
    from neural_admixture_amd.model import epoch_order, _EpochOrders
    n, epochs = 700, 5
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    orders = _EpochOrders(g2, n, torch.device("cpu"))
    for e in range(epochs):
        want = epoch_order(g1, n)
        got = orders.take(e, prefetch=e + 1 < epochs)
        assert got.dtype == torch.int32 and torch.equal(got, want)
        orders.epoch_queued()
    assert torch.equal(g1.get_state(), g2.get_state())


def test_capped_host_threads_only_lowers_and_restores():
    """train.capped_host_threads: pools above the limit come down to it, pools below it are left alone (raising an OpenBLAS
    pool that was started with one thread crashes it), and everything is back afterwards -- also when the block raises."""
    from neural_admixture_amd.train import capped_host_threads
    from threadpoolctl import threadpool_info, threadpool_limits
    before_torch, before = torch.get_num_threads(), [(i["user_api"], i["num_threads"]) for i in threadpool_info()]
    with capped_host_threads(2):
        assert torch.get_num_threads() == min(before_torch, 2)
        assert all(i["num_threads"] <= 2 for i in threadpool_info())
    assert torch.get_num_threads() == before_torch and [(i["user_api"], i["num_threads"]) for i in threadpool_info()] == before
    with threadpool_limits(limits=1):
        torch.set_num_threads(1)
        try:
            with capped_host_threads(4):
                assert torch.get_num_threads() == 1 and all(i["num_threads"] == 1 for i in threadpool_info())
        finally:
            torch.set_num_threads(before_torch)
    with pytest.raises(ZeroDivisionError):
        with capped_host_threads(2):
            1 / 0
    assert torch.get_num_threads() == before_torch and [(i["user_api"], i["num_threads"]) for i in threadpool_info()] == before


def test_vcf_reader_follows_the_reference_reader_conventions(tmp_path):
    """io.read_vcf (nadm_vcf_parse_gt) against a plain-Python statement of what the reference's reader computes
    (src/snp_reader.py:73-87,108-110: scikit-allel GT as int8 with -1 fills, summed over two alleles, negatives -> 3, then
    the minor-allele orientation).  scikit-allel is not installed here, so the expected values are spelled out: phased and
    unphased calls, extra FORMAT keys, missing, half-missing (sums to 0), haploid (sums to 0), .gz input."""
    import gzip
    from neural_admixture_amd.io import read_vcf, read_vcf_packed
    hdr = "##fileformat=VCFv4.2\n##source=test\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\ts3\ts4\ts5\n"
    rows = [
        ("GT",       ["0/0", "0/1", "1/1", "./.", "1|0"]),
        ("GT:DP:GQ", ["0|0:12:99", "1|1:3:20", "0/1:7:50", ".:0:0", "./1:1:1"]),
        ("GT",       ["1", "0", ".", "0/0", "0/0"]),
        ("GT:AD",    ["0/0:1,0", "0/0:2,0", "0/1:1,1", "0/0:.", ".|.:."]),
    ]
    body = "".join(f"1\t{100 + i}\trs{i}\tA\tG\t.\tPASS\t.\t{fmt}\t" + "\t".join(calls) + "\n" for i, (fmt, calls) in enumerate(rows))

    def allel_sum(call):                                   # two alleles, -1 for a missing / absent one, summed; < 0 -> 3
        gt = call.split(":")[0].replace("|", "/").split("/")
        a = [(-1 if x == "." else int(x)) for x in gt][:2]
        a += [-1] * (2 - len(a))
        return 3 if sum(a) < 0 else sum(a)
    def oriented(raw):                                     # snp_reader.py:110 (uint8 2 - G, codes masked with 3 downstream)
        return raw if raw.mean() < 1 else np.where(raw == 3, 3, 2 - raw).astype(np.uint8)
    raw1 = np.array([[allel_sum(c) for c in calls] for _, calls in rows], dtype=np.uint8).T       # [samples, variants]
    want = oriented(raw1)
    p = tmp_path / "t.vcf"
    p.write_text(hdr + body)
    got = read_vcf(str(p))
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    with gzip.open(tmp_path / "t.vcf.gz", "wt") as f:
        f.write(hdr + body)
    assert np.array_equal(read_vcf(str(tmp_path / "t.vcf.gz")), want)
    pk = read_vcf_packed(str(p))
    assert (pk.N, pk.M) == want.shape and np.array_equal(pk.unpack_rows(0, pk.N), want)
    # orientation: mostly-alternate genotypes are flipped, missing stays missing (snp_reader.py:110 + the &3 of the packer)
    rows2 = [("GT", ["0/0", "0/0", "0/1", "./.", "0/0"]), ("GT", ["0/0", "1/1", "0/0", "0/0", "0/0"])]
    body2 = "".join(f"1\t{i}\t.\tA\tG\t.\t.\t.\t{fmt}\t" + "\t".join(c) + "\n" for i, (fmt, c) in enumerate(rows2))
    (tmp_path / "u.vcf").write_text(hdr + body2)
    raw = np.array([[allel_sum(c) for c in calls] for _, calls in rows2], dtype=np.uint8).T
    assert raw.mean() < 1 <= raw1.mean()                   # one file of each kind: kept as is / flipped
    assert np.array_equal(read_vcf(str(tmp_path / "u.vcf")), raw)
    (tmp_path / "bad.vcf").write_text(hdr + "1\t1\t.\tA\tG,T\t.\t.\t.\tGT\t2/2\t0/0\t0/0\t0/0\t0/0\n")
    with pytest.raises(AssertionError):
        read_vcf(str(tmp_path / "bad.vcf"))


def test_python_dash_m_reaches_the_cli():
    """`python -m neural_admixture_amd train ...` from the repo root (the package directory has a hyphen, the root shim
    forwards to cli.main).  Without a GPU the CLI stops with its own message -- which proves it was reached."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "neural_admixture_amd", "train", "--k", "3", "--name", "x", "--data_path", "none.bed",
                        "--save_dir", "."], cwd=root, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode != 0 and "none" in (r.stdout + r.stderr)          # reaches the reader, no such file
    else:
        assert r.returncode != 0 and "needs a ROCm GPU" in (r.stdout + r.stderr)


def test_hudsons_fst_of_the_product_matches_the_reference_table():
    """model.hudsons_fst (neural_admixture.py:532-553) on the reference's own final P of the demo run against the Fst values
    the reference computed from it (tests/golden/demo_k3.npz: hi_e5_fst), and the printed table of display_divergences."""
    from neural_admixture_amd.model import hudsons_fst
    d = np.load(f"{G}/demo_k3.npz")
    P = torch.from_numpy(d["hi_e5_P"])
    fst = d["hi_e5_fst"]
    for a in range(3):
        for b in range(a):                      # the fixture holds the lower triangle, like the printed table
            assert abs(hudsons_fst(P[:, b], P[:, a]) - float(fst[a, b])) < 1e-6
            assert abs(hudsons_fst(P[:, a], P[:, b]) - float(fst[a, b])) < 1e-6     # symmetric in its arguments
    assert hudsons_fst(P[:, 0], P[:, 0]) == 0.0
    from neural_admixture_amd.model import fst_table        # all pairs from one product (what display_divergences prints)
    T = fst_table(P)
    for a in range(P.shape[1]):
        for b in range(P.shape[1]):
            assert abs(float(T[a, b]) - hudsons_fst(P[:, a], P[:, b])) < 1e-6


def test_bed_reader_applies_the_reference_biallelic_check(tmp_path):
    """src/snp_reader.py:109: ``int(G.min()) == 0 and int(G.max()) in (2, 3)`` -- a matrix without a single 0, or with
    nothing above 1, is refused with the reference's message (computed from the code counts, no uint8 matrix)."""
    from neural_admixture_amd.io import read_bed_packed
    inv = np.array([3, 2, 0, 1], dtype=np.uint8)                           # genotype code -> PLINK 2-bit code

    def write(Gm, name):
        N, M = Gm.shape
        codes = inv[Gm.T]
        pad = np.zeros((M, (N + 3) // 4 * 4), dtype=np.uint8)
        pad[:, :N] = codes
        bed = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
        (tmp_path / f"{name}.bed").write_bytes(bytes([0x6C, 0x1B, 0x01]) + bed.tobytes())
        (tmp_path / f"{name}.fam").write_text("\n".join(["s"] * N) + "\n")
        return str(tmp_path / f"{name}.bed")
    rng = np.random.default_rng(0)
    ok = rng.choice(np.array([0, 0, 0, 1, 2], dtype=np.uint8), size=(9, 21))
    assert read_bed_packed(write(ok, "ok")).shape == (9, 21)
    for bad, name in ((np.ones((9, 21), dtype=np.uint8), "only1"),                        # max == 1
                      (rng.choice(np.array([1, 2], dtype=np.uint8), size=(9, 21)), "no0"),  # min == 1
                      (rng.choice(np.array([0, 1], dtype=np.uint8), size=(9, 21)), "no2")): # max == 1
        with pytest.raises(AssertionError, match="biallelic"):
            read_bed_packed(write(bad, name))


def test_pack2bit_module_has_the_reference_names_and_refuses_misplaced_tensors():
    """neural_admixture_amd.pack2bit = the reference's JIT-built module by its own names (pack2bit.cu:144-147); the device / shape
    checks raise RuntimeError with the reference's TORCH_CHECK messages (pack2bit.cu:66-76,121-130) before anything touches a GPU."""
    import inspect
    from neural_admixture_amd import pack2bit
    assert list(inspect.signature(pack2bit._pack2bit_cpu_to_gpu_ctypes).parameters) == ["input_cpu", "output_gpu"]
    assert list(inspect.signature(pack2bit._unpack2bit_gpu_to_gpu_ctypes).parameters) == ["input_gpu", "output_gpu"]
    if pack2bit.extension is not None:                      # the torch C++ extension (csrc/pack2bit_ext.cpp): two Tensors in, None out
        for f in (pack2bit.pack2bit_cpu_to_gpu, pack2bit.unpack2bit_gpu_to_gpu):
            assert "(arg0: torch.Tensor, arg1: torch.Tensor) -> None" in f.__doc__
    g = torch.zeros((3, 10), dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="Output tensor must be on CUDA device"):
        pack2bit.pack2bit_cpu_to_gpu(g, torch.zeros((3, 3), dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="Input tensor must be on CUDA device"):
        pack2bit.unpack2bit_gpu_to_gpu(torch.zeros((3, 3), dtype=torch.uint8), g)


def test_dz_image_hand_off_compiles_to_the_instructions_its_contract_names(tmp_path):
    """the cross-block hand-off of the dZ image (mlp_bwd_a kernels) is ordered without a fence.  What it relies on is
    which INSTRUCTIONS the compiler emits: write-through payload stores (`global_store_dword ... sc1`), `s_waitcnt vmcnt(0)` before the
    block is counted, a returning device-scope counter update (`global_atomic_add ... sc0`), and `sc1` loads by the block that arrives
    last.  A toolchain upgrade that changes any of them shows up here, not as a one-in-a-million wrong gradient."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "neural-admixture_amd", "csrc", "nadm_small_kernels.hip")
    out = tmp_path / "small.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    funcs, cur = {}, None
    for line in open(out):
        if line.startswith("_Z") and ":" in line:              # "<mangled name>:   ; @<mangled name>"
            cur = line.split(":")[0]
            funcs[cur] = []
        elif cur is not None and line.startswith("\t") and not line.startswith("\t."):
            funcs[cur].append(line.strip())
    kernels = {k: v for k, v in funcs.items() if "mlp_bwd_a" in k and any(i.startswith("global_atomic_add ") for i in v)}
    assert len(kernels) >= 2                                   # the instantiations of the MLP backward that build the image
    for name, ins in kernels.items():
        at = [i for i, s_ in enumerate(ins) if s_.startswith("global_atomic_add ")]
        assert len(at) == 1 and ins[at[0]].endswith("sc0"), (name, [ins[i] for i in at])       # ONE counter update, returning the old value
        before, after = ins[: at[0]], ins[at[0] + 1:]
        st = [i for i, s_ in enumerate(before) if s_.startswith("global_store_dword ") and s_.endswith(" sc1")]
        assert st, name                                        # the dZ payload goes through to memory
        waits = [i for i, s_ in enumerate(before) if s_.startswith("s_waitcnt") and "vmcnt(0)" in s_ and i > st[-1]]
        assert waits, name                                     # ... and is acknowledged before the block is counted
        assert any(s_.startswith("s_barrier") for s_ in before[waits[-1]:]), name      # every wave of the block has waited
        assert any(s_.startswith("global_load_dword ") and s_.endswith(" sc1") for s_ in after), name      # the last block reads past its L1 / L2 copies


def test_pass2_pair_product_loss_is_one_formula_for_the_three_calls():
    """The algebra of pass 2's fast loss (csrc/nadm_genotype_passes.hip, bce_loss_prod2), restated in numpy float32 and held against the
    oracle's BCE (neural_admixture.py:288 with ATen's -100 clamp): with q = sat(1 - d) - x (x = call / 2) twice a genotype's term is
    log f, f = | q^2 - x(1-x) | = (1-d)^2 | d(1-d) | d^2 for the calls 0 | 1 | 2.  And the fallback condition: f is EXACTLY 0 -- the wave
    then recomputes the tile pair in the exact form -- for every (call, d) whose reference term is a clamped logarithm, d = 1 + a rounding
    error included; never negative, so no product of two factors can hide one."""
    from oracle import nadm_oracle as O
    F = np.float32

    def fast_f(d, x):
        o = np.clip(F(1) - d, F(0), F(1)).astype(F)                                # v_sub_f32 ... clamp
        q = (o - x).astype(F)
        mh = (x.astype(np.float64) * x - x).astype(F)                              # v_pk_fma_f32 (one rounding; exact here)
        return np.abs((q.astype(np.float64) * q + mh).astype(F))                   # v_pk_fma_f32, | . | in the v_log

    rng = np.random.default_rng(5)
    n = 1 << 16
    d = np.exp(rng.uniform(np.log(1e-5), 0.0, n)).astype(F)
    d[::2] = (F(1) - d[::2]).astype(F)                                             # both ends of (0, 1)
    d = np.clip(d, F(1e-5), F(1) - F(1e-5)).astype(F)
    x = (rng.integers(0, 3, n) / 2).astype(F)
    f = fast_f(d, x)
    want = np.where(x == 0, (1 - d.astype(np.float64)) ** 2, np.where(x == 1, d.astype(np.float64) ** 2, d.astype(np.float64) * (1 - d.astype(np.float64))))
    assert np.all(f > 0)
    assert np.abs(f / want - 1).max() < 1.3e-7 / 1e-5                              # the ABSOLUTE rounding error of 1 - d, 6e-8, under d >= 1e-5
    fast = -0.5 * np.log(f.astype(np.float64)).sum()
    assert abs(fast - O.bce_sum(d, x)) / O.bce_sum(d, x) < 2e-6                    # the tolerance the GPU test holds the kernel to
    # the clamp cases: exact zeros, never a negative factor
    one_up = np.nextafter(F(1), F(2))
    for dv, calls_with_clamped_term in ((F(0), (0.5, 1.0)), (F(1), (0.0, 0.5)), (one_up, (0.0, 0.5)), (F(1.5), (0.0, 0.5))):
        for xv in (0.0, 0.5, 1.0):
            fv = fast_f(np.array([dv], F), np.array([xv], F))[0]
            assert fv >= 0
            assert (fv == 0) == (xv in calls_with_clamped_term), (dv, xv, fv)
            if fv != 0:                                                            # the other call's term is log 1 = 0 in the reference too
                assert fv == 1 and O.bce_sum(np.clip(np.array([dv], F), 0, 1), np.array([xv], F)) == 0


def test_matrix_passes_refuse_rows_of_4_gib():
    """Passes 1 and 2 form a tile's row address as ONE v_mad_u64_u32 of unsigned 32-bit factors (row index x row length): the launchers
    refuse ld >= 2^32 before anything is launched (include/nadm.h; no GPU needed: the check precedes the launch)."""
    from neural_admixture_amd._lib import lib
    fake = C.c_void_p(0x1000)                                   # never dereferenced: the call fails in the argument checks
    ld = 1 << 32
    assert lib.nadm_encode_fwd(fake, ld, fake, 16, 4 * ld, fake, 8, fake, None) != 0 and b"4 GiB" in lib.nadm_last_error()
    assert lib.nadm_decode_bce(fake, ld, fake, 16, 4 * ld, fake, 8, fake, 8, fake, fake, fake, 1, None) != 0 and b"4 GiB" in lib.nadm_last_error()
