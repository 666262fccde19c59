"""Soak tests of the step's hand-offs.

The production step carries work from one launch into another: the MLP backward's block that completes a 32-sample group LAST builds
pass 3's FP6 operand image of dZ (cross-block hand-off inside one launch: store-through + counter, no fence -- DESIGN 4.3), the MLP
forward leaves Q as pass 2's operand images, the small-parameter update rides in the NEXT step's pass 1.  A rare ordering failure
would corrupt dV for 32 samples silently; short parity tests would not see a one-in-10^5 event.  Here thousands of consecutive steps
run with changing batch sizes and the result must equal, BIT FOR BIT, the same steps issued as the unfused launch sequence
(tests/unfused_step.py: every hand-off a plain launch of its own)."""
import os

import numpy as np
import pytest
import torch

from oracle import nadm_oracle as O
from unfused_step import unfused_step
from test_gpu_parity import decode_dz_image

pytestmark = pytest.mark.gpu


def _engines(M, ks, Hd, N, seed, bmax=800):
    import neural_admixture_amd as na
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    Gm = O.synth_genotypes(N, M, max(ks), seed=seed, missing=0.01)
    V0 = (rng.standard_normal((M, 8)) / np.sqrt(M)).astype(np.float32)
    P0 = rng.uniform(0.02, 0.98, size=(sum(ks), M)).astype(np.float32)
    p = O.make_params(seed, V0, P0, Hd, ks)
    small = np.concatenate([p.g, p.W1.reshape(-1), p.b1] + [np.concatenate([p.Wk[h].reshape(-1), p.bk[h]]) for h in range(len(ks))]).astype(np.float32)
    out = []
    Gt = torch.from_numpy(np.ascontiguousarray(Gm))
    for _ in range(2):
        e = na.Engine(M, 8, Hd, ks, dev, bmax)
        e.load_params(V0, P0, small)
        e.pack_from_host(Gt)
        out.append(e)
    return out


def _same_state(a, b):
    return all(torch.equal(x, y) for x, y in ((a.big, b.big), (a.mbig, b.mbig), (a.vbig, b.vbig), (a.small, b.small), (a.msmall, b.msmall),
                                              (a.vsmall, b.vsmall)))


def test_5000_production_steps_equal_the_unfused_sequence_bit_for_bit():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neural_admixture_amd._lib import lib, check, ptr
    M, N, steps = 60_000, 4000, 5000
    prod, ref = _engines(M, [8], 1024, N, seed=17)
    dev = prod.device
    gen = torch.Generator().manual_seed(5)
    sizes = (800, 790, 37, 800)
    lr = 2e-3
    alone = torch.zeros_like(prod._dzimg)
    checked = 0
    for s in range(steps):
        b = sizes[s % 4]
        idx = torch.randint(0, N, (b,), generator=gen, dtype=torch.int32).to(dev)
        with_loss = (s % 7) != 3
        prod.train_step(idx, b, lr, with_loss)
        unfused_step(ref, idx, b, lr, with_loss)
        if s % 50 == 49:
            # the image the MLP backward's last blocks left behind == the image a launch of its own builds from the same dZ
            check(lib.nadm_dz_image(ptr(prod._dZ), b, prod.lay.CP, ptr(alone), None), "dz_image")
            torch.cuda.synchronize()
            # the tiles the batch fills, byte for byte; the tile it ends in by value (its groups past the batch hold zeros either way, but
            # nadm_step clears them with a memset and nadm_dz_image writes zero PIECES: other scale bytes, same numbers)
            nb = int(lib.nadm_dz_image_bytes((b // 128) * 128)) if b >= 128 else 0
            assert torch.equal(prod._dzimg[:nb], alone[:nb]), f"step {s}: fused dZ image differs from nadm_dz_image"
            if b % 128:
                CP, tb = prod.lay.CP, int(lib.nadm_dz_image_tile_bytes())
                last = [t[nb: nb + tb].cpu().numpy() for t in (prod._dzimg, alone)]
                assert np.array_equal(decode_dz_image(last[0], b % 128, CP), decode_dz_image(last[1], b % 128, CP)), f"step {s}"
            assert int(prod._dzcnt.abs().sum().item()) == 0, f"step {s}: group counters did not return to zero"
            assert torch.equal(prod._dZ[: b * prod.lay.CP], ref._dZ[: b * ref.lay.CP]), f"step {s}: dZ differs"
            checked += 1
    torch.cuda.synchronize()
    assert checked == steps // 50
    assert prod.read_loss() == ref.read_loss()
    assert _same_state(prod, ref), "5000 production steps left other bits than the unfused launch sequence"


def test_2000_multihead_production_steps_equal_the_unfused_sequence():
    """The same with three heads (two pass-2 streams in flight, one Q image region per head) and a narrow hidden layer."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    M, N, steps = 20_000, 1500, 2000
    prod, ref = _engines(M, [2, 5, 9], 256, N, seed=23)
    dev = prod.device
    gen = torch.Generator().manual_seed(9)
    sizes = (800, 65, 790, 31)
    for s in range(steps):
        b = sizes[s % 4]
        idx = torch.randint(0, N, (b,), generator=gen, dtype=torch.int32).to(dev)
        prod.train_step(idx, b, 2e-3, s % 5 == 0)
        unfused_step(ref, idx, b, 2e-3, s % 5 == 0)
    torch.cuda.synchronize()
    assert int(prod._dzcnt.abs().sum().item()) == 0
    assert prod.read_loss() == ref.read_loss()
    assert _same_state(prod, ref)


def test_600_large_batch_steps_with_both_passes_in_sample_slices_equal_the_unfused_sequence():
    """An SNP-sharded rank's shape in miniature (r06): 4200-row batches on 30k SNPs -- pass 2 AND pass 3 run in sample slices (park, count,
    the last block adds the partials: the hand-off happens in two kernels of every step) -- against the unfused launch sequence, which takes
    the library's same cut; ragged and tiny batches in between switch the sliced forms off and on."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neural_admixture_amd._lib import lib
    M, N, steps = 30_000, 4400, 600
    sizes = (4200, 4100, 37, 4200, 800)
    assert lib.nadm_encode_slices(4200, M, 8) == 2 and lib.nadm_decode_slices(4200, M, 8) > 1 and lib.nadm_encode_slices(800, M, 8) == 1
    prod, ref = _engines(M, [8], 1024, N, seed=31, bmax=4200)
    dev = prod.device
    gen = torch.Generator().manual_seed(13)
    for s in range(steps):
        b = sizes[s % len(sizes)]
        idx = torch.randint(0, N, (b,), generator=gen, dtype=torch.int32).to(dev)
        prod.train_step(idx, b, 2e-3, s % 3 == 0)
        unfused_step(ref, idx, b, 2e-3, s % 3 == 0)
    torch.cuda.synchronize()
    assert _same_state(prod, ref)
    assert int(prod._p3_cnt.abs().sum().item()) == 0 and int(prod._p2_cnt.abs().sum().item()) == 0 and int(prod._dzcnt.abs().sum().item()) == 0
    assert prod.read_loss() == ref.read_loss()


def test_1000_data_parallel_steps_on_a_one_rank_rccl_communicator_equal_the_plain_step():
    """nadm_step in NADM_MODE_DP on a 1-rank RCCL communicator: message A on the side stream, message B in four SNP-range buckets on the
    second one (pass 3 range by range, the next pass 1 in the same ranges on streams of their own: event hand-offs in both directions
    for every bucket, every step), Adam as launches of its own -- 1000 steps with changing batch sizes must leave the bits of the
    single-GPU step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import neural_admixture_amd as na
    from neural_admixture_amd.comm import rccl_comm
    dev = torch.device("cuda:0")
    comm = rccl_comm(0, 1)
    M, N, steps = 60_000, 4000, 1000
    plain, tmp = _engines(M, [8], 1024, N, seed=29)
    ddp = na.Engine(M, 8, 1024, [8], dev, 800, mode="dp", comm=comm, n_buckets=4)
    assert ddp.lay.n_buckets == 4
    ddp.pflat.copy_(tmp.pflat)
    ddp.set_packed(tmp.xp)
    gen = torch.Generator().manual_seed(3)
    sizes = (800, 790, 37, 800)
    for s in range(steps):
        b = sizes[s % 4]
        idx = torch.randint(0, N, (b,), generator=gen, dtype=torch.int32).to(dev)
        ddp.train_step(idx, b, 2e-3, s % 3 == 0)
        plain.train_step(idx, b, 2e-3, s % 3 == 0)
    torch.cuda.synchronize()
    assert int(ddp._dzcnt.abs().sum().item()) == 0
    assert ddp.read_loss() == plain.read_loss()
    assert _same_state(ddp, plain)
    del ddp
    comm.close()
