"""The single-GPU training step as its UNFUSED launch sequence, straight on the C ABI (TEST INFRASTRUCTURE).

The production step (Engine.train_step) hands Q to pass 2 as ready-made operand images, lets the MLP backward's last blocks build
pass 3's operand image of dZ, and leaves the small-parameter update to side blocks of the NEXT step's pass 1.  Here every one of
those is a plain launch of its own at the place the arithmetic belongs: nadm_mlp_fwd (no images), nadm_mlp_bwd (no image) +
nadm_dz_image, nadm_small_grads right behind pass 3.  Same element functions, same reduction orders: the two sequences must
leave the same bits in every parameter and moment (tests/test_soak_handoffs.py)."""
import ctypes as C

import torch

from neural_admixture_amd._lib import lib, check, ptr, AdamArgs, MlpWeights

NADM_X_CLEAN = 1


def unfused_step(e, idx: torch.Tensor, b: int, lr: float, with_loss: bool = True) -> None:
    fsz = 4
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = e.lay
    big, mbig, vbig, gbig, small = e.big, e.mbig, e.vbig, e.gbig, e.small      # (the accessors settle what the last step still owes)
    msmall, vsmall, gsmall = e.msmall, e.vsmall, e.gsmall
    check(lib.nadm_encode_fwd(ptr(e.xp), e.ld, ptr(idx), b, L.M, ptr(big), L.CP, ptr(e.zpart), st), "encode_fwd")
    check(lib.nadm_mlp_fwd(C.byref(L.heads), ptr(small), ptr(e.zpart), L.enc_chunks, b, ptr(e.Z), ptr(e.rinv), ptr(e.Zn), ptr(e.H),
                           ptr(e._Q), st), "mlp_fwd")
    e.step_count += 1
    dq_offs, _ = L.dq_offsets(b)
    loss_offs = L.loss_offsets()
    tiled = L.CP <= 8
    xg = e._xg if tiled else None
    for h, kp in enumerate(L.kp):
        ad = AdamArgs(mbig.data_ptr() + L.p_off[h] * fsz, vbig.data_ptr() + L.p_off[h] * fsz, lr, e.step_count, 1.0, 0)
        args = (ptr(e.xp), e.ld, ptr(idx), b, L.M, C.c_void_p(big.data_ptr() + L.p_off[h] * fsz), kp,
                C.c_void_p(e._Q.data_ptr() + L.qoff[h] * fsz), L.SP, C.c_void_p(gbig.data_ptr() + L.p_off[h] * fsz),
                C.c_void_p(e.dqpart.data_ptr() + dq_offs[h] * fsz), C.c_void_p(e.losspart.data_ptr() + loss_offs[h] * fsz),
                ((1 if e.p_unit else 3) if with_loss else 0), ptr(xg) if (h == 0 and tiled) else None, C.byref(ad))
        slices = int(lib.nadm_decode_slices(b, L.M, kp)) if e._p2_slab is not None else 1
        if slices > 1:                          # the library's cut of the batch into sample slices (the sum over them has an order of its own)
            check(lib.nadm_decode_bce_sliced(*args, None, slices, C.c_void_p(e._p2_slab.data_ptr() + e._p2_slab_off[h] * fsz),
                                             C.c_void_p(e._p2_cnt.data_ptr() + loss_offs[h] * 4), st), "decode_bce_sliced")
        else:
            check(lib.nadm_decode_bce_step(*args, st), "decode_bce_step")
    n_loss = L.n_loss
    if e.labels is not None:
        check(lib.nadm_supervised_ce(ptr(e._Q), L.SP, L.ks[0], L.kp[0], ptr(e.labels), ptr(idx), b, e.n_classes, e.sup_weight, ptr(e.dqpart),
                                     C.c_void_p(e.losspart.data_ptr() + L.n_loss * fsz), st), "supervised_ce")
        n_loss += 1
    check(lib.nadm_mlp_bwd(C.byref(L.heads), ptr(small), ptr(e.dqpart), L.M, b, ptr(e.Z), ptr(e.rinv), ptr(e.Zn), ptr(e.H), ptr(e._Q),
                           ptr(e.dL), ptr(e.dHpre), ptr(e.dgp), ptr(e.small_part), ptr(e._dZ), None, ptr(e.losspart),
                           n_loss if with_loss else 0, ptr(e.loss_acc), st), "mlp_bwd")
    dzimg = None
    if tiled:
        check(lib.nadm_dz_image(ptr(e._dZ), b, L.CP, ptr(e._dzimg), st), "dz_image")
        dzimg = ptr(e._dzimg)
    mw = MlpWeights(C.pointer(L.heads), e.Zn.data_ptr(), e.H.data_ptr(), e.dL.data_ptr(), e.dHpre.data_ptr(), e.dgp.data_ptr(),
                    e.small_part.data_ptr())
    av = AdamArgs(mbig.data_ptr(), vbig.data_ptr(), lr, e.step_count, 1.0, 0)
    src, rows, flags = (xg, e._iota, NADM_X_CLEAN) if tiled else (e.xp, idx, 0)
    slices = int(lib.nadm_encode_slices(b, L.M, L.CP)) if (e._p3_slab is not None and tiled) else 1
    if slices > 1:
        check(lib.nadm_encode_bwd_sliced(ptr(src), e.ld, ptr(rows), b, L.M, ptr(e._dZ), dzimg, L.CP, ptr(big), ptr(gbig), C.byref(av), C.byref(mw),
                                         flags, slices, ptr(e._p3_slab), ptr(e._p3_cnt), st), "encode_bwd_sliced")
    else:
        check(lib.nadm_encode_bwd_step(ptr(src), e.ld, ptr(rows), b, L.M, ptr(e._dZ), dzimg, L.CP, ptr(big), ptr(gbig), C.byref(av), C.byref(mw),
                                       flags, st), "encode_bwd_step")
    sa = AdamArgs(msmall.data_ptr(), vsmall.data_ptr(), lr, e.step_count, 1.0, 0)
    check(lib.nadm_small_grads(ptr(e.small_part), int(lib.nadm_sample_splits(b)), L.n_small, ptr(gsmall), ptr(small), C.byref(sa), st),
          "small_grads")
    e.p_unit = True
    e._qimg_b = e._dzimg_b = -1
