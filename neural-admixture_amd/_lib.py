"""ctypes binding of the C ABI declared in include/nadm.h (csrc/libnadm.so).

There is no CPU fallback: if the shared library is missing or a call fails, a RuntimeError is
raised (the reference raises RuntimeError through TORCH_CHECK, pack2bit.cu:67-76)."""
import ctypes as C
import os

# torch must load first: it bundles its own libamdhip64.so.7, and the HIP runtime has to be the one
# torch initialised (same soname as /opt/rocm's; whichever is loaded first serves both).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NADM_LIB") or os.path.join(_HERE, "csrc", "libnadm.so")   # NADM_LIB: another build of the same library

MAX_HEADS = 32
MAX_BUCKETS = 8


class Heads(C.Structure):
    """Mirror of ``nadm_heads_t``."""
    _fields_ = [("n_heads", C.c_int32), ("C", C.c_int32), ("CP", C.c_int32), ("Hd", C.c_int32),
                ("SP", C.c_int32), ("n_small", C.c_int32),
                ("k", C.c_int32 * MAX_HEADS), ("kp", C.c_int32 * MAX_HEADS), ("qoff", C.c_int32 * MAX_HEADS),
                ("wk_off", C.c_int32 * MAX_HEADS), ("bk_off", C.c_int32 * MAX_HEADS),
                ("g_off", C.c_int32), ("w1_off", C.c_int32), ("b1_off", C.c_int32)]


class AdamArgs(C.Structure):
    """nadm_adam_t (include/nadm.h)."""
    _fields_ = [("m", C.c_void_p), ("v", C.c_void_p), ("lr", C.c_float), ("step", C.c_int32), ("grad_scale", C.c_float),
                ("reserved", C.c_int32)]


class FlatLayout(C.Structure):
    """nadm_flat_layout_t (include/nadm.h)."""
    _fields_ = [("n_flat", C.c_int64), ("off_v", C.c_int64), ("off_p", C.c_int64 * MAX_HEADS), ("slice_b", C.c_int64),
                ("slice_a", C.c_int64), ("msg_a_off", C.c_int64), ("n_buckets", C.c_int32), ("reserved", C.c_int32),
                ("bkt_off", C.c_int64 * (MAX_BUCKETS + 1)), ("bkt_slice", C.c_int64 * MAX_BUCKETS),
                ("bkt_m0", C.c_int64 * (MAX_BUCKETS + 1)), ("bkt_mom", C.c_int64 * MAX_BUCKETS)]


COMM_SLICES_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)      # (ctx, buf, slice | n, stream)
COMM_DESTROY_FN = C.CFUNCTYPE(None, C.c_void_p)
COMM_CHECK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class CommStruct(C.Structure):
    """nadm_comm_t (include/nadm.h)."""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("ctx", C.c_void_p), ("reduce_scatter", COMM_SLICES_FN),
                ("all_gather", COMM_SLICES_FN), ("all_reduce", COMM_SLICES_FN), ("destroy", COMM_DESTROY_FN),
                ("async_error", COMM_CHECK_FN)]


class PlanDesc(C.Structure):
    """nadm_plan_desc_t (include/nadm.h)."""
    _fields_ = ([("mode", C.c_int32), ("bmax", C.c_int32), ("M", C.c_int64), ("ld", C.c_int64), ("heads", Heads), ("xp", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("params", "grads", "m", "v", "zpart", "Z", "rinv", "Zn", "H", "Q", "dL", "dHpre", "dgp", "dZ",
                                             "dqpart", "losspart", "small_part", "zsum", "dqsum", "qimg")]
                + [("qimg_head_bytes", C.c_int64), ("dzimg", C.c_void_p), ("dzcnt", C.c_void_p), ("xg", C.c_void_p), ("loss_acc", C.c_void_p),
                   ("comm", C.POINTER(CommStruct)), ("comm_a", C.POINTER(CommStruct)), ("n_buckets", C.c_int32), ("p3_whole", C.c_int32),
                   ("debug", C.c_int32), ("reserved", C.c_int32), ("p2_slab", C.c_void_p), ("p2_cnt", C.c_void_p),
                   ("p3_slab", C.c_void_p), ("p3_cnt", C.c_void_p)])


MODE_SINGLE, MODE_DP, MODE_SNP = 0, 1, 2
T_NAMES = ("encode_fwd", "mlp_fwd", "decode_bce", "mlp_bwd", "encode_bwd", "sync_a", "sync_b")     # NADM_T_* slots


class MlpWeights(C.Structure):
    """nadm_mlp_weights_t (include/nadm.h)."""
    _fields_ = [("hd", C.POINTER(Heads)), ("Zn", C.c_void_p), ("H", C.c_void_p), ("dL", C.c_void_p), ("dHpre", C.c_void_p),
                ("dgp", C.c_void_p), ("small_part", C.c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"neural_admixture_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the training path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64
    HP = C.POINTER(Heads)
    sig = {
        "nadm_abi_version": (C.c_int, []),
        "nadm_last_error": (C.c_char_p, []),
        "nadm_pad_k": (C.c_int, [C.c_int]),
        "nadm_heads_init": (C.c_int, [HP, C.c_int, C.c_int, C.POINTER(i32), C.c_int]),
        "nadm_encode_chunks": (i64, [i64]),
        "nadm_decode_chunks": (i64, [i64, C.c_int]),
        "nadm_decode_chunk_snps": (i32, [C.c_int]),
        "nadm_sample_splits": (i32, [C.c_int]),
        "nadm_pack2bit_host": (C.c_int, [vp, vp, i64, i64, i64]),
        "nadm_pack2bit": (C.c_int, [vp, vp, i64, i64, i64, vp]),
        "nadm_unpack2bit": (C.c_int, [vp, vp, i64, i64, i64, vp]),
        "nadm_bed_to_packed": (C.c_int, [vp, i64, i64, vp, i64, C.POINTER(i64), i32, C.POINTER(i32)]),
        "nadm_bed_to_packed_dev": (C.c_int, [vp, i64, i64, vp, i64, vp, i32, vp, vp]),
        "nadm_encode_fwd": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, vp]),
        "nadm_encode_fwd_part": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, i64, vp]),
        "nadm_pca_project": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, vp]),
        "nadm_pca_project_t": (C.c_int, [vp, i64, vp, i32, i64, vp, vp, i32, vp, vp]),
        "nadm_loglik_blocks": (i64, [i64]),
        "nadm_loglik": (C.c_int, [vp, i64, i64, i64, vp, vp, i32, i32, C.c_double, vp, vp]),
        "nadm_savetxt_f32": (C.c_int, [C.c_char_p, vp, i64, i64, i64]),
        "nadm_gmm_fit_means": (C.c_int, [vp, i64, i32, i32, vp, i32, C.c_double, i32, C.c_double, vp, C.POINTER(C.c_double), C.POINTER(i32)]),
        "nadm_gmm_fit_means_dev": (C.c_int, [vp, i64, i32, i32, vp, i32, C.c_double, i32, C.c_double, vp, C.POINTER(C.c_double), C.POINTER(i32), vp]),
        "nadm_mlp_fwd": (C.c_int, [HP, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]),
        "nadm_decode_bce": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, i32, vp, vp, vp, i32, vp]),
        "nadm_decode_bce_gather": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, i32, vp, vp, vp, i32, vp, vp]),
        "nadm_decode_bce_step": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp]),
        "nadm_decode_bce_images": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
        "nadm_decode_bce_sliced": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp]),
        "nadm_decode_slices": (i32, [i32, i64, i32]),
        "nadm_decode_slices_max": (i32, [i32, i64, i32]),
        "nadm_decode_slab_floats": (i64, [i64, i32, i32]),
        "nadm_mlp_fwd_images": (C.c_int, [HP, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, i64, vp]),
        "nadm_q_image_bytes": (C.c_int64, [i32]),
        "nadm_encode_fwd_small": (C.c_int, [vp, i64, vp, i32, i64, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp]),
        "nadm_encode_bwd_step": (C.c_int, [vp, i64, vp, i32, i64, vp, vp, i32, vp, vp, vp, vp, i32, vp]),
        "nadm_encode_slices": (i32, [i32, i64, i32]),
        "nadm_encode_slices_max": (i32, [i32, i64, i32]),
        "nadm_encode_slab_floats": (i64, [i64, i32, i32]),
        "nadm_encode_bwd_chunks": (i64, [i64]),
        "nadm_encode_bwd_sliced": (C.c_int, [vp, i64, vp, i32, i64, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
        "nadm_small_grads": (C.c_int, [vp, i32, i32, vp, vp, vp, vp]),
        "nadm_mlp_bwd": (C.c_int, [HP, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]),
        "nadm_mlp_bwd_image": (C.c_int, [HP, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
        "nadm_mlp_bwd_weights": (C.c_int, [HP, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "nadm_sum_rows": (C.c_int, [vp, i64, i64, vp, vp]),
        "nadm_supervised_ce": (C.c_int, [vp, i32, i32, i32, vp, vp, i32, i32, f32, vp, vp, vp]),
        "nadm_encode_bwd": (C.c_int, [vp, i64, vp, i32, i64, vp, vp, i32, vp, i32, vp]),
        "nadm_dz_image_bytes": (C.c_int64, [i32]),
        "nadm_dz_image_tile_bytes": (C.c_int64, []),
        "nadm_batch_copy_bytes": (C.c_int64, [i32, i64]),
        "nadm_dz_image": (C.c_int, [vp, i32, i32, vp, vp]),
        "nadm_vcf_parse_gt": (C.c_int, [C.c_char_p, i64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]),
        "nadm_adam": (C.c_int, [vp, vp, vp, vp, i64, i64, f32, i32, f32, vp]),
        "nadm_synth_packed": (C.c_int, [vp, i64, i64, i64, i64, vp, vp, i32, f32, u64, vp]),
        "nadm_flat_layout": (C.c_int, [HP, i64, i32, i32, C.POINTER(FlatLayout)]),
        "nadm_comm_rccl_probe": (C.c_int, [C.c_char_p]),
        "nadm_comm_rccl_unique_id": (C.c_int, [C.c_char_p, vp]),
        "nadm_comm_rccl": (C.c_int, [C.c_char_p, vp, i32, i32, i32, C.POINTER(C.POINTER(CommStruct))]),
        "nadm_comm_emulated": (C.c_int, [i32, C.POINTER(C.POINTER(CommStruct))]),
        "nadm_comm_free": (None, [C.POINTER(CommStruct)]),
        "nadm_comm_abort": (None, [C.POINTER(CommStruct)]),
        "nadm_plan_create": (C.c_int, [C.POINTER(PlanDesc), C.POINTER(vp)]),
        "nadm_plan_destroy": (None, [vp]),
        "nadm_plan_set_rows": (C.c_int, [vp, vp]),
        "nadm_plan_set_labels": (C.c_int, [vp, vp, i32, f32]),
        "nadm_plan_set_state": (C.c_int, [vp, i32, i32]),
        "nadm_plan_step_count": (i32, [vp]),
        "nadm_plan_p_in_unit_range": (i32, [vp]),
        "nadm_step": (C.c_int, [vp, vp, i32, f32, i32, vp]),
        "nadm_plan_flush": (C.c_int, [vp, vp]),
        "nadm_plan_infer": (C.c_int, [vp, vp, i32, vp]),
        "nadm_plan_timing": (C.c_int, [vp, C.c_uint32]),
        "nadm_plan_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(i32)]),
        "nadm_plan_bucket_ms": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(i32)]),
        "nadm_plan_poisoned": (i32, [vp]),
        "nadm_calib_clock": (C.c_int, [i32, vp, i32, vp, vp]),
        "nadm_wall_clock_khz": (i64, []),
        "nadm_clock_probe": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    for name in ("nadm_test_force_slices", "nadm_test_force_generic_mlp", "nadm_test_force_p3_slices"):      # the TEST build only (csrc/libnadm_testhooks.so, -DNADM_TEST_HOOKS)
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = None, [i32]
    if lib.nadm_abi_version() != 14:
        raise RuntimeError("neural_admixture_amd: libnadm.so ABI version mismatch")
    return lib, tuple(sig)


lib, EXPORTS = _load()


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib.nadm_last_error()
        raise RuntimeError(f"libnadm {what}: {msg.decode() if msg else 'error'} (status {status})")


def ptr(t):
    """Raw device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
