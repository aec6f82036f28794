"""ctypes binding of the C-ABI HIP library (``include/bigc_hip.h``).

There is no fallback: if ``lib/libbigc_hip.so`` is missing or a call fails, an exception
is raised (``BigcError``; shape violations map to ``AssertionError`` like the reference's
inline asserts).  Build the library with ``python __graft_entry__.py`` or
``make -C vidsgg-big_amd/csrc``.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libbigc_hip.so")


class BigcError(RuntimeError):
    pass


class bigc_config(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "variant", "num_enti_cats", "num_pred_cats", "dim_feat", "dim_extra", "dim_clsme", "dim_enti",
        "dim_pred", "dim_att", "dim_ffn", "enco_pool_len", "n_enco_layers", "n_deco_layers",
        "n_att_head", "num_querys", "use_clsme", "has_enti_emb")]


class bigc_batch(ctypes.Structure):
    _fields_ = [("n_vid", c_int32), ("n_trk", c_int32), ("n_rows", c_int32), ("n_conv", c_int32),
                ("max_trk", c_int32),
                ("feats", c_void_p), ("boxes", c_void_p), ("trk_row_off", c_void_p),
                ("trk_conv_off", c_void_p), ("trk_vid", c_void_p), ("trk_dura", c_void_p),
                ("trk_cat", c_void_p), ("trk_score", c_void_p), ("vid_trk_off", c_void_p),
                ("vid_tmax", c_void_p), ("vid_wh", c_void_p)]


class bigc_outputs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "pred_logits", "att", "pred_query", "enti2enco", "enco_output", "soid",
        "quint", "scores", "dura", "qid", "n_out", "range_flag")]


class grd_config(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("dim_feat", "dim_clsme", "dim_hidden", "num_bins", "n_enti_emb", "n_pred_emb")]


class grd_outputs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("pooled_se", "bins_probs", "bins_mask", "status", "regrs", "conf", "cls",
                                        "video_enc", "query_enc", "combined_enc")]


_SIGS = {
    "bigc_last_error": (c_char_p, []),
    "bigc_build_info": (c_char_p, []),
    "bigc_prof_enable": (None, [c_int]),
    "bigc_prof_reset": (None, []),
    "bigc_prof_collect": (c_int, []),
    "bigc_prof_get": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_longlong), POINTER(ctypes.c_double),
                              POINTER(ctypes.c_double)]),
    "bigc_prof_family_count": (c_int, []),
    "bigc_prof_family_name": (c_char_p, [c_int]),
    "bigc_linear_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "bigc_gemm_ps": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, ctypes.c_long,
                             c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "bigc_split_rows_f32": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, ctypes.c_long, c_int, c_void_p]),
    "bigc_linear_set_exact": (c_int, [c_int]),
    "bigc_linear_get_exact": (c_int, []),
    "bigc_layernorm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    "bigc_layernorm_add_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "bigc_add_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "bigc_mha_core_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bigc_role_attention_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_float]),
    "bigc_role_values_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int]),
    "bigc_role_fused_proj_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, ctypes.c_long, c_void_p, c_void_p, c_int,
                                         c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float]),
    "bigc_role_fused_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_int, c_float]),
    "bigc_assemble_rows_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_int]),
    "bigc_bbox_feat_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "bigc_conv_index": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "bigc_adaptive_maxpool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "bigc_stretched_mean": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int]),
    "bigc_head_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                 POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), c_void_p, c_int, c_int,
                                 c_void_p, c_void_p, c_int, c_void_p]),
    "bigc_construct_triplets": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bigc_create": (c_int, [POINTER(bigc_config), POINTER(c_void_p)]),
    "bigc_destroy": (None, [c_void_p]),
    "bigc_set_param": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "bigc_finalize": (c_int, [c_void_p, c_void_p]),
    "bigc_workspace_bytes": (c_size_t, [c_void_p, POINTER(bigc_batch)]),
    "bigc_forward": (c_int, [c_void_p, c_void_p, POINTER(bigc_batch), c_void_p, c_size_t, c_int,
                             POINTER(bigc_outputs)]),
    "bigc_pair_workspace_bytes": (c_size_t, [c_void_p, POINTER(bigc_batch), ctypes.c_long]),
    "bigc_pair_train_workspace_bytes": (c_size_t, [c_void_p, POINTER(bigc_batch), ctypes.c_long]),
    "bigc_pair_train_forward": (c_int, [c_void_p, c_void_p, POINTER(bigc_batch), c_void_p, ctypes.c_long, c_void_p, c_size_t, c_void_p]),
    "bigc_pair_train_backward": (c_int, [c_void_p, c_void_p, POINTER(bigc_batch), c_void_p, ctypes.c_long, c_void_p, c_size_t, c_void_p, c_int]),
    "bigc_bce_logits_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.c_long, c_int, c_float, c_void_p, c_void_p, c_int]),
    "bigc_pair_triplets_workspace_bytes": (c_size_t, [c_int, c_int, ctypes.c_long, c_int, ctypes.c_long]),
    "bigc_pair_triplets": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_long,
                                   c_void_p, c_void_p, ctypes.c_long, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "bigc_pair_forward": (c_int, [c_void_p, c_void_p, POINTER(bigc_batch), c_void_p, ctypes.c_long, c_void_p, c_size_t, c_void_p, c_void_p]),
    # training step
    "bigc_transpose_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "bigc_linear_dw_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                   c_int, c_int, c_int]),
    "bigc_relu_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long]),
    "bigc_axpy_f32": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long]),
    "bigc_layernorm_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    "bigc_reduce_periodic_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "bigc_maxpool_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "bigc_conv_gather_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "bigc_mha_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                 c_int, c_int, c_int, ctypes.c_long, c_int, c_int]),
    "bigc_role_attention_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_int, c_float]),
    "bigc_role_small_gemm_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                         c_int, c_int, c_int, c_int, c_int]),
    "bigc_viou_matrix": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_void_p]),
    "bigc_viou_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "bigc_match_cost": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_int, c_float, c_float, c_void_p]),
    "bigc_loss_grad": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float,
                               c_void_p, c_void_p, c_int, c_void_p]),
    "bigc_set_grad": (c_int, [c_void_p, c_char_p, c_void_p]),
    "bigc_set_dropout": (c_int, [c_void_p, c_float, ctypes.c_uint32]),
    "bigc_dropout_f32": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_float, ctypes.c_uint32, ctypes.c_uint32]),
    "bigc_mha_core_drop_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, ctypes.c_uint32, ctypes.c_uint32]),
    "bigc_mha_bwd_drop_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                      c_int, c_int, c_int, ctypes.c_long, c_int, c_int, c_float, ctypes.c_uint32, ctypes.c_uint32]),
    "bigc_train_workspace_bytes": (c_size_t, [c_void_p, POINTER(bigc_batch)]),
    "bigc_train_forward": (c_int, [c_void_p, c_void_p, POINTER(bigc_batch), c_void_p, c_size_t, c_void_p, c_void_p]),
    "bigc_train_backward": (c_int, [c_void_p, c_void_p, POINTER(bigc_batch), c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    # grounding scorer
    "bigc_dwconv_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    "grd_coattention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "grd_postprocess": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "grd_create": (c_int, [POINTER(grd_config), POINTER(c_void_p)]),
    "grd_destroy": (None, [c_void_p]),
    "grd_set_param": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "grd_finalize": (c_int, [c_void_p, c_void_p]),
    "grd_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "grd_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p,
                            c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_size_t, POINTER(grd_outputs)]),
    "grd_set_grad": (c_int, [c_void_p, c_char_p, c_void_p]),
    "grd_set_dropout": (c_int, [c_void_p, c_float, ctypes.c_uint32]),
    "grd_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "grd_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "grd_loss_grad": (c_int, [c_void_p] + [c_void_p] * 5 + [c_void_p] * 3 + [c_int] + [c_void_p] * 2 + [c_int, c_int, c_int]
                      + [c_float] * 6 + [c_void_p] * 5),
    "grd_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
}

# grounding entry points are appended by bigc_hip._lib_grd when that part of the library exists
EXPORTED_SYMBOLS = list(_SIGS)

_lib = None


def load():
    """Load the shared library (once). Raises BigcError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BigcError("HIP library %s not found: build it with `python __graft_entry__.py` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # torch must be imported first: it ships its own libamdhip64 and the process must end up with ONE HIP
    # runtime (loading ours first binds the library to /opt/rocm's copy, whose device list is then empty)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Map a bigc_status to an exception (ARG -> AssertionError as the reference's asserts)."""
    if rc == 0:
        return
    msg = load().bigc_last_error().decode()
    if rc == -1:
        raise AssertionError(msg)
    raise BigcError("bigc_hip error %d: %s" % (rc, msg))


def ptr(t):
    """Device (or host) address of a torch tensor, or None."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
