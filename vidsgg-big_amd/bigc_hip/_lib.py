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
                ("vid_tmax", c_void_p), ("vid_wh", c_void_p),
                ("feats_hr8", c_void_p), ("ld_feats_hr8", ctypes.c_int64), ("feats_hr8_flag", c_void_p),
                ("extra", c_void_p), ("ld_extra", ctypes.c_int64), ("feat_row_ptr", c_void_p)]


class bigc_outputs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "pred_logits", "att", "pred_query", "enti2enco", "enco_output", "soid",
        "quint", "scores", "dura", "qid", "n_out", "range_flag")]


class grd_config(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("dim_feat", "dim_clsme", "dim_hidden", "num_bins", "n_enti_emb", "n_pred_emb")]


class grd_outputs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("pooled_se", "bins_probs", "bins_mask", "status", "regrs", "conf", "cls",
                                        "video_enc", "query_enc", "combined_enc")]


class grd_batch(ctypes.Structure):
    _fields_ = [("n_vid", ctypes.c_int32), ("n_query", ctypes.c_int32), ("max_T", ctypes.c_int32), ("total_T", ctypes.c_int64),
                ("total_rows", ctypes.c_int64), ("vid_row_off", c_void_p), ("q_vid", c_void_p), ("q_row_off", c_void_p), ("vid_len", c_void_p)]


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
    "bigc_gemm_ps_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "bigc_gemm_ps_ln": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_void_p, c_int, c_void_p, c_void_p, ctypes.c_long, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "bigc_split_rows_f32": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, ctypes.c_long, c_int, c_void_p]),
    "bigc_linear_hr8out_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, ctypes.c_long,
                                       c_int, c_int, c_int, c_int, c_void_p]),
    "bigc_set_mode": (c_int, [c_void_p, c_int, c_int]),
    "grd_set_mode": (c_int, [c_void_p, c_int, c_int]),
    "bigc_set_accum_policy": (c_int, [c_int]),
    "bigc_get_accum_policy": (c_int, []),
    "bigc_linear_set_exact": (c_int, [c_int]),
    "bigc_linear_splitk_allow": (c_int, [c_int]),
    "bigc_linear_get_exact": (c_int, []),
    "bigc_linear_set_fast": (c_int, [c_int]),
    "bigc_linear_get_fast": (c_int, []),
    "bigc_layernorm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    "bigc_layernorm_hr8_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_float, c_void_p]),
    "bigc_layernorm_res_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_float,
                                       c_void_p, c_int, c_void_p]),
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
    "bigc_pack_tracklets": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_long, c_int, c_int, c_int, c_void_p, ctypes.c_long,
                                    c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, c_void_p, c_void_p]),
    "bigc_row_pointers": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "bigc_stretched_mean_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "bigc_pack_video_meta": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "bigc_compact_triplets": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "bigc_forward_reads_hr8": (c_int, [c_void_p]),
    "bigc_train_reads_rows": (c_int, [c_void_p]),
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
    "bigc_role_small_gemm_pair_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                              c_int, c_int, c_int, c_int, c_int]),
    "bigc_viou_matrix": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_void_p]),
    "bigc_viou_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "bigc_match_cost": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_int, c_float, c_float, c_void_p]),
    "bigc_loss_grad": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float,
                               c_void_p, c_void_p, c_int, c_void_p]),
    "bigc_loss_grad_dev": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p,
                                   c_void_p, c_void_p, c_int, c_void_p]),
    "bigc_align_adjacency": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_float, c_void_p]),
    "bigc_hungarian": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "bigc_copy_mapped": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long]),
    "bigc_stamp": (c_int, [c_void_p, c_void_p, ctypes.c_ulonglong, c_void_p]),
    "bigc_set_grad": (c_int, [c_void_p, c_char_p, c_void_p]),
    "bigc_rebase_grads": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
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
    "grd_batch_workspace_bytes": (c_size_t, [c_void_p, POINTER(grd_batch)]),
    "grd_forward_batch": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(grd_batch), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_float, c_float, c_float, c_float, c_void_p, c_size_t, POINTER(grd_outputs)]),
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
    fit_threads_to_cpu_quota()
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



class _Staging(object):
    """One pinned host block per device, allocated once and used as two halves in turn.  Pinned blocks from torch's caching
    host allocator are only recycled once the copy that read them has run; a host that runs ahead of the GPU therefore makes
    it allocate new pinned memory every step, and every such allocation maps pages into the GPU's address space -- measured as
    random 40 ms stalls of the GPU queue in the 64-video training step."""

    def __init__(self, nbytes=16 << 20):
        import torch
        self.buf = torch.empty((nbytes,), dtype=torch.uint8).pin_memory()
        self.half, self.off, self.cur, self.events = nbytes // 2, 0, 0, [None, None]
        self.down = {}

    def put(self, t, device, by_kernel=False):
        import torch
        n = t.numel() * t.element_size()
        n_al = (n + 255) & ~255
        if n_al > self.half:
            return None
        if self.off + n_al > self.half:          # move to the other half once the copies that read it last time have run
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))          # the stream of the TARGET device (the copies below are queued there)
            self.events[self.cur] = ev
            self.cur ^= 1
            self.off = 0
            if self.events[self.cur] is not None:
                self.events[self.cur].synchronize()
        start = self.cur * self.half + self.off
        view = self.buf[start:start + n].view(t.dtype).view(t.shape)
        ctypes.memmove(view.data_ptr(), t.data_ptr(), n)          # (not view.copy_(t): from 32 K elements on that is a job for torch's thread pool)
        self.off += n_al
        with torch.cuda.device(device):               # the copy goes to the current stream of `device`, whatever the caller's current device is
            if not by_kernel:
                return view.to(device, non_blocking=True)
            # a kernel reads the pinned block through its mapping into the device's address space (csrc/ingest.hip: bigc_copy_mapped)
            out = torch.empty(t.shape, dtype=t.dtype, device=device)
            check(load().bigc_copy_mapped(current_stream(), out.data_ptr(), view.data_ptr(), n))
            return out

    def download_buffer(self, shape, dtype, slot="default"):
        """a pinned host tensor for device-to-host copies whose result the host consumes before the slot's next use"""
        import torch
        n = 1
        for s in shape:
            n *= int(s)
        cur = self.down.get(slot)
        if cur is None or cur.numel() < n or cur.dtype != dtype:
            cur = self.down[slot] = torch.empty((max(2 * n, 1 << 16),), dtype=dtype).pin_memory()
        return cur[:n].view(shape)


def cpu_quota():
    """CPUs this process may keep busy: the smaller of its affinity mask and its cgroup's CPU quota (cgroup v2 ``cpu.max``, v1
    ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``); None when nothing limits it"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    return n if quota is None else min(n, max(1, int(quota)))


_threads_fitted = [False]


def fit_threads_to_cpu_quota():
    """torch sizes its intra-op (OpenMP) pool by the CPUs of the MACHINE; in a container with a CPU quota (this pool's GPU boxes:
    256 CPUs visible, a quota of 16) every CPU-side tensor operation of 32 K elements or more then spins up to 128 threads, the
    cgroup's quota for the 100 ms scheduling period is spent in a few milliseconds and the kernel freezes the whole process until the
    next period -- the Python thread, HIP's submission threads, everything: measured as random 40-95 ms holes in the GPU's timeline
    whenever the host staged an upload beyond 128 KB (benchmarks/probes/upload_probe2.py: ``nr_throttled`` of the cgroup counts them).
    Called once when the library is loaded: if the pool is larger than the quota it is cut to half the quota (at most 8: what this
    package does on the host is packing and bookkeeping); BIGC_KEEP_TORCH_THREADS=1 leaves it alone."""
    if _threads_fitted[0] or os.environ.get("BIGC_KEEP_TORCH_THREADS", "0") == "1":
        return
    _threads_fitted[0] = True
    import torch
    n = cpu_quota()
    if torch.get_num_threads() > n:
        torch.set_num_threads(max(1, min(8, n // 2)))


_staging = {}


def staging(device):
    import torch
    key = str(torch.device(device))
    if key not in _staging:
        _staging[key] = _Staging()
    return _staging[key]


def upload(host, device):
    """Host tensor / ndarray -> device without stalling the host: staged in the device's pinned block (``_Staging``) and queued on
    the current stream.  A plain ``.to(device)`` from pageable memory waits for everything queued before it."""
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(host)) if isinstance(host, np.ndarray) else host.contiguous()
    if torch.device(device).type != "cuda" or os.environ.get("BIGC_UPLOAD_BLOCKING", "0") == "1":
        return t.to(device)
    # Beyond the cap the copy is done by a kernel of the library that reads the pinned block directly (no copy queue involved, the
    # host does not wait); BIGC_UPLOAD_KERNEL=0 restores the blocking copy from pageable memory that rounds 2-3 used there.  (What
    # made asynchronous uploads of that size stall for 40-95 ms then was not the copy: staging 32 K elements or more with
    # ``Tensor.copy_`` wakes torch's intra-op thread pool -- one thread per CPU the machine has -- inside a container whose CPU quota
    # is a fraction of that, and the kernel then throttles the whole process, HIP's threads included, until the next scheduling
    # period: ``fit_threads_to_cpu_quota`` below, benchmarks/probes/upload_probe2.py.)
    cap = int(os.environ.get("BIGC_UPLOAD_ASYNC_MAX", "65536"))
    if not t.numel():
        return t.to(device)
    if t.numel() * t.element_size() > cap:
        out = staging(device).put(t, device, by_kernel=True) if os.environ.get("BIGC_UPLOAD_KERNEL", "1") != "0" else None
        return out if out is not None else t.to(device)
    out = staging(device).put(t, device)
    return out if out is not None else t.to(device)


def download(t, slot="default"):
    """Device tensor -> host tensor through the device's pinned block, waiting only for the copy (and what was queued before it).
    ``t.cpu()`` into pageable memory makes the runtime pin and unpin the destination for copies beyond its small staging buffers:
    a page-table update of the GPU per call, harmless while the GPU idles, a stall of tens of ms when work is queued behind it."""
    import torch
    if not t.is_cuda:
        return t
    if os.environ.get("BIGC_DOWNLOAD_BLOCKING", "0") == "1":
        return t.cpu()
    host = staging(t.device).download_buffer(t.shape, t.dtype, slot)
    with torch.cuda.device(t.device):                 # copy and event on the current stream of the TENSOR's device, not of the caller's
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
    ev.synchronize()
    return host
