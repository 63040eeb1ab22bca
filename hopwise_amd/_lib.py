"""ctypes binding of libhopwise_amd.so -- the C ABI declared in include/hopwise_amd.h.

There is NO fallback: if the library is missing or a symbol is absent the import raises, and every
compute wrapper in :mod:`hopwise_amd.ops` refuses CPU tensors.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhopwise_amd.so")

HW_TRANSE, HW_ROTATE, HW_COMPLEX, HW_DISTMULT, HW_RESCAL = 0, 1, 2, 3, 4
MODEL_IDS = {"TransE": HW_TRANSE, "RotatE": HW_ROTATE, "ComplEx": HW_COMPLEX, "DistMult": HW_DISTMULT,
             "RESCAL": HW_RESCAL, "TorusE": 5, "TransH": 6, "TransD": 7, "Analogy": 8, "HolE": 9}

STATUS = {0: "HW_OK", -1: "HW_ERR_INVALID_ARG", -2: "HW_ERR_UNSUPPORTED", -3: "HW_ERR_LAUNCH", -4: "HW_ERR_WORKSPACE"}

c_f32p = C.c_void_p  # device pointers travel as integers
c_i64p = C.c_void_p
c_i32p = C.c_void_p


MAX_TABLES = 3  # HW_KGE_MAX_TABLES


class KgeDesc(C.Structure):
    _fields_ = [
        ("model", C.c_int32),
        ("dim", C.c_int32),
        ("head", C.c_void_p * MAX_TABLES),
        ("rel", C.c_void_p * MAX_TABLES),
        ("tail", C.c_void_p * MAX_TABLES),
        ("n_head", C.c_int64),
        ("n_rel", C.c_int64),
        ("n_tail", C.c_int64),
        ("margin", C.c_float),
        ("eps", C.c_float),
    ]


class PgprGraph(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p),
        ("rel", C.c_void_p),
        ("dst", C.c_void_p),
        ("n_users", C.c_int32),
        ("n_entities", C.c_int32),
        ("n_items", C.c_int32),
        ("ui_relation", C.c_int32),
        ("max_degree", C.c_int32),
    ]


class CafeNode(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("parent", "rel", "eh_is_entity", "k", "is_leaf", "lp_row", "depth", "vocab", "cap", "slot_off")]


class CafeModule(C.Structure):
    """hw_cafe_module (include/hopwise_amd.h): device pointers to one relation module's evaluation-mode parameters."""

    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "bn1_weight", "bn1_bias", "bn1_mean", "bn1_var", "w2", "b2", "bn2_weight",
                                          "bn2_bias", "bn2_mean", "bn2_var", "w3", "b3")] + [
        ("bn1_eps", C.c_float), ("bn2_eps", C.c_float), ("deep", C.c_int32), ("reserved", C.c_int32)]


class KgeGrads(C.Structure):
    _fields_ = [("head", C.c_void_p * MAX_TABLES), ("rel", C.c_void_p * MAX_TABLES), ("tail", C.c_void_p * MAX_TABLES)]


ADAM_MAX_SEGS = 16


class AdamSeg(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


class AdamArgs(C.Structure):
    _fields_ = [("seg", AdamSeg * ADAM_MAX_SEGS), ("n_segs", C.c_int32)]


# name -> (restype, argtypes); kept in one table so tests can check that every symbol declared in
# include/hopwise_amd.h is exported.
SIGNATURES = {
    "hw_kge_score_grouped": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
         c_f32p, C.c_int64, C.c_int64, c_i32p, C.c_void_p],
    ),
    "hw_kge_score_grouped_sampled_selfadv": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p,
         C.c_int64, c_i64p, C.c_int64, C.c_uint64, C.c_uint64, c_f32p, C.c_int64, C.c_int64, C.c_float, c_f32p, c_f32p,
         c_i32p, c_i32p, C.c_void_p],
    ),
    "hw_kge_score_grouped_sampled_selfadv_chained": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p,
         C.c_int64, c_i64p, C.c_int64, C.c_uint64, C.c_uint64, c_f32p, C.c_int64, C.c_int64, C.c_float, c_f32p, c_f32p,
         C.c_int64, c_f32p, c_i32p, C.c_void_p],
    ),
    "hw_selfadv_finish": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "hw_kge_full_sort": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_int64,
         c_i32p, C.c_void_p],
    ),
    "hw_kge_full_sort_topk_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32]),
    "hw_kge_full_sort_topk": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p, C.c_int64,
         C.c_int32, C.c_int32, C.c_void_p, C.c_int64, c_f32p, c_i64p, C.c_void_p],
    ),
    "hw_kge_grad_grouped": (
        C.c_int,
        [C.POINTER(KgeDesc), C.POINTER(KgeGrads), c_i64p, c_i64p, C.c_int64, c_i64p, C.c_int64, C.c_int64,
         C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_int64, c_f32p, C.c_void_p],
    ),
    "hw_kge_grad_grouped_workspace_floats": (C.c_int64, [C.POINTER(KgeDesc), C.c_int64, C.c_int64, C.c_int32]),
    "hw_batchnorm1d_workspace_doubles": (C.c_int64, [C.c_int32]),
    "hw_batchnorm1d_forward": (
        C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                  C.c_void_p, C.c_int32, C.c_void_p]),
    "hw_batchnorm1d_backward": (
        C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    "hw_dropout": (C.c_int, [c_f32p, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, c_f32p, C.c_void_p]),
    "hw_batchnorm2d_workspace_doubles": (C.c_int64, [C.c_int32]),
    "hw_batchnorm2d_forward": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int32, C.c_int32, c_f32p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p,
         c_f32p, C.c_void_p, C.c_void_p],
    ),
    "hw_batchnorm2d_backward": (
        C.c_int,
        [c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
         C.c_void_p],
    ),
    "hw_adam_step": (
        C.c_int,
        [C.POINTER(AdamArgs), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_i64p, C.c_int32,
         C.c_void_p],
    ),
    "hw_loss_workspace_floats": (C.c_int64, []),
    "hw_loss_bce_logits": (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "hw_loss_bce_logits_weighted": (
        C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_float, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "hw_loss_margin_ranking": (
        C.c_int,
        [c_f32p, c_f32p, C.c_int64, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_kge_project_rows": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, c_i64p, C.c_int64, c_i64p, C.c_int64, c_f32p, C.c_void_p],
    ),
    "hw_kge_project_rows_grad": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, c_i64p, C.c_int64, c_i64p, C.c_int64, c_f32p, c_f32p, c_f32p,
         C.c_void_p],
    ),
    "hw_cafe_level_loss": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_cafe_level_loss_grad": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, c_i64p, c_i64p, c_i64p, C.c_float, C.c_float, c_f32p, C.c_int64,
         C.c_void_p],
    ),
    "hw_cafe_level_loss_grad_dev": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, c_f32p, C.c_int64,
         C.c_void_p],
    ),
    "hw_conv2d_1ch_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "hw_conv2d_1ch_forward": (
        C.c_int,
        [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_void_p],
    ),
    "hw_conv2d_1ch_backward": (
        C.c_int,
        [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p,
         c_f32p, C.c_void_p],
    ),
    "hw_tucker_contract_dropout": (
        C.c_int,
        [c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_uint64, c_f32p, C.c_void_p],
    ),
    "hw_tucker_contract_shared": (
        C.c_int, [c_f32p, c_f32p, c_i64p, c_f32p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_uint64, c_f32p,
                  C.c_void_p]),
    "hw_tucker_contract_shared_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32]),
    "hw_tucker_contract_shared_grad": (
        C.c_int, [c_f32p, c_f32p, c_i64p, c_i64p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, C.c_uint64,
                  C.c_uint64, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "hw_tucker_keep_words": (C.c_int64, [C.c_int64, C.c_int32]),
    "hw_tucker_contract_shared_keep": (
        C.c_int, [c_f32p, c_f32p, c_i64p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_uint64, c_f32p, C.c_void_p,
                  C.c_void_p]),
    "hw_tucker_contract_shared_grad_keep": (
        C.c_int, [c_f32p, c_f32p, c_i64p, c_i64p, C.c_int64, C.c_void_p, c_f32p, C.c_int64, C.c_int32, C.c_float, c_f32p,
                  c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "hw_tucker_contract_dropout_grad": (
        C.c_int,
        [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_uint64, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_kge_score_grouped_sampled": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p,
         C.c_int64, c_i64p, C.c_int64, C.c_uint64, C.c_uint64, c_f32p, C.c_int64, C.c_int64, c_i32p, C.c_void_p],
    ),
    "hw_loss_margin_ranking_sigmoid": (
        C.c_int,
        [c_f32p, c_f32p, C.c_int64, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_loss_self_adversarial": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_float, c_f32p, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_kge_full_sort_topk_if": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p, C.c_int64,
         C.c_int32, C.c_int32, C.c_void_p, C.c_int64, c_f32p, c_i64p, c_i32p, C.c_void_p],
    ),
    "hw_kge_query_rows": (
        C.c_int,
        [C.POINTER(KgeDesc), c_i64p, c_i64p, C.c_int64, C.c_int64, c_f32p, C.c_int64, c_f32p, c_i32p, C.c_void_p],
    ),
    "hw_rows_sqnorm": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int32, c_f32p, C.c_void_p]),
    "hw_kge_dist_finish": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p,
         C.c_float, C.c_float, C.c_int32, c_i32p, C.c_void_p],
    ),
    "hw_gemm_rank": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
         c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_gemm_f32_masked": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
         c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_rank_planes_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "hw_rank_planes": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "hw_gemm_rank_planes": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_int64, c_f32p,
         C.c_int64, C.c_void_p],
    ),
    "hw_mask_bits_words": (C.c_int64, [C.c_int64]),
    "hw_mask_bits": (
        C.c_int,
        [c_i64p, c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p],
    ),
    "hw_topk_from_tile_max": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i64p, c_i32p, C.c_void_p],
    ),
    "hw_gemm_dist": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_f32p, c_f32p, C.c_float,
         C.c_float, C.c_void_p],
    ),
    "hw_mask_scores_ld": (
        C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p, C.c_int64, C.c_int32, C.c_void_p]),
    "hw_topk_rerank": (
        C.c_int,
        [c_i64p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, c_f32p, C.c_float, C.c_float, c_f32p,
         c_i64p, c_i32p, C.c_void_p],
    ),
    "hw_topk_rerank_dot": (
        C.c_int,
        [c_i64p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, c_f32p, C.c_float, c_f32p, c_i64p,
         c_i32p, C.c_void_p],
    ),
    "hw_mask_scores": (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_i64p, c_i64p, C.c_int64, C.c_void_p]),
    "hw_mask_scores_ex": (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_i64p, c_i64p, C.c_int64, C.c_int32, C.c_void_p]),
    "hw_topk_rows": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, c_f32p, c_i64p, C.c_void_p]),
    "hw_topk_rows_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32]),
    "hw_topk_rows_ws": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, c_f32p, c_i64p, C.c_void_p, C.c_int64, C.c_void_p],
    ),
    "hw_neg_sample": (
        C.c_int,
        [c_i64p, c_i64p, C.c_int64, c_i64p, C.c_int64, C.c_int32, C.c_int64, c_f32p, c_i64p, c_i64p, C.c_int64,
         C.c_uint64, C.c_uint64, c_i64p, C.c_int64, C.c_int64, c_i32p, C.c_void_p],
    ),
    "hw_pgpr_init_paths": (C.c_int, [c_i64p, C.c_int64, c_i32p, c_f32p, c_i32p, C.c_void_p]),
    "hw_pgpr_rel_entity_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "hw_pgpr_user_scores": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_i64p, C.c_int64, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p],
    ),
    "hw_pgpr_user_scores_tp": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_i64p, C.c_int64, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "hw_pgpr_actions": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, C.c_void_p, C.c_void_p, c_i32p, c_i32p, C.c_int64, C.c_int32,
         C.c_int32, c_i32p, c_i32p, C.c_void_p, C.c_int64, C.c_int64, c_i32p, C.c_int64, C.c_void_p],
    ),
    "hw_pgpr_actions_workspace_ints": (C.c_int64, [C.c_int64]),
    "hw_pgpr_state": (
        C.c_int,
        [C.POINTER(PgprGraph), c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_i32p, c_i32p, C.c_int64, c_f32p,
         C.c_void_p],
    ),
    "hw_mlp_layer": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_int64, c_i32p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
         c_f32p, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p],
    ),
    "hw_gemm_f32": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int64, c_i32p, C.c_int64, C.c_int64,
         C.c_int64, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_uint64, C.c_uint64, c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_tucker_contract": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_void_p]),
    "hw_tucker_contract_grad": (
        C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, C.c_void_p]),
    "hw_tucker_bce_workspace_floats": (C.c_int64, []),
    "hw_tucker_sigmoid_bce": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_void_p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_tucker_sigmoid_bce_ld": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p,
         C.c_void_p],
    ),
    "hw_gemm_bce_workspace_floats": (C.c_int64, [C.c_int64, C.c_int64]),
    "hw_gemm_f32_bce": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
         C.c_float, C.c_float, c_f32p, c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_sigmoid": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "hw_pgpr_layer1_lookup": (
        C.c_int,
        [C.POINTER(C.c_void_p), c_f32p, C.c_int32, C.c_int32, c_i32p, c_i32p, C.c_int64, c_f32p, C.c_int32, C.c_uint64,
         C.c_uint64, c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_pgpr_policy_planes_bytes": (C.c_int64, [C.c_int32]),
    "hw_pgpr_policy_prepare": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_void_p, C.c_void_p]),
    "hw_pgpr_policy_fused": (
        C.c_int,
        [C.POINTER(C.c_void_p), c_f32p, c_f32p, C.c_int32, C.c_void_p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_i32p, c_i32p,
         C.c_int64, C.c_int32, c_f32p, c_f32p, C.c_uint64, C.c_uint64, C.c_uint64, c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_pgpr_policy_topk": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int32, c_i32p, c_i32p, C.c_int64, C.c_int32, c_i32p, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_pgpr_expand": (
        C.c_int,
        [C.POINTER(PgprGraph), c_i32p, c_f32p, c_i32p, C.c_int32, c_i32p, c_i32p, C.c_int64, c_i32p, c_f32p, C.c_int32,
         c_i32p, c_i32p, c_i32p, c_f32p, c_i32p, C.c_int64, C.c_void_p],
    ),
    "hw_pgpr_topk_expand": (
        C.c_int,
        [C.POINTER(PgprGraph), c_i32p, c_f32p, c_i32p, C.c_int32, c_i32p, c_i32p, C.c_int64, c_f32p, C.c_int64, C.c_int32,
         c_i32p, c_i32p, c_i32p, c_f32p, c_i32p, c_i32p, C.c_int64, C.c_int32, C.c_void_p],
    ),
    "hw_pgpr_materialize_leaves": (
        C.c_int,
        [C.POINTER(PgprGraph), c_i32p, c_f32p, c_i32p, c_i32p, C.c_int64, c_i32p, C.c_int64, c_i32p, c_f32p, C.c_void_p],
    ),
    "hw_pgpr_action_dropout": (
        C.c_int,
        [c_i32p, c_i32p, C.c_int64, C.c_int32, C.c_double, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p],
    ),
    "hw_pgpr_policy_sample": (
        C.c_int,
        [c_f32p, C.c_int64, C.c_int32, C.c_void_p, c_i32p, C.c_int64, c_i32p, C.c_uint64, C.c_uint64, c_f32p, c_i32p,
         c_f32p, c_f32p, c_f32p, C.c_void_p],
    ),
    "hw_pgpr_policy_grad": (
        C.c_int,
        [c_f32p, C.c_int32, c_i32p, c_f32p, C.c_float, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_pgpr_policy_grad_scaled": (
        C.c_int,
        [c_f32p, C.c_int32, c_i32p, c_f32p, C.c_float, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_void_p],
    ),
    "hw_act_dropout_backward": (
        C.c_int,
        [c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int32, c_f32p, C.c_int32, C.c_uint64, C.c_uint64, c_f32p, C.c_void_p],
    ),
    "hw_colsum_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_void_p]),
    "hw_colsum_workspace_floats": (C.c_int64, [C.c_int64, C.c_int64]),
    "hw_colsum_f32_ws": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "hw_pgpr_reward": (
        C.c_int,
        [c_i32p, C.c_int64, c_i32p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
         C.c_void_p, C.c_void_p],
    ),
    "hw_pgpr_reinforce_loss": (
        C.c_int,
        [C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_void_p, c_f32p, c_f32p,
         C.c_void_p],
    ),
    "hw_pgpr_collect": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_i64p, C.c_int64, C.c_void_p, C.c_int64,
         c_i32p, c_f32p, c_i32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, c_i32p, c_f32p, c_i32p,
         C.c_void_p, c_i32p, c_i32p, c_i32p, C.c_void_p],
    ),
    "hw_pgpr_collect_leaves": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_i64p, C.c_int64, C.c_void_p, C.c_int64,
         c_i32p, c_f32p, c_i32p, c_i32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, c_i32p, c_f32p, c_i32p,
         C.c_void_p, c_i32p, c_i32p, c_i32p, C.c_void_p],
    ),
    "hw_pgpr_collect_leaves_ex": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_i64p, C.c_int64, C.c_void_p, C.c_int64,
         c_i32p, c_f32p, c_i32p, c_i32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, c_i32p, c_f32p, c_i32p,
         C.c_void_p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_int64, C.c_void_p],
    ),
    "hw_gather_rows": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_i64p, C.c_int64, c_f32p, c_i32p, C.c_void_p]),
    "hw_scatter_add_rows": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_i64p, C.c_int64, c_f32p, C.c_void_p]),
    "hw_fill_bytes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "hw_shard_plan_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "hw_shard_plan": (
        C.c_int,
        [c_i64p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, c_i64p, C.c_int64, c_i32p, C.c_int64, c_i32p,
         C.c_int64, c_i32p, c_i32p, C.c_void_p],
    ),
    "hw_shard_gather": (
        C.c_int, [c_f32p, C.c_int64, C.c_int32, c_i32p, C.c_int64, c_i32p, C.c_int64, C.c_int32, C.c_int32, c_f32p,
                  C.c_int64, C.c_int64, c_i32p, C.c_void_p]),
    "hw_shard_scatter_add": (
        C.c_int, [c_f32p, C.c_int64, C.c_int32, c_i32p, C.c_int64, c_i32p, C.c_int64, C.c_int32, C.c_int32, c_f32p,
                  C.c_int64, C.c_int64, c_i32p, C.c_void_p]),
    "hw_log_softmax_rows": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_void_p]),
    "hw_cafe_modules_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32]),
    "hw_cafe_modules_forward": (
        C.c_int,
        [C.POINTER(CafeModule), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int32, c_f32p, C.c_int64, C.c_int32, c_i64p,
         c_i64p, c_i64p, c_f32p, c_f32p, C.c_int64, C.c_void_p],
    ),
    "hw_cafe_modules_forward_fused": (
        C.c_int,
        [C.POINTER(CafeModule), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32, c_f32p, C.c_int64,
         C.c_int32, c_i64p, c_i64p, c_i64p, c_f32p, C.c_void_p],
    ),
    "hw_cafe_collect": (
        C.c_int,
        [C.c_void_p, c_i32p, c_i64p, c_i32p, c_f32p, c_i32p, c_i32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, c_i32p,
         c_f32p, C.c_int32, c_i32p, c_f32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_void_p],
    ),
    "hw_cafe_execute": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, c_i32p, c_i64p, C.c_int64, C.c_int32, c_f32p, C.c_int64, C.c_void_p,
         C.c_int64, C.c_int32, c_i32p, c_f32p, c_i32p, c_i64p, C.c_void_p],
    ),
    "hw_cafe_execute_logits": (
        C.c_int,
        [C.POINTER(PgprGraph), C.c_void_p, c_i32p, c_i64p, C.c_int64, C.c_int32, c_f32p, C.c_int64, c_f32p, C.c_void_p,
         C.c_int64, C.c_int32, c_i32p, c_f32p, c_i32p, c_i64p, C.c_void_p],
    ),
    "hw_gemm_f32_lse": (
        C.c_int,
        [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_f32p, c_f32p,
         C.c_int64, C.c_void_p],
    ),
    "hw_lse_rows": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_void_p]),
    "hw_cafe_plan": (
        C.c_int,
        [c_i64p, C.c_int64, C.c_int32, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_int64, c_i32p, c_i32p,
         c_i32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, c_i32p, c_i64p, c_i64p, c_i64p, c_i64p,
         c_i32p, C.c_void_p],
    ),
    "hw_version": (C.c_char_p, []),
}

_lib = None


class HopwiseAmdError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach the argtypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; import it first so that this library's DT_NEEDED entry resolves
    # to the SAME runtime instance (streams / device pointers are only valid inside one runtime).
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise HopwiseAmdError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -m hopwise_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise HopwiseAmdError(f"{what} failed: {STATUS.get(rc, rc)}")
