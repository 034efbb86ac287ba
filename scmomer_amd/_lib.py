"""ctypes binding of libscmomer_hip.so (C ABI declared in include/scmomer_hip.h).

The library is the only compute path of this package: there is no CPU or eager fallback.
If the shared object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCMOMER_HIP_LIB: another build of the same library (diagnostic builds with in-kernel stamps, experiments)
LIB_PATH = os.environ.get("SCMOMER_HIP_LIB") or os.path.join(_HERE, "libscmomer_hip.so")


class View4(C.Structure):
    """scm_view4: bf16 [B,H,N,64] view, strides in elements."""
    _fields_ = [("p", C.c_void_p), ("sb", C.c_int64), ("sh", C.c_int64), ("sn", C.c_int64)]


class FavorPlan(C.Structure):
    """scm_favor_plan"""
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("N", C.c_int), ("kmax_chunks", C.c_int),
                ("ctx_chunks", C.c_int), ("apply_chunks", C.c_int), ("group_cells", C.c_int)]


vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64


class ReduceJob(C.Structure):
    """scm_reduce_job"""
    _fields_ = [("kind", i32), ("accumulate", i32), ("src", vp), ("S", i32), ("P", i32), ("Q", i32), ("trans", i32),
                ("out", vp * 4), ("lo", i32 * 4), ("hi", i32 * 4)]


class WgradItem(C.Structure):
    """scm_wgrad_item"""
    _fields_ = [("Pm", vp), ("ldp", i64), ("Qm", vp), ("ldq", i64), ("P", i32), ("Q", i32), ("with_bias", i32), ("ws", vp)]


class AdamItem(C.Structure):
    """scm_adam_item"""
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i32), ("tensor", i32)]


class CellTailArgs(C.Structure):
    """scm_cell_tail_args"""
    _fields_ = [("P", i32), ("Hd", i32), ("O", i32), ("normalize", i32), ("Wp", vp), ("ln_p_w", vp), ("ln_p_b", vp),
                ("W1", vp), ("b1", vp), ("ln_h_w", vp), ("ln_h_b", vp), ("W2", vp), ("b2", vp), ("Wo", vp), ("bo", vp)]


class DistillArgs(C.Structure):
    """scm_distill_args"""
    _fields_ = [("tail", CellTailArgs), ("B", i32), ("G", i32), ("Na", i32), ("pos_weight_n", i32), ("distill_weight", C.c_float),
                ("group_cells", i32), ("Wa", vp), ("ba", vp),
                ("Wr", vp), ("br", vp), ("genes", vp), ("x", vp), ("atac_raw", vp), ("latent", vp), ("drop", vp),
                ("pos_weight", vp)]


class DistillGrads(C.Structure):
    """scm_distill_grads"""
    _fields_ = [(n, vp) for n in ("dW1", "db1", "dln_w", "dln_b", "dW2", "db2", "dWo", "dbo", "dWa", "dba", "dWr", "dbr")]


class FfnFwdArgs(C.Structure):
    """scm_ffn_fwd_args"""
    _fields_ = [("M", i64), ("D", i32), ("F", i32), ("x", vp), ("ldx", i64), ("gamma", vp), ("beta", vp),
                ("w1", vp), ("ldw1", i64), ("b1", vp), ("w2c", vp), ("ldw2c", i64), ("b2", vp),
                ("h", vp), ("ldh", i64), ("mean", vp), ("rstd", vp), ("pre", vp), ("ldpre", i64), ("act", vp),
                ("ldact", i64), ("y", vp), ("ldy", i64), ("gamma_next", vp), ("beta_next", vp), ("h_next", vp),
                ("ldh_next", i64), ("mean_next", vp), ("rstd_next", vp)]


# name -> (restype, argtypes); must list every symbol of include/scmomer_hip.h
SIGNATURES = {
    "scm_version": (i32, []),
    "scm_last_error": (C.c_char_p, []),
    "scm_gemm_general_launches": (i64, []),
    "scm_cast_weight": (i32, [vp, i32, i32, vp, vp, vp]),
    "scm_cast_weights_batch": (i32, [vp, i32, vp]),
    "scm_prep_omega": (i32, [vp, vp, vp]),
    "scm_embed_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "scm_embed_bwd_ws_floats": (i64, [i64, i32, i32]),
    "scm_embed_bwd": (i32, [vp, vp, i64, i32, i32, vp, vp, i32, vp]),
    "scm_ln_fwd": (i32, [vp, vp, vp, i64, i32, vp, vp, vp, vp]),
    "scm_ln_bwd_ws_floats": (i64, [i64, i32]),
    "scm_ln_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "scm_gemm_ln_bwd_supported": (i32, [i64, i32, i32, i64, i64]),
    "scm_gemm_res_ln_supported": (i32, [i64, i32, i32, i64, i64]),
    "scm_gemm_res_ln": (i32, [vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "scm_gemm_ln_bwd_ws_floats": (i64, [i64, i32]),
    "scm_gemm_ln_bwd": (i32, [vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "scm_ln_bwd_noalloc": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, i32, vp]),
    "scm_gemm_nt": (i32, [i32, i32, vp, i64, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp]),
    "scm_reduce_batch": (i32, [C.POINTER(ReduceJob), i32, vp]),
    "scm_ffn_chain_supported": (i32, [i64, i32, i32]),
    "scm_ffn_chain_fwd": (i32, [C.POINTER(FfnFwdArgs), vp]),
    "scm_wgrad_ws_floats": (i64, [i32, i32]),
    "scm_wgrad_multi": (i32, [C.POINTER(WgradItem), i32, i64, i32, vp]),
    "scm_wgrad": (i32, [vp, i64, vp, i64, i32, i64, i32, i32, i32, vp, vp, vp, i32, i32, vp]),
    "scm_favor_make_plan": (i32, [i32, i32, i32, C.POINTER(FavorPlan)]),
    "scm_favor_ws_floats": (i64, [C.POINTER(FavorPlan)]),
    "scm_favor_fwd": (i32, [C.POINTER(FavorPlan), View4, View4, View4, vp, View4, View4, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "scm_favor_bwd": (i32, [C.POINTER(FavorPlan), View4, View4, View4, View4, View4, vp, vp, vp, vp, vp, vp, vp, vp,
                            View4, View4, View4, vp, vp, vp, vp, vp, vp]),
    "scm_feature_map_ws_floats": (i64, [i64, i32]),
    "scm_feature_map": (i32, [vp, vp, i64, i32, i32, i32, i32, C.c_float, vp, vp, vp]),
    "scm_linear_attention_ws_floats": (i64, [i64, i32, i32]),
    "scm_linear_attention": (i32, [vp, vp, vp, i64, i32, i32, vp, vp, vp]),
    "scm_trace_kernel_count": (i32, []),
    "scm_trace_kernel_name": (C.c_char_p, [i32]),
    "scm_trace_enable": (i32, [i32]),
    "scm_trace_collect": (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_int64)]),
    "scm_s1_mask": (i32, [vp, vp, vp, i32, i32, C.c_float, C.c_float, i32, i64, i64, vp, vp, vp]),
    "scm_ce_ws_floats": (i64, [i64, i64]),
    "scm_ce_groups": (i64, [i64, i64]),
    "scm_cell_tail_ws_floats": (i64, [i32, i32, i32, i32]),
    "scm_cell_tail_fwd": (i32, [vp, vp, i32, i32, C.POINTER(CellTailArgs), vp, vp, vp]),
    "scm_distill_max_cells": (i32, []),
    "scm_distill_tape_floats": (i64, [i32, i32, i32, i32]),
    "scm_distill_ws_floats": (i64, [i32, i32, i32, i32, i32, i32]),
    "scm_distill_fwd": (i32, [C.POINTER(DistillArgs), vp, vp, vp, vp]),
    "scm_distill_bwd": (i32, [C.POINTER(DistillArgs), vp, vp, vp, C.POINTER(DistillGrads), i32, vp]),
    "scm_adam_chunk_elems": (i32, []),
    "scm_adam_step": (i32, [vp, i32, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64, vp, vp]),
    "scm_s1_accuracy_ws_ints": (i64, [i32, i32]),
    "scm_s1_accuracy": (i32, [vp, vp, i32, i32, i32, i64, vp, vp, vp, vp]),
    "scm_bin_max_genes": (i32, []),
    "scm_bin_csr": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, vp]),
    "scm_bin_dense": (i32, [vp, i64, i32, i32, vp, vp]),
    "scm_orthogonal_omega": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "scm_ce_fwd": (i32, [vp, vp, i64, i32, i64, i64, vp, vp, vp]),
    "scm_ce_bwd": (i32, [vp, vp, i64, i32, i64, i64, vp, vp, vp, vp]),
    "scm_head_fwd": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp]),
    "scm_head_bwd_ws_floats": (i64, [i64, i32, i32]),
    "scm_head_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C scmomer_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class ScmError(RuntimeError):
    pass


def check(code: int, what: str):
    if code != 0:
        msg = lib.scm_last_error().decode("utf-8", "replace")
        raise ScmError(f"{what} failed ({code}): {msg}")
