"""ctypes binding of librevs_admm.so (include/revs_admm.h: the boundary; include/revs_admm_ops.h: the operator's
building blocks the Python driver issues one by one).

Loading fails loudly when the library has not been built: nothing in this
package computes on the CPU instead.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (REVS_LIB: a tuning build of the same sources made with `python -m revs_admm_amd.build --out`)
LIB_PATH = os.environ.get("REVS_LIB") or os.path.join(HERE, "librevs_admm.so")

# numpy mirror of revs_home_t (include/revs_admm.h)
HOME_DTYPE = np.dtype([("ev", "<i4"), ("start", "<i4"), ("end", "<i4"), ("nmin", "<i4"),
                       ("nmax", "<i4"), ("rating", "<f4"), ("capacity", "<f4"),
                       ("initial", "<f4")])
assert HOME_DTYPE.itemsize == 32

MODE_BINARY, MODE_RELAXED_PDHG, MODE_RELAXED_EXACT = 0, 1, 2
MODES = {"binary": MODE_BINARY, "relaxed": MODE_RELAXED_PDHG, "pdhg": MODE_RELAXED_PDHG,
         "relaxed_exact": MODE_RELAXED_EXACT}


class PDHG(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("check", C.c_int32), ("tol", C.c_float),
                ("tau_scale", C.c_float), ("sigma_scale", C.c_float), ("full_rows", C.c_int32),
                ("polish", C.c_int32), ("lanes", C.c_int32), ("keys64", C.c_int32)]


class PlanDesc(C.Structure):
    """revs_plan_desc_t"""
    _fields_ = [("n_homes", C.c_int64), ("m", C.c_int32), ("T", C.c_int32),
                ("node_ptr", C.c_void_p), ("R", C.c_void_p), ("Rt", C.c_void_p),
                ("kappa", C.c_double), ("vlo", C.c_double), ("vhi", C.c_double),
                ("kadd", C.c_int32), ("ksplit", C.c_int32),
                ("d_slabs", C.c_void_p), ("v_slabs", C.c_void_p), ("pnq", C.c_void_p),
                ("vfull", C.c_void_p), ("viol", C.c_void_p), ("partial", C.c_void_p),
                ("cand_idx", C.c_void_p), ("cand_cnt", C.c_void_p), ("cand_val", C.c_void_p),
                ("stats", C.c_void_p), ("stats_host", C.c_void_p),
                ("cost", C.c_void_p), ("homes", C.c_void_p), ("load", C.c_void_p),
                ("diff", C.c_void_p), ("dsq", C.c_void_p), ("status", C.c_void_p),
                ("pdhg_dual", C.c_void_p), ("mode", C.c_int32), ("pdhg", PDHG),
                ("node_of", C.c_void_p), ("recompute_pe_new", C.c_int32),
                ("cand_idx1", C.c_void_p), ("cand_cnt1", C.c_void_p), ("cand_val1", C.c_void_p),
                ("stats1", C.c_void_p), ("stats1_host", C.c_void_p),
                ("yhat", C.c_void_p), ("k_full", C.c_void_p), ("info", C.c_void_p),
                ("delta", C.c_double), ("eps", C.c_double), ("max_pivots", C.c_int32)]


class SpecState(C.Structure):
    """revs_spec_state_t"""
    _fields_ = [("p_est", C.c_void_p), ("p_est_new", C.c_void_p), ("p_est_alt", C.c_void_p),
                ("p_sch", C.c_void_p), ("p_sch_alt", C.c_void_p), ("gamma", C.c_void_p),
                ("gamma_alt", C.c_void_p), ("p0", C.c_void_p), ("p_alt", C.c_void_p),
                ("fused_p", C.c_void_p), ("fused_ready", C.c_int32)]


class ChainState(C.Structure):
    """revs_chain_state_t"""
    _fields_ = [("y", C.c_void_p), ("y_trial", C.c_void_p), ("use_y", C.c_int32), ("sup0", C.c_int32),
                ("p_est", C.c_void_p), ("p_est_new", C.c_void_p), ("p_sch", C.c_void_p),
                ("p_sch_alt", C.c_void_p), ("gamma", C.c_void_p), ("gamma_alt", C.c_void_p)]


class ChainFoldState(C.Structure):
    """revs_chain_fold_state_t"""
    _fields_ = [("y", C.c_void_p), ("y_trial", C.c_void_p), ("y_spare", C.c_void_p), ("use_y", C.c_int32),
                ("sup0", C.c_int32), ("p_est", C.c_void_p), ("p_est_new", C.c_void_p), ("p_sch", C.c_void_p),
                ("p_sch_alt", C.c_void_p), ("gamma", C.c_void_p), ("gamma_alt", C.c_void_p),
                ("s_out", C.c_void_p), ("c_out", C.c_void_p), ("resume", C.c_int32), ("pivots", C.c_int32),
                ("redone", C.c_int32), ("p_est_3", C.c_void_p), ("p_sch_3", C.c_void_p), ("gamma_3", C.c_void_p),
                ("pdhg_dual", C.c_void_p), ("pdhg_dual_new", C.c_void_p), ("pdhg_dual_3", C.c_void_p)]


class NewtonOpts(C.Structure):
    """revs_newton_opts_t"""
    _fields_ = [("k_slabs", C.c_void_p), ("nks", C.c_int32), ("alpha_host", C.c_void_p), ("alpha_dev", C.c_void_p),
                ("info_host", C.c_void_p), ("newton_max", C.c_int32), ("ls_max", C.c_int32)]


class NewtonState(C.Structure):
    """revs_newton_state_t"""
    _fields_ = [("y", C.c_void_p), ("y_trial", C.c_void_p), ("use_y", C.c_int32), ("sup", C.c_int32),
                ("p_est", C.c_void_p), ("p_sch", C.c_void_p), ("gamma", C.c_void_p), ("p_est_new", C.c_void_p),
                ("have_first", C.c_int32), ("have_pre", C.c_int32), ("chain_few_in", C.c_int32),
                ("ok", C.c_int32), ("newton", C.c_int32), ("evals", C.c_int32), ("pivots", C.c_int32),
                ("models_small", C.c_int32), ("models_general", C.c_int32), ("last_small", C.c_int32),
                ("few", C.c_int32), ("pre_kept", C.c_int32), ("cur", C.c_int32), ("nsup_sum", C.c_int32),
                ("nsup_max", C.c_int32), ("first_tag", C.c_double), ("pre_tag", C.c_double),
                ("big_needed", C.c_int32), ("_pad", C.c_int32)]


class Tree(C.Structure):
    """revs_tree_t"""
    _fields_ = [("n", C.c_int32), ("pack", C.c_void_p), ("w", C.c_void_p)]


class StreamState(C.Structure):
    """revs_stream_state_t"""
    _fields_ = [("p_est", C.c_void_p * 3), ("p_sch", C.c_void_p * 2), ("gamma", C.c_void_p * 2),
                ("p", C.c_void_p * 3), ("diff_hist", C.c_void_p)]


class StreamSets(C.Structure):
    """revs_stream_sets_t"""
    _fields_ = [("p_est", C.c_void_p * 4), ("p_sch", C.c_void_p * 4), ("gamma", C.c_void_p * 4),
                ("pdhg_dual", C.c_void_p * 4), ("p0", C.c_void_p), ("p0_out", C.c_void_p),
                ("p_est_next", C.c_void_p),
                ("diff_hist", C.c_void_p)]


# revs_host_allreduce_fn
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int32)

TREE_MAX = 16384         # REVS_TREE_MAX
TREE_SWEEP_MAX = 2048    # REVS_TREE_SWEEP_MAX
CHAIN_FOLD_MAX_M = 2048  # REVS_CHAIN_FOLD_MAX_M
STREAM_BLOCK_MAX = 256   # REVS_STREAM_BLOCK_MAX
AGENT_MAX_INNER = 32     # REVS_AGENT_MAX_INNER


class RevsError(RuntimeError):
    pass


_p = C.c_void_p
_i32, _i64, _f32, _f64 = C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol include/revs_admm.h and include/revs_admm_ops.h declare
SIGNATURES = {
    "revs_version": (C.c_char_p, []),
    "revs_last_error": (C.c_char_p, []),
    "revs_host_device_ptr": (C.c_int, [_p, C.POINTER(C.c_void_p)]),
    "revs_plan_create": (C.c_void_p, [C.POINTER(PlanDesc)]),
    "revs_plan_destroy": (None, [_p]),
    "revs_plan_chain_step": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                      _p, _p, _p, _p]),
    "revs_plan_chain_run": (C.c_int, [_p, _i32, C.POINTER(ChainState), _i32, _p, _p]),
    "revs_plan_chain_fold_run": (C.c_int, [_p, _i32, C.POINTER(ChainFoldState), _p, _p]),
    "revs_plan_spec_run": (C.c_int, [_p, _i32, _p, C.POINTER(SpecState), _f64, _f64, _p, _p, _p, _p]),
    "revs_plan_spec_step": (C.c_int, [_p, _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p,
                                      _p, _p, C.POINTER(C.c_double), _p, _p, _p]),
    "revs_op_dual_rows": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _f64, _f64, _p, _p, _p, _p, _p]),
    "revs_pdhg_defaults": (None, [C.POINTER(PDHG)]),
    "revs_residual_num_chunks": (_i32, [_i64]),
    "revs_agent_step": (C.c_int, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                  _f32, _i32, C.POINTER(PDHG), _p]),
    "revs_agent_step_out": (C.c_int, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                      _p, _p, _f32, _i32, C.POINTER(PDHG), _p]),
    "revs_agent_step_select": (C.c_int, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                         _p, _p, _p, _f32, _i32, C.POINTER(PDHG), _i32, _p, _p,
                                         _f64, _f64, _i32, _p, _p, _p, _p, _p, _p, _f64, _p, _p, _p,
                                         _i32, _p]),
    "revs_agent_max_inner": (C.c_int32, [C.c_int32, C.c_int32]),
    "revs_agent_step_multi": (C.c_int, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p,
                                        _p, _p, _f32, _i32, C.POINTER(PDHG), _p, _p, _i64, _p, _i32, _p]),
    "revs_op_dual_product_rows": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _f64, _f64, _i32, _p, _p, _p,
                                            _p, _p, _p, _p]),
    "revs_residual_finalize": (C.c_int, [_p, _p, _i64, _i32, _f32, _f32, _p, _p, _p]),
    "revs_residence_solve": (C.c_int, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "revs_gemm_tn_f64": (C.c_int, [_i32, _i32, _i32, _p, _i32, _p, _i32, _p, _i32, _i32, _p]),
    "revs_gemm_tn_f32": (C.c_int, [_i32, _i32, _i32, _p, _i32, _p, _i32, _p, _i32, _i32, _p]),
    "revs_gemm_tn_f64_split": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _i32, _p]),
    "revs_gemm_tn_f64_x2": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "revs_gemm_tn_f64_cat": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, _i32, _p]),
    "revs_voltage_f32": (C.c_int, [_i32, _i32, _p, _p, _p, _p]),
    "revs_aggregate_f64": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p]),
    "revs_aggregate_f32": (C.c_int, [_i32, _i32, _p, _p, _p, _p]),
    "revs_op_g0": (C.c_int, [_i64, _i32, _p, _p, _p, _f32, _p, _p]),
    "revs_op_init_home": (C.c_int, [_i64, _i32, _p, _p, _p]),
    "revs_op_init_node": (C.c_int, [_i32, _i32, _p, _p, _p, _f64, _f64, _p, _p, _p, _p]),
    "revs_op_home_pass": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _p, _f64, _f64, _p, _p, _p,
                                    _p]),
    "revs_op_home_pass_fused": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _f64, _f64, _p, _i32, _p,
                                          _p, _p, _p, _f64, _f64, _p, _p, _p, _p, _p]),
    "revs_op_node_w": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p]),
    "revs_op_row_scale": (C.c_int, [_i32, _i32, _p, _p, _p, _p]),
    "revs_op_node_scale": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, _f64, _p, _p, _p]),
    "revs_op_node_update": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _f64, _f64, _f64,
                                      _f64, _p, _p, _p, _p, _p, _p]),
    "revs_op_node_prep": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _f64, _i32, _p, _p, _p, _p]),
    "revs_op_nodefast_feas": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _f64, _f64, _p, _p, _p]),
    "revs_op_nodefast_scale": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _f64, _p, _p, _p]),
    "revs_op_nodefast_update": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _f64, _f64, _f64, _p, _p,
                                          _p, _p, _p]),
    "revs_op_nodefast_dualres": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _f64, _p, _p]),
    "revs_op_nodefast_finish": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "revs_op_node_apply": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _f64, _i32, _p, _p, _p]),
    "revs_op_export": (C.c_int, [_i64, _i32, _p, _p, _p]),
    "revs_op_dual_eval": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _i32, _p, _f64, _p, _p, _p]),
    "revs_op_dual_eval_rows": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _f64, _p, _p, _p]),
    "revs_op_dual_blocks": (_i32, [_i32]),
    "revs_op_dual_select": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _f64, _f64, _i32, _p, _p, _p,
                                      _p, _p, _p, _p, _f64, _p]),
    "revs_op_dual_model": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _f64, _f64, _i32, _i32, _p,
                                     _p, _p, _p, _p]),
    "revs_op_dual_evaluate": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _f64,
                                        _f64, _f64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                        _p, _p, _f64, _p, _p]),
    "revs_op_dual_model_small": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _f64, _f64, _i32, _p, _p,
                                           _p, _p]),
    "revs_op_dual_step": (C.c_int, [_i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "revs_op_dual_select_big": (C.c_int, [_i32, _i32, _p, _p, _p, _f64, _f64, _i32, _p, _p, _p, _p]),
    "revs_op_dual_model_big": (C.c_int, [_i32, _i32, _p, _p, _p, _p, _p, _f64, _f64, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    "revs_op_dual_step_big": (C.c_int, [_i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p]),
    "revs_op_dual_select_model_step": (C.c_int, [_i32, _i32, _p, _i32, _p, _f64, _f64, _i32, _p, _p, _p,
                                                _p, _p, _p, _f64, _p, _p, _f64, _f64, _i32, _p, _p, _p,
                                                _f64, _f64, _p, _p, _p]),
    "revs_op_dual_rows_tree": (C.c_int, [_i32, _i32, C.POINTER(Tree), _p, _p, _f64, _f64, _i32, _p, _p, _p, _p,
                                         _p, _p, _p, _p, _f64, _i32, _p]),
    "revs_op_dual_evaluate_tree": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, C.POINTER(Tree), _p, _i32,
                                             _f64, _f64, _f64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                             _p, _f64, _p]),
    "revs_op_dual_tree_select_model_step": (C.c_int, [_i32, _i32, C.POINTER(Tree), _p, _p, _f64, _f64, _i32, _p,
                                                      _p, _p, _p, _p, _p, _p, _f64, _p, _f64, _f64, _i32, _p, _p,
                                                      _p, _f64, _f64, _p, _p, _p]),
    "revs_newton_chain_accept": (C.c_int, [_i32, _p, _p, _f64, _f64, _i32, _i32, _i32, _p, _p]),
    "revs_tree_voltage": (C.c_int, [_i32, _i32, C.POINTER(Tree), _p, _f64, _f64, _p, _p, _p]),
    "revs_comm_unique_id": (C.c_int, [_p]),
    "revs_comm_create": (C.c_void_p, [_p, _i32, _i32]),
    "revs_comm_create_hook": (C.c_void_p, [_p, _p, _i32, _i32]),
    "revs_comm_destroy": (None, [_p]),
    "revs_comm_allreduce_f64": (C.c_int, [_p, _p, _i64, _i32, _p]),
    "revs_plan_set_tree": (C.c_int, [_p, C.POINTER(Tree)]),
    "revs_plan_set_comm": (C.c_int, [_p, _p]),
    "revs_plan_set_stream_block": (C.c_int, [_p, C.c_int32, C.c_int32]),
    "revs_plan_set_stream_inner": (C.c_int, [_p, C.c_int32]),
    "revs_plan_set_fold_redo": (C.c_int, [_p, C.c_int32]),
    "revs_plan_set_kadd_cold": (C.c_int, [_p, C.c_int32, C.c_int32]),
    "revs_plan_prepare": (C.c_int, [_p]),
    "revs_status_or": (C.c_int, [C.c_int64, _p, _p, _p]),
    "revs_plan_set_newton": (C.c_int, [_p, C.POINTER(NewtonOpts)]),
    "revs_plan_newton_solve": (C.c_int, [_p, C.POINTER(NewtonState), _p]),
    "revs_plan_set_pdhg_dual": (C.c_int, [_p, _p]),
    "revs_plan_stream_run_blocks": (C.c_int, [_p, _i32, C.POINTER(StreamSets), _f64, _f64, _p, _p, _p, _p]),
    "revs_plan_stream_timing": (C.c_int, [_p, C.c_int32]),
    "revs_plan_stream_elapsed_ms": (C.c_int, [_p, _p]),
    "revs_plan_stream_launches": (_i64, [_p]),
    "revs_plan_collective_ms": (C.c_int, [_p, _p, _p, _p]),
    "revs_plan_stream_run": (C.c_int, [_p, _i32, C.POINTER(StreamState), _f64, _f64, _p, _p, _p]),
    "revs_plan_status_flags": (_i32, [_p, _i32]),
    "revs_op_dual_step_pending": (C.c_int, [_i32, _p, _p, _p, _p, _p, _f64, _f64, _p, _i32, _p, _p, _p]),
}
DUAL_AMAX = 128          # REVS_DUAL_AMAX
DUAL_FEW = 48            # REVS_DUAL_FEW (16 / 32: the same times on the 121144 feeder; 80 / 128: 11.2 ms against 9.5, r05)
DUAL_AMAX_BIG = 512      # REVS_DUAL_AMAX_BIG

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RevsError(
            f"{LIB_PATH} is missing: build it with `python -m revs_admm_amd.build` "
            "(hipcc, gfx950).  revs_admm_amd has no CPU fallback.")
    # PyTorch-ROCm carries its own libamdhip64; load it FIRST so that this library's
    # NEEDED libamdhip64.so.7 resolves to the runtime torch's tensors and streams live
    # in.  (Loaded the other way round the process ends up with two HIP runtimes and
    # the second one finds "no ROCm-capable device".)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().revs_last_error().decode()
        raise RevsError(f"{what or 'librevs_admm'} failed ({rc}): {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
