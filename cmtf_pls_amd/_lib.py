"""ctypes binding of libcmtfpls.so (the C ABI declared in include/cmtfpls.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CMTFPLS_LIB overrides the path (used only by tools/tune_sweeps.sh to A/B kernel variants)
LIB_PATH = os.environ.get("CMTFPLS_LIB") or os.path.join(_HERE, "lib", "libcmtfpls.so")

_P = c_void_p

class XcovBlock(ctypes.Structure):
    """cmtfpls_xcov_block (include/cmtfpls.h): one block of cmtfpls_xcov_iterate_blocks_f64."""
    _fields_ = [("S", _P), ("S2", _P), ("colcnt", _P), ("n_samples", c_double), ("order", c_int), ("A", c_int), ("B", c_int),
                ("n_squarings", c_int), ("Z", _P), ("wA", _P), ("wB", _P), ("info", _P)]


# name -> (restype, argtypes); mirrors include/cmtfpls.h one to one
SIGNATURES = {
    "cmtfpls_xcov_iterate_blocks_f64": (c_int, [ctypes.POINTER(XcovBlock), c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_abi_version": (c_int, []),
    "cmtfpls_last_error": (c_char_p, []),
    "cmtfpls_clear_error": (c_int, []),
    "cmtfpls_status_to_host": (c_int, [_P, _P, c_size_t, _P, _P]),
    "cmtfpls_colstats_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "cmtfpls_colstats_f32": (c_int, [_P, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_colstats_f64": (c_int, [_P, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_sweep_partials": (c_int, []),
    "cmtfpls_center_f32": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P]),
    "cmtfpls_center_f64": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P]),
    "cmtfpls_mode0_contract_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "cmtfpls_mode0_contract_f32": (c_int, [_P, c_int64, c_int64, _P, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_mode0_contract_f64": (c_int, [_P, c_int64, c_int64, _P, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_mode0_contract_yq_f32": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_mode0_contract_yq_f64": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_colscale_f64": (c_int, [_P, c_int64, _P, c_double, _P]),
    "cmtfpls_rank1_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cmtfpls_rank1_score_f64": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    "cmtfpls_rank1_f64": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_rank1_launches_f64": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_rank1_chain_enable": (None, [c_int]),
    "cmtfpls_rank1_chain_enabled": (c_int, []),
    "cmtfpls_normalize_f64": (c_int, [_P, c_int64, _P, _P]),
    "cmtfpls_rank1_tensor_workspace_bytes": (c_size_t, [_P, c_int]),
    "cmtfpls_rank1_tensor_f64": (c_int, [_P, _P, c_int, c_double, _P, c_int, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_kron_f64": (c_int, [_P, c_int, _P, c_int, _P, _P]),
    "cmtfpls_xcov_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "cmtfpls_xcov_f32": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_xcov_f64": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_quadform_f64": (c_int, [_P, c_int, _P, _P, _P, _P]),
    "cmtfpls_xcov_iterate_f64": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, _P, c_size_t, _P, c_size_t, _P]),
    "cmtfpls_s_downdate_f64": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "cmtfpls_project_rows_f32": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "cmtfpls_project_rows_f64": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "cmtfpls_project_rows2_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int64, c_int, _P, c_int, _P]),
    "cmtfpls_project_rows2_f64": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int64, c_int, _P, c_int, _P]),
    "cmtfpls_project_rows_idx_f32": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "cmtfpls_project_rows_idx_f64": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "cmtfpls_project_rows2_idx_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, _P, c_int, _P]),
    "cmtfpls_project_rows2_idx_f64": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, _P, c_int, _P]),
    "cmtfpls_allreduce_sum_f64": (c_int, [_P, _P, c_size_t, _P]),
    "cmtfpls_allreduce_sum_f32": (c_int, [_P, _P, c_size_t, _P]),
    "cmtfpls_axpy_scalar_f64": (c_int, [_P, c_int64, _P, _P, _P]),
    "cmtfpls_xcov_deflate_f32": (c_int, [_P, c_int64, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_xcov_deflate_f64": (c_int, [_P, c_int64, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_xcov_stats_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "cmtfpls_xcov_stats_f32": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_xcov_stats_f64": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_xcov_ssq_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "cmtfpls_xcov_ssq_f32": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_xcov_ssq_f64": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_score_contract_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "cmtfpls_score_contract_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_double, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_score_contract_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_double, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_kr_axpy_f64": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, _P, _P]),
    "cmtfpls_xcov_f32_mixed": (c_int, [_P, c_int64, c_int64, _P, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_mttkrp_f32_mixed": (c_int, [_P, c_int64, c_int, c_int, _P, _P, c_int, _P, c_int, _P]),
    "cmtfpls_mttkrp_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, c_int, _P, c_int, _P]),
    "cmtfpls_mttkrp_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, c_int, _P, c_int, _P]),
    "cmtfpls_score_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
    "cmtfpls_score_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
    "cmtfpls_score_s_f64": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "cmtfpls_deflate_contract_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "cmtfpls_deflate_contract_yq_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, c_int, _P, _P, c_size_t, _P]),
    "cmtfpls_deflate_contract_yq_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, c_int, _P, _P, c_size_t, _P]),
    "cmtfpls_score_gram_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "cmtfpls_score_gram_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "cmtfpls_q_update_f64": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "cmtfpls_deflate_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
    "cmtfpls_deflate_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
    "cmtfpls_score_deflate_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "cmtfpls_score_deflate_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "cmtfpls_small_workspace_bytes": (c_size_t, []),
    "cmtfpls_gram_tn_f64": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int64, _P, _P, c_size_t, _P]),
    "cmtfpls_rowdot_f64": (c_int, [_P, c_int, c_int, c_int64, _P, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_scores_mean_f64": (c_int, [_P, c_int, c_int64, _P, _P]),
    "cmtfpls_y_deflate_f64": (c_int, [_P, c_int, c_int, c_int64, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_sum_f64": (c_int, [_P, c_int64, _P, _P]),
    "cmtfpls_normal_solve_f64": (c_int, [_P, _P, c_int, _P, c_int, _P]),
    "cmtfpls_normal_solve_workspace_bytes": (c_size_t, [c_int]),
    "cmtfpls_normal_solve_ws_f64": (c_int, [_P, _P, c_int, _P, c_int, _P, c_size_t, _P]),
    "cmtfpls_unit_upper_solve_rows_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "cmtfpls_kr_gram_row_f64": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    "cmtfpls_kr_gram_f64": (c_int, [_P, c_int, c_int, _P, c_int, c_double, _P]),
    "cmtfpls_khatri_rao_f64": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P]),
    "cmtfpls_predict_rows_f64": (c_int, [_P, c_int64, c_int, c_int, _P, c_int, _P, _P, c_int, _P]),
    "cmtfpls_recon_f32": (c_int, [_P, c_int64, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P]),
    "cmtfpls_recon_f64": (c_int, [_P, c_int64, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P]),
    "cmtfpls_recon_r2_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "cmtfpls_recon_r2_f32": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_recon_r2_f64": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_loo_fold_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cmtfpls_loo_tpls_f64": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_loo_xcov_fold_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cmtfpls_loo_xcov_f64": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "cmtfpls_fit_small_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cmtfpls_fit_small_f64": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_double, c_int] + [_P] * 11 + [_P, c_size_t, _P]),
    "cmtfpls_add_noise_f32": (c_int, [_P, c_int64, c_double, c_uint64, c_uint64, c_double, _P]),
    "cmtfpls_add_noise_f64": (c_int, [_P, c_int64, c_double, c_uint64, c_uint64, c_double, _P]),
    "cmtfpls_ceiling_max_blocks": (c_int, []),
    "cmtfpls_ceiling_read": (c_int, [_P, c_size_t, c_int64, c_int, _P, c_int, _P]),
    "cmtfpls_ceiling_rmw": (c_int, [_P, c_size_t, c_int64, c_int, c_int, _P]),
    "cmtfpls_ceiling_copy": (c_int, [_P, _P, c_size_t, c_int64, c_int, c_int, _P]),
}

_lib = None


class CmtfplsError(RuntimeError):
    pass


def load():
    """Load libcmtfpls.so (built by ``__graft_entry__.build()`` / ``csrc/build.sh``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CmtfplsError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "cmtf_pls_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().cmtfpls_last_error()
        raise CmtfplsError(f"{what or 'cmtfpls call'} failed with status {rc}: {msg.decode() if msg else ''}")
