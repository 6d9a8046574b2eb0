"""ctypes binding of libqcnet_hip.so (the C ABI declared in include/qcnet_hip.h).

The library is the product: there is no CPU or eager fallback.  If the shared object is missing the
import of any op raises immediately with the build command.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int64, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# QCN_HIP_LIB: load another build of the same library instead (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get('QCN_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libqcnet_hip.so')

_lib = None


class QcnError(RuntimeError):
    pass


def _declare(lib) -> None:
    P = c_void_p
    lib.qcn_version.restype = c_int
    lib.qcn_version.argtypes = []
    lib.qcn_status_string.restype = ctypes.c_char_p
    lib.qcn_status_string.argtypes = [c_int]
    radius_common = [P, P, P, P, c_int64, c_int64, c_int, c_float, c_int, c_int, P, P]
    lib.qcn_radius_count.restype = c_int
    lib.qcn_radius_count.argtypes = radius_common + [P, P]
    lib.qcn_radius_fill.restype = c_int
    lib.qcn_radius_fill.argtypes = radius_common + [P, P, P, P]
    lib.qcn_scan_workspace_bytes.restype = c_size_t
    lib.qcn_scan_workspace_bytes.argtypes = [c_int64]
    lib.qcn_exclusive_scan_i32.restype = c_int
    lib.qcn_exclusive_scan_i32.argtypes = [P, c_int64, P, P, c_size_t, P]
    lib.qcn_csr_workspace_bytes.restype = c_size_t
    lib.qcn_csr_workspace_bytes.argtypes = [c_int64, c_int64]
    lib.qcn_csr_build.restype = c_int
    lib.qcn_csr_build.argtypes = [P, P, c_int64, c_int64, P, P, P, P, c_size_t, P]
    lib.qcn_csr_static_max_edges.restype = c_int64
    lib.qcn_csr_static_max_edges.argtypes = []
    lib.qcn_csr_build_static.restype = c_int
    lib.qcn_csr_build_static.argtypes = [P, P, c_int64, c_int64, P, P, P, P]
    lib.qcn_attn_fwd_f32.restype = c_int
    lib.qcn_attn_fwd_f32.argtypes = [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, P, P]
    lib.qcn_attn_fused_f32.restype = c_int
    lib.qcn_attn_fused_f32.argtypes = [P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                       c_int64, P, P, P, P]
    lib.qcn_split_rows_f16.restype = c_int
    lib.qcn_split_rows_f16.argtypes = [P, c_int64, c_int, P, P]
    lib.qcn_attn_modes_pack_bytes.restype = c_size_t
    lib.qcn_attn_modes_pack_bytes.argtypes = []
    lib.qcn_attn_modes_f32.restype = c_int
    lib.qcn_attn_modes_f32.argtypes = [P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int,
                                       c_int, c_int64, P, P]
    lib.qcn_fourier_pack_floats.restype = c_size_t
    lib.qcn_fourier_pack_floats.argtypes = [c_int]
    lib.qcn_fourier_embed_f32.restype = c_int
    lib.qcn_fourier_embed_f32.argtypes = [P, c_int, P, c_int64, P, c_int64, c_int, P, c_int, P]
    lib.qcn_ln_linear_f32.restype = c_int
    lib.qcn_ln_linear_f32.argtypes = [P, c_int64, P, P, P, P, c_int, P, P, P, P, P, c_int, P]
    lib.qcn_attn_post_pack_floats.restype = c_size_t
    lib.qcn_attn_post_pack_floats.argtypes = []
    lib.qcn_attn_post_f32.restype = c_int
    lib.qcn_attn_post_f32.argtypes = [P, P, P, P, P, c_int64, P, c_int, P]
    lib.qcn_attn_post_chain_pack_floats.restype = c_size_t
    lib.qcn_attn_post_chain_pack_floats.argtypes = []
    lib.qcn_attn_post_chain_f32.restype = c_int
    lib.qcn_attn_post_chain_f32.argtypes = [P, P, P, c_int64, P, P, P, P, P, c_float, c_int, P, P, P, c_int, P]
    lib.qcn_gru_f32.restype = c_int
    lib.qcn_gru_f32.argtypes = [P, P, P, P, P, P, c_int64, c_int, P, c_int, P]
    lib.qcn_row_normalize_f32.restype = c_int
    lib.qcn_row_normalize_f32.argtypes = [P, P, c_int64, c_float, P]
    lib.qcn_mlp_heads_pack_floats.restype = c_size_t
    lib.qcn_mlp_heads_pack_floats.argtypes = [c_int]
    lib.qcn_mlp_heads_f32.restype = c_int
    lib.qcn_mlp_heads_f32.argtypes = [P, c_int64, P, c_int, POINTER(c_int), P, P, P, c_int, P]
    lib.qcn_proposal_cumsum_f32.restype = c_int
    lib.qcn_proposal_cumsum_f32.argtypes = [P, c_int, c_int64, c_int, c_int, c_int, c_float, c_int, P, P]
    lib.qcn_rel_geometry_f32.restype = c_int
    lib.qcn_rel_geometry_f32.argtypes = [P, c_int, P, c_int, P, P, P, P, P, c_int64, P, P]
    lib.qcn_fourier_linear_f32.restype = c_int
    lib.qcn_fourier_linear_f32.argtypes = [P, c_int, P, c_int64, P, c_int64, P, P, c_int, P, P, P, c_int, P]
    lib.qcn_fourier_folded_linear_f32.restype = c_int
    lib.qcn_fourier_folded_linear_f32.argtypes = [P, c_int, P, c_int64, P, c_int64, P, P, c_int, P, P, P, c_int, P]
    lib.qcn_gru_pack_floats.restype = c_size_t
    lib.qcn_gru_pack_floats.argtypes = []
    lib.qcn_attn_modes_bf16_pack_bytes.restype = c_size_t
    lib.qcn_attn_modes_bf16_pack_bytes.argtypes = []
    lib.qcn_attn_modes_bf16.restype = c_int
    lib.qcn_attn_modes_bf16.argtypes = list(lib.qcn_attn_modes_f32.argtypes)
    for name in ('qcn_attn_modes_store_f32', 'qcn_attn_modes_store_bf16'):      # ..., out, kk_out, vv_out, stream
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = list(lib.qcn_attn_modes_f32.argtypes[:-1]) + [P, P, c_int64, P]
    lib.qcn_attn_modes_cached_f32.restype = c_int
    lib.qcn_attn_modes_cached_f32.argtypes = [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int, c_int, P, P]
    lib.qcn_attn_modes_table_slots.restype = c_int64
    lib.qcn_attn_modes_table_slots.argtypes = [c_int64, c_int64]
    lib.qcn_gru_blocked_rows.restype = c_int64
    lib.qcn_gru_blocked_rows.argtypes = [c_int64, c_int]
    lib.qcn_fourier_gru_input_f32.restype = c_int
    lib.qcn_fourier_gru_input_f32.argtypes = [P, c_int, P, c_int64, c_int, P, P, c_int, P, P, P, c_int, P]
    lib.qcn_gru_blocked_f32.restype = c_int
    lib.qcn_gru_blocked_f32.argtypes = [P, P, P, P, P, P, c_int64, c_int, P, P]
    lib.qcn_stream_copy.restype = c_int
    lib.qcn_stream_copy.argtypes = [P, P, c_size_t, P]
    lib.qcn_motion_metrics_f32.restype = c_int
    lib.qcn_motion_metrics_f32.argtypes = [P, c_int, P, P, c_int, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_float, P, P]
    lib.qcn_to_agent_frame_f32.restype = c_int
    lib.qcn_to_agent_frame_f32.argtypes = [P, P, c_int64, c_int, c_int, c_int, c_int, c_int, P, P]
    lib.qcn_to_world_frame_f32.restype = c_int
    lib.qcn_to_world_frame_f32.argtypes = [P, c_int, P, P, P, c_int64, c_int, c_int, c_int, c_int, P, P]
    lib.qcn_mask_pairs_count.restype = c_int
    lib.qcn_mask_pairs_count.argtypes = [P, P, c_int64, c_int, c_int, c_int, c_int, P, P]
    lib.qcn_mask_pairs_fill.restype = c_int
    lib.qcn_mask_pairs_fill.argtypes = [P, P, c_int64, c_int, c_int, c_int, c_int, P, P, P, P]
    lib.qcn_coalesce_workspace_bytes.restype = c_size_t
    lib.qcn_coalesce_workspace_bytes.argtypes = [c_int64]
    lib.qcn_coalesce_count.restype = c_int
    lib.qcn_coalesce_count.argtypes = [P, P, P, c_int64, c_int64, P, c_size_t, P, P]
    lib.qcn_coalesce_fill.restype = c_int
    lib.qcn_coalesce_fill.argtypes = [P, P, c_int64, c_int64, P, P, P, P]
    lib.qcn_nms_select_f32.restype = c_int
    lib.qcn_nms_select_f32.argtypes = [P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P, P, P]


EXPORTED_SYMBOLS = (
    'qcn_version', 'qcn_status_string', 'qcn_radius_count', 'qcn_radius_fill', 'qcn_scan_workspace_bytes',
    'qcn_exclusive_scan_i32', 'qcn_csr_workspace_bytes', 'qcn_csr_build', 'qcn_attn_fwd_f32',
    'qcn_attn_fused_f32', 'qcn_row_normalize_f32', 'qcn_fourier_pack_floats', 'qcn_fourier_embed_f32', 'qcn_ln_linear_f32', 'qcn_attn_post_pack_floats', 'qcn_attn_post_f32', 'qcn_attn_post_chain_pack_floats', 'qcn_attn_post_chain_f32', 'qcn_gru_f32', 'qcn_attn_modes_bf16_pack_bytes', 'qcn_attn_modes_bf16', 'qcn_gru_pack_floats', 'qcn_fourier_linear_f32', 'qcn_fourier_folded_linear_f32', 'qcn_rel_geometry_f32', 'qcn_proposal_cumsum_f32', 'qcn_mlp_heads_pack_floats', 'qcn_mlp_heads_f32',
    'qcn_split_rows_f16', 'qcn_attn_modes_pack_bytes', 'qcn_attn_modes_f32',
    'qcn_attn_modes_store_f32', 'qcn_attn_modes_store_bf16', 'qcn_attn_modes_cached_f32', 'qcn_attn_modes_table_slots', 'qcn_stream_copy', 'qcn_gru_blocked_rows', 'qcn_fourier_gru_input_f32', 'qcn_gru_blocked_f32',
    'qcn_motion_metrics_f32', 'qcn_to_agent_frame_f32', 'qcn_to_world_frame_f32', 'qcn_nms_select_f32',
    'qcn_csr_static_max_edges', 'qcn_csr_build_static',
    'qcn_mask_pairs_count', 'qcn_mask_pairs_fill', 'qcn_coalesce_workspace_bytes', 'qcn_coalesce_count', 'qcn_coalesce_fill',
)


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QcnError('libqcnet_hip.so not found at %s — build it with `make -C %s` '
                           '(or `python -c "import __graft_entry__ as g; g.build()"`); there is no fallback path.'
                           % (LIB_PATH, os.path.dirname(LIB_PATH)))
        handle = ctypes.CDLL(LIB_PATH)
        _declare(handle)
        _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise QcnError('%s failed: %s (status %d)' % (what, lib().qcn_status_string(status).decode(), status))
