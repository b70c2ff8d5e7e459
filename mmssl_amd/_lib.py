"""ctypes binding of libmmssl_hip.so (the C ABI declared in include/mmssl_hip.h).

The library is REQUIRED: there is no eager / CPU fallback anywhere in this package. If
the shared object is missing or a symbol cannot be resolved the import of the op fails
loudly (RuntimeError) instead of silently computing something else.
"""
import ctypes
import os

# torch MUST be imported before libmmssl_hip.so is loaded: the torch wheel bundles its own HIP
# runtime under the same SONAME (libamdhip64.so.7) as /opt/rocm's. Whichever is loaded first serves
# both, and streams / device pointers are only interchangeable inside ONE runtime instance.
import torch  # noqa: F401
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMSSL_LIB: another build of the same library (tools/ decomposition builds only; tests and bench never set it)
LIB_PATH = os.environ.get("MMSSL_LIB") or os.path.join(_HERE, "libmmssl_hip.so")

_i32p = POINTER(c_int32)
_i64p = POINTER(c_int64)
_f32p = POINTER(c_float)

# name -> (restype, argtypes); mirrors include/mmssl_hip.h one to one
SIGNATURES = {
    "mmssl_abi_version": (c_int, []),
    "mmssl_strerror": (c_char_p, [c_int]),
    "mmssl_graph_create": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p,
                                   POINTER(c_void_p)]),
    "mmssl_graph_destroy": (c_int, [c_void_p]),
    "mmssl_graph_info": (c_int, [c_void_p, _i64p]),
    "mmssl_graph_export_transpose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmssl_graph_export_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, _i64p, c_void_p]),
    "mmssl_graph_pair_create": (c_int, [c_int32, c_int32, c_int64, POINTER(c_void_p)]),
    "mmssl_graph_pair_destroy": (c_int, [c_void_p]),
    "mmssl_graph_pair_get": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p)]),
    "mmssl_graph_pair_rebuild": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mmssl_csr_validate_host": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int64]),
    "mmssl_csr_transpose_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64,
                                         c_void_p, c_void_p, c_void_p]),
    "mmssl_graph_create_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int, c_void_p,
                                      POINTER(c_void_p)]),
    "mmssl_graph_create_banded": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int, c_void_p, c_void_p,
                                          c_void_p, POINTER(c_void_p)]),
    "mmssl_plan_band_host": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "mmssl_plan_band_group_items_host": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    "mmssl_plan_band_wave_blocks_host": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    "mmssl_plan_count_host": (c_int, [c_void_p, c_int32, _i64p]),
    "mmssl_plan_fill_host": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "mmssl_graph_rows_mask_normalize_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_float, c_void_p,
                                                    c_void_p]),
    "mmssl_graph_rows_mask_normalize_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                                        c_int64, c_float, c_void_p, c_void_p]),
    "mmssl_graph_rows_dense_f32": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_int64, c_void_p]),
    "mmssl_sim_rows_parts": (c_int, [c_int64]),
    "mmssl_sim_rows_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float,
                                   c_void_p, c_int64, c_void_p, c_void_p]),
    "mmssl_graph_sim_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_float, c_void_p,
                                         c_int64, c_void_p, c_void_p]),
    "mmssl_rows_scale_parts_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_float, c_void_p,
                                           c_void_p]),
    "mmssl_graph_rows_mask_normalize_bwd_ld_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                                           c_int64, c_void_p, c_int64, c_float, c_void_p, c_int64,
                                                           c_void_p]),
    "mmssl_topk_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "mmssl_rows_membership_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "mmssl_spmm_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "mmssl_spmm_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t,
                               c_void_p]),
    "mmssl_l2norm_rows_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int, c_float, c_void_p,
                                      c_void_p]),
    "mmssl_l2norm_rows_bwd_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int, c_float, c_void_p,
                                          c_void_p]),
    "mmssl_softmax_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mmssl_spmm_ld_f32": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_float,
                                  c_void_p, c_size_t, c_void_p]),
    "mmssl_softmax_rows_bwd_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int, c_void_p, c_void_p]),
    "mmssl_spmm_ex_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_float, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "mmssl_spmm_mask_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_size_t,
                                    c_void_p]),
    "mmssl_fuse_blocks": (c_int, [c_int64, c_int, c_int]),
    "mmssl_fuse_fwd_f32": (c_int, [c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_float, c_void_p, c_int, c_float,
                                   c_void_p, c_void_p, c_void_p]),
    "mmssl_fuse_bwd_f32": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_float, c_void_p,
                                   c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmssl_layer_combine_blocks": (c_int, [c_int64, c_int]),
    "mmssl_layer_combine_f32": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p, c_float, c_int64, c_int,
                                        c_float, c_void_p, c_void_p, c_void_p]),
    "mmssl_layer_combine_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_float,
                                            c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmssl_sum_partials_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mmssl_fuse_fwd_rows_f32": (c_int, [c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int,
                                        c_float, c_void_p, c_void_p]),
    "mmssl_fuse_fwd_owned_rows_f32": (c_int, [c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_float, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "mmssl_loss_add_partials_f32": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "mmssl_sumsq_workspace_bytes": (c_size_t, [c_int64]),
    "mmssl_sumsq_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_linear_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "mmssl_linear_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int,
                                 c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_transpose_mask_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "mmssl_transpose_mask_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "mmssl_proj_supported": (c_int, [c_int, c_void_p, c_int64, c_int, c_int]),
    "mmssl_proj_workspace_bytes": (c_size_t, [c_int, c_void_p, c_int64, c_int, c_int]),
    "mmssl_proj_fwd_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                   c_void_p, c_float, c_float, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "mmssl_proj_wgrad_f32": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "mmssl_proj_wgrad_adamw_f32": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                           c_float, c_float, c_float, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "mmssl_projx_supported": (c_int, [c_int, c_void_p, c_int64, c_int]),
    "mmssl_projx_image_floats": (c_size_t, [c_int64, c_int64]),
    "mmssl_projx_pack_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mmssl_projx_workspace_bytes": (c_size_t, [c_int, c_void_p, c_int64, c_int, c_int, c_int]),
    "mmssl_projx_fwd_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                    c_void_p, c_float, c_float, c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "mmssl_projx_wimg_bytes": (c_size_t, [c_int, c_void_p]),
    "mmssl_projx_wsplit_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmssl_projx_fwd_img_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                        c_void_p, c_float, c_float, c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "mmssl_projx_wgrad_f32": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                      c_int, c_void_p, c_size_t, c_void_p]),
    "mmssl_projx_wgrad_adamw_f32": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                            c_float, c_float, c_float, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mmssl_projx_wgrad_adamw_img_f32": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_float, c_float, c_float, c_float, c_float, c_int, c_void_p, c_int, c_void_p,
                                                c_size_t, c_void_p]),
    "mmssl_linear_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "mmssl_linear_wgrad_fuses_mask": (c_int, [c_int64, c_int, c_int]),
    "mmssl_adamw_sliced_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "mmssl_linear_wgrad_f32": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_ell_spmm_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "mmssl_ell_spmm_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "mmssl_mul_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mmssl_mul_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "mmssl_ngcf_combine_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_float, c_void_p, c_void_p,
                                       c_void_p]),
    "mmssl_ngcf_combine_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64,
                                           c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "mmssl_eval_workspace_bytes": (c_size_t, [c_int64]),
    "mmssl_eval_accumulate_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, POINTER(c_int), c_int,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_usim_workspace_bytes": (c_size_t, [c_int, c_int64]),
    "mmssl_usim_rows_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_graph_usim_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_float, c_void_p, c_int64,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_peer_create": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "mmssl_peer_destroy": (c_int, [c_void_p]),
    "mmssl_peer_info": (c_int, [c_void_p, _i64p]),
    "mmssl_peer_set_timeout_ms": (c_int, [c_void_p, c_int64]),
    "mmssl_peer_handle_bytes": (c_int, []),
    "mmssl_peer_flags_handle": (c_int, [c_void_p, c_void_p]),
    "mmssl_peer_open_flags": (c_int, [c_void_p, c_void_p]),
    "mmssl_peer_window_create": (c_int, [c_void_p, c_int64, POINTER(c_int), c_void_p, POINTER(c_void_p)]),
    "mmssl_peer_window_open": (c_int, [c_void_p, c_int, c_void_p]),
    "mmssl_peer_push_rows_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_int, c_int64, c_int64,
                                         c_int, c_void_p]),
    "mmssl_peer_signal_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "mmssl_peer_signal": (c_int, [c_void_p, c_int, c_void_p]),
    "mmssl_peer_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "mmssl_peer_pull_sum_rows_f32": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_int64, c_void_p, c_int64,
                                             c_void_p]),
    "mmssl_peer_sum_slots_f32": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "mmssl_peer_error": (c_int, [c_void_p, POINTER(ctypes.c_uint32)]),
    "mmssl_mask_scale_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_void_p, c_void_p]),
    "mmssl_mask_packed_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "mmssl_dropout_mask_u8": (c_int, [c_void_p, c_float, c_int64, c_void_p, c_void_p]),
    "mmssl_adamw_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                                c_float, c_float, c_float, c_void_p]),
    "mmssl_gather_owned_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "mmssl_scatter_owned_rows_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mmssl_loss_assemble_tick_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p, c_int,
                                             c_void_p, c_int, c_void_p]),
    "mmssl_adamw_ex_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                                   c_float, c_float, c_float, c_int, c_void_p]),
    "mmssl_dropout_mask_ex_u8": (c_int, [c_void_p, c_float, c_int64, c_void_p, c_int, c_void_p]),
    "mmssl_tick_u64": (c_int, [c_void_p, c_void_p]),
    "mmssl_select_slot_i64": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "mmssl_loss_assemble_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "mmssl_loss_assemble_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "mmssl_infonce_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "mmssl_infonce_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "mmssl_infonce_fwd_eps_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "mmssl_infonce_bwd_f32": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "mmssl_infonce_multi_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "mmssl_infonce_multi_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "mmssl_infonce_multi_bwd_f32": (c_int, [c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "mmssl_infonce_multi_fwd_phase_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p,
                                                  c_void_p, c_size_t, c_int, c_void_p]),
    "mmssl_infonce_multi_fwd_ticket_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p,
                                                   c_void_p, c_size_t, c_void_p, c_void_p]),
    "mmssl_infonce_multi_fwd_ticket_bpr_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float,
                                                       c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_int64, c_float, c_int64, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_infonce_multi_bwd_finish_bpr_f32": (c_int, [c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_int64, c_float, c_int64, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float,
                                                       c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t,
                                                       c_void_p, c_int64, c_void_p]),
    "mmssl_layer_combine_bwd2_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                             c_float, c_float, c_void_p, c_float, c_int, c_float, c_void_p]),
    "mmssl_bpr_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int64,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float,
                                   c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p,
                                   c_int64, c_void_p]),
    "mmssl_infonce_multi_bwd_phase_f32": (c_int, [c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "mmssl_bpr_workspace_bytes": (c_size_t, [c_int64]),
    "mmssl_bpr_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int, c_float, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mmssl_bpr_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int, c_float, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
}

_lib = None


class MmsslError(RuntimeError):
    pass


def lib():
    """The loaded library (loaded once). Raises if libmmssl_hip.so is absent: build it with
    `python -m mmssl_amd.build` (or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmsslError("%s not found — the HIP extension is required (no fallback path). "
                         "Run `python -m mmssl_amd.build`." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise MmsslError("libmmssl_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    if L.mmssl_abi_version() != 1:
        raise MmsslError("libmmssl_hip.so ABI version mismatch")
    _lib = L
    return L


def check(code, what):
    if code != 0:
        msg = lib().mmssl_strerror(code)
        raise MmsslError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))


def stream_ptr():
    """hipStream_t of torch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream
