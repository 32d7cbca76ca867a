"""ctypes binding of libwxhip.so - the only way the Python host reaches the HIP kernels.

There is NO fallback: if the library is missing or a symbol is absent, importing the
compute entry points raises.  (Signatures mirror include/wxhip.h one to one.)
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_size_t, c_void_p

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WXHIP_LIB") or os.path.join(PKG, "lib", "libwxhip.so")  # WXHIP_LIB: A/B variants

WX_OK = 0
WX_ERR_COMM = 5
WX_COMM_ID_BYTES = 128
WX_RESERVE_STAGE, WX_RESERVE_JVP = 1, 2
WX_F64, WX_C128, WX_DUAL128 = 0, 1, 2
WX_REGION_ALL, WX_REGION_INTERIOR, WX_REGION_BOUNDARY = 0, 1, 2
WX_REDUCE_SUM, WX_REDUCE_MAX, WX_REDUCE_MIN = 0, 1, 2
WX_KERNEL_RHS, WX_KERNEL_STAGE, WX_KERNEL_JVP, WX_KERNEL_BATCH_RHS, WX_KERNEL_BATCH_JVP = 0, 1, 2, 3, 4


class WxError(RuntimeError):
    pass


class DfrOps(ctypes.Structure):
    _fields_ = [(k, POINTER(c_double)) for k in ("extrap_neg", "extrap_pos", "diff_solpt", "correction", "highfilter")]


EULER3D_METRIC_FIELDS = (
    "sqrtG", "h_contra", "christoffel", "inv_dzdeta",
    "sqrtG_itf_i", "sqrtG_itf_j", "sqrtG_itf_k",
    "h_contra_itf_i", "h_contra_itf_j", "h_contra_itf_k",
    "damp_coef", "damp_uref", "boundary_sn", "boundary_we",
)


class Euler3DMetric(ctypes.Structure):
    _fields_ = [(k, c_void_p) for k in EULER3D_METRIC_FIELDS]


SW_METRIC_FIELDS = (
    "sqrtG", "H_contra_11", "H_contra_12", "H_contra_21", "H_contra_22",
    "christoffel_1_01", "christoffel_1_02", "christoffel_1_11", "christoffel_1_12",
    "christoffel_2_01", "christoffel_2_02", "christoffel_2_12", "christoffel_2_22",
    "sqrtG_itf_i", "sqrtG_itf_j", "H_contra_11_itf_i", "H_contra_21_itf_i", "H_contra_12_itf_j", "H_contra_22_itf_j",
    "hsurf", "dzdx1", "dzdx2", "hsurf_itf_i", "hsurf_itf_j", "boundary_sn", "boundary_we",
)


class SwMetric(ctypes.Structure):
    _fields_ = [(k, c_void_p) for k in SW_METRIC_FIELDS]


# every symbol include/wxhip.h declares: (restype, argtypes)
SIGNATURES = {
    "wx_last_error": (c_char_p, []),
    "wx_version": (c_char_p, []),
    "wx_build_info": (c_char_p, []),
    "wx_device_count": (c_int, []),
    "wx_stream_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wx_stream_read_sink_doubles": (c_int, []),
    "wx_stream_read": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "wx_phase_timer_create": (c_int, [POINTER(c_void_p)]),
    "wx_phase_timer_destroy": (c_int, [c_void_p]),
    "wx_phase_timer_stamp": (c_int, [c_void_p, c_int, c_void_p]),
    "wx_phase_timer_elapsed": (c_int, [c_void_p, POINTER(c_double)]),
    "wx_phase_timer_since_start": (c_int, [c_void_p, POINTER(c_double)]),
    "wx_stream_priority_range": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "wx_stream_create": (c_int, [POINTER(c_void_p), c_int]),
    "wx_stream_destroy": (c_int, [c_void_p]),
    "wx_euler3d_plan_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int,
                                       POINTER(DfrOps), POINTER(Euler3DMetric)]),
    "wx_euler3d_plan_create_tile": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int),
                                            POINTER(DfrOps), POINTER(Euler3DMetric)]),
    "wx_euler3d_plan_destroy": (c_int, [c_void_p]),
    "wx_euler3d_edge_count": (c_size_t, [c_void_p]),
    "wx_euler3d_bytes_per_point": (c_double, [c_void_p]),
    "wx_euler3d_plan_set_column_metric": (c_int, [c_void_p, c_void_p]),
    "wx_euler3d_plan_has_column_metric": (c_int, [c_void_p]),
    "wx_euler3d_uses_matrix_cores": (c_int, [c_void_p, c_int]),
    "wx_lean_log": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wx_euler3d_plan_one_kernel": (c_int, [c_void_p]),
    "wx_euler3d_plan_set_one_kernel": (c_int, [c_void_p, c_int]),
    "wx_euler3d_extrap_pack": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p]),
    "wx_euler3d_rhs": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_int, c_void_p]),
    "wx_euler3d_rhs_axpy": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_double, c_double,
                                    c_double, c_int, c_void_p]),
    "wx_euler3d_rhs_axpy2": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_double,
                                     c_double, c_double, c_double, c_int, c_void_p]),
    "wx_euler3d_shifted_extrap_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), c_void_p]),
    "wx_euler3d_shifted_rhs_axpy2": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), c_void_p, c_void_p,
                                             c_void_p, c_double, c_double, c_double, c_double, c_int, c_void_p]),
    "wx_euler3d_extrap_pack_slot": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_int, c_void_p]),
    "wx_euler3d_set_exp_filter": (c_int, [c_void_p, POINTER(c_double)]),
    "wx_euler3d_stage": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_double, c_double,
                                 c_double, c_double, c_int, c_int, POINTER(c_void_p), c_int, c_void_p, c_void_p]),
    "wx_multi_dot_workspace": (c_size_t, [c_int]),
    "wx_multi_dot": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "wx_multi_axpy": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p]),
    "wx_multi_axpy_scaled": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_double, c_void_p]),
    "wx_krylov_aug_update": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "wx_pmex_workspace": (c_size_t, [c_int]),
    "wx_euler3d_batch_pmex_vector": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_size_t, c_int, c_double, c_double,
                                             c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int, c_size_t, c_void_p]),
    "wx_pmex_vector": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                               c_double, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "wx_pmex_vector_split": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_double, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "wx_kiops_finish_workspace": (c_size_t, [c_size_t]),
    "wx_kiops_finish": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "wx_kiops_long_workspace": (c_size_t, []),
    "wx_kiops_long_a": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "wx_kiops_long_b": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wx_kiops_long_c": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "wx_kiops_long_a_scaled": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "wx_kiops_long_a_formed": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wx_kiops_long_a_finish": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "wx_kiops_long_b_scaled": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "wx_kiops_long_c_lazy": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wx_kiops_long_b_fold_finish": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                            c_void_p, c_void_p]),
    "wx_multi_dot2": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "wx_pair_update": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_size_t, c_double,
                               c_double, c_double, c_void_p]),
    "wx_fgmres_workspace": (c_size_t, [c_int]),
    "wx_fgmres_rotate_columns": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                         c_void_p, c_void_p, c_void_p]),
    "wx_fgmres_back_substitute": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "wx_fgmres_vector": (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "wx_euler3d_batch_fgmres_vector": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_size_t, c_double,
                                               c_double, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
    "wx_expfilter_create": (c_int, [POINTER(c_void_p), c_int, POINTER(c_double)]),
    "wx_expfilter_destroy": (c_int, [c_void_p]),
    "wx_expfilter_uses_matrix_cores": (c_int, [c_void_p, c_int]),
    "wx_expfilter_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_void_p, c_void_p]),
    "wx_expfilter_apply_stacked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_int, c_void_p,
                                           c_void_p]),
    "wx_check_nan": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    "wx_cart2d_sponge": (c_int, [c_void_p, c_void_p, c_double, c_size_t, c_int, c_void_p]),
    "wx_euler3d_batch_create": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, c_void_p, c_void_p]),
    "wx_euler3d_batch_pulls": (c_int, [c_void_p]),
    "wx_euler3d_batch_destroy": (c_int, [c_void_p]),
    "wx_euler3d_batch_extrap_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_size_t, c_void_p]),
    "wx_euler3d_batch_rhs_axpy2": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_int, c_double, c_double, c_double, c_double, c_int, c_void_p]),
    "wx_euler3d_batch_jvp": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_double, c_size_t, c_int, c_void_p]),
    "wx_euler3d_batch_kiops_vector": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_double,
                                              c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wx_euler3d_jvp_prepare": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p]),
    "wx_euler3d_jvp_tangent_extrap_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), c_void_p]),
    "wx_euler3d_jvp_tangent_extrap_pack_fix": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wx_euler3d_jvp_prepared": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                        c_double, c_int, c_void_p]),
    "wx_euler3d_jvp_prepared_axpy": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                             c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "wx_euler3d_jvp_workgroups": (c_size_t, [c_void_p, c_int]),
    "wx_euler3d_jvp_extrap_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), c_void_p]),
    "wx_euler3d_jvp": (c_int, [c_void_p, c_void_p, c_void_p, c_double, POINTER(c_void_p), c_void_p, c_double, c_int,
                               c_void_p]),
    "wx_sw_plan_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, POINTER(DfrOps), POINTER(SwMetric)]),
    "wx_sw_plan_create_tile": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(DfrOps),
                                       POINTER(SwMetric)]),
    "wx_sw_plan_destroy": (c_int, [c_void_p]),
    "wx_sw_edge_count": (c_size_t, [c_void_p]),
    "wx_sw_plan_dtype": (c_int, [c_void_p]),
    "wx_sw_extrap_pack": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p]),
    "wx_sw_rhs": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_int, c_void_p]),
    "wx_sw_rhs_axpy": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_double, c_double, c_double,
                               c_int, c_void_p]),
    "wx_sw_batch_rhs_axpy": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_double, c_double, c_double,
                                     c_int, c_void_p]),
    "wx_sw_extrap_pack_ring": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p]),
    "wx_sw_rhs_direct": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_double, c_double, c_double, c_int,
                                 c_int, c_void_p]),
    "wx_sw_batch_extrap_pack_ring": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wx_sw_batch_direct_pulls": (c_int, [c_void_p, POINTER(c_int)]),
    "wx_sw_batch_rhs_direct": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_double, c_double, c_double, c_int,
                                       c_int, c_void_p]),
    "wx_sw_plan_reserve": (c_int, [c_void_p, c_int]),
    "wx_sw_extrap_pack_slot": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_int, c_void_p]),
    "wx_sw_stage": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_double, c_double, c_double, c_int,
                            c_int, POINTER(c_void_p), c_int, c_void_p]),
    "wx_sw_batch_create_pipelined": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "wx_sw_batch_extrap_pack_slot": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "wx_sw_batch_stage": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_double, c_double, c_double, c_int, c_int,
                                  c_int, c_void_p]),
    "wx_sw_batch_create": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, c_void_p, c_void_p]),
    "wx_sw_batch_destroy": (c_int, [c_void_p]),
    "wx_sw_batch_extrap_pack": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wx_sw_batch_rhs": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "wx_pointwise_eulercartesian_2d": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wx_riemann_eulercartesian_ausm_2d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                  c_void_p]),
    "wx_forcing_euler_cubesphere_3d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                               c_int, c_int, c_int, c_void_p]),
    "wx_cart2d_plan_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_double, c_double, c_int, POINTER(DfrOps)]),
    "wx_cart2d_plan_destroy": (c_int, [c_void_p]),
    "wx_cart2d_rhs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "wx_euler3d_plan_dtype": (c_int, [c_void_p]),
    "wx_euler3d_plan_reserve": (c_int, [c_void_p, c_int]),
    "wx_euler3d_plan_reserved": (c_int, [c_void_p]),
    "wx_comm_rccl_version": (c_int, []),
    "wx_comm_unique_id": (c_int, [c_void_p]),
    "wx_comm_init_rank": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_int]),
    "wx_comm_adopt": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int]),
    "wx_comm_destroy": (c_int, [c_void_p]),
    "wx_comm_users": (c_int, [c_void_p]),
    "wx_comm_allreduce": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "wx_hip_runtime_version": (c_int, []),
    "wx_hip_driver_version": (c_int, []),
    "wx_exchange_create": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_size_t, c_int]),
    "wx_exchange_destroy": (c_int, [c_void_p]),
    "wx_exchange_local_tiles": (c_int, [c_void_p, POINTER(c_int), c_int]),
    "wx_exchange_neighbor": (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "wx_exchange_needs_comm": (c_int, [c_void_p]),
    "wx_exchange_send_doubles": (c_size_t, [c_void_p]),
    "wx_exchange_recv_doubles": (c_size_t, [c_void_p]),
    "wx_exchange_peer_counts": (c_int, [c_void_p, POINTER(c_size_t), POINTER(c_size_t)]),
    "wx_exchange_bind": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wx_exchange_send_ptr": (c_void_p, [c_void_p, c_int, c_int]),
    "wx_exchange_halo_ptr": (c_void_p, [c_void_p, c_int, c_int]),
    "wx_exchange_set_timer": (c_int, [c_void_p, c_void_p]),
    "wx_exchange_start": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wx_exchange_wait": (c_int, [c_void_p, c_void_p]),
    "wx_exchange_fork": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wx_exchange_join": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wx_euler3d_rhs_overlapped": (c_int, [POINTER(c_void_p), c_int, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                          c_void_p]),
    "wx_sw_rhs_overlapped": (c_int, [POINTER(c_void_p), c_int, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                     c_void_p]),
}

_lib = None


def load():
    """Load libwxhip.so (once).  Raises WxError when it is not built: no silent fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WxError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m wxfactory_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != WX_OK:
        msg = load().wx_last_error().decode(errors="replace")
        raise WxError(f"{what} failed (status {status}): {msg}")
