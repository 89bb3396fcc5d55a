"""ctypes binding of libcine_hip.so (C ABI: include/cine_hip.h).

The library is looked up next to this file (built in-tree by
``make -C deep-cine-cardiac-mri_amd/csrc`` or ``__graft_entry__.build()``).
Loading is lazy so host-only helpers (synth, metrics) import without it, but
every compute call goes through ``lib()`` and raises ``CineHipError`` if the
library is missing -- there is no CPU fallback.
"""
import ctypes
import os
import re
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CINE_HIP_LIB") or os.path.join(_HERE, "libcine_hip.so")   # override: A/B builds of the library
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "cine_hip.h"))


class CineHipError(RuntimeError):
    pass


P = c_void_p
_SIGS = {
    "cine_version": (c_int, []),
    "cine_last_error": (c_char_p, []),
    "cine_build_arch": (c_char_p, []),
    "cine_pad16": (c_int, [c_int]),
    "cine_fft2c": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "cine_fft1c": (c_int, [P, P, c_long, c_int, c_int, c_int, P]),
    "cine_sens_reduce": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_sens_expand_dc": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_kspace_to_hybrid": (c_int, [P, P, c_long, c_int, c_int, P]),
    "cine_hybrid_reduce": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_expand_dc_hybrid": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_image_dc_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cine_normal_op": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_image_dc": (c_int, [P, P, P, P, P, c_float, c_float, c_float, P, c_int, c_int, c_int, c_int, c_int, c_int,
                              P, c_size_t, P]),
    "cine_masked_kspace_to_hybrid": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "cine_apply_mask": (c_int, [P, P, P, c_long, c_int, c_int, c_int, P]),
    "cine_scale": (c_int, [P, c_long, c_float, P]),
    "cine_zero_filled_rss": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_image_metrics_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "cine_image_metrics": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_double, c_int, c_double,
                                   P, P, c_size_t, P]),
    "cine_sens_prologue": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_acs_window": (c_int, [P, c_int, c_int, P, P]),
    "cine_sens_prologue_win": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "cine_rss_normalise": (c_int, [P, c_int, c_int, c_int, c_int, P]),
    "cine_normunet_pack": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "cine_normunet_unpack": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "cine_xfyf_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cine_xfyf_pack": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_xfyf_unpack": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_conv3x3_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_tconv2x2_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_conv1x1_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_pack_conv3x3": (c_int, [P, P, c_int, c_int, P]),
    "cine_pack_tconv2x2": (c_int, [P, P, c_int, c_int, P]),
    "cine_pack_conv1x1": (c_int, [P, P, c_int, c_int, P]),
    "cine_conv_stat_partials": (c_int, [c_int, c_int, c_int, c_int]),
    "cine_conv3x3_in": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int,
                                P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "cine_conv3x3_ex": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "cine_crnn_step": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_crnn_step2": (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_conv3x3_dgrad_gated": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_bcrnn_sweep": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_bcrnn_sweep_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_mwcnn_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int]),
    "cine_mwcnn_forward2": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "cine_conv3x3_ex2": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                 P, P, P, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "cine_mwcnn_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "cine_tconv2x2_in": (c_int, [P, P, c_int, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int,
                                 c_float, c_float, P]),
    "cine_conv1x1_bias": (c_int, [P, P, c_int, c_int, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int,
                                  c_float, c_float, P]),
    "cine_instnorm_partials": (c_int, [P, P, c_long, c_long, P]),
    "cine_instnorm_finalize": (c_int, [P, P, c_long, c_int, c_float, P]),
    "cine_instnorm_lrelu_apply": (c_int, [P, P, c_int, P, c_long, c_long, c_float, c_float, P]),
    "cine_unet2d_ws_bytes": (c_size_t, [c_int] * 7),
    "cine_unet2d_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "cine_complex_abs": (c_int, [P, P, c_long, P]),
    "cine_conv3d_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_tconv3d_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_pack_conv3d": (c_int, [P, P, c_int, c_int, P]),
    "cine_pack_tconv3d": (c_int, [P, P, c_int, c_int, P]),
    "cine_conv_stat_partials3d": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "cine_conv3d_in": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                               P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "cine_tconv3d_in": (c_int, [P, P, c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "cine_conv1x1x1_bias": (c_int, [P, P, c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "cine_instnorm_merge": (c_int, [P, P, c_long, c_int, P]),
    "cine_pool3d_act": (c_int, [P, P, c_int, P, c_long, c_int, c_int, c_int, c_float, c_float, P]),
    "cine_conv3d_pools_on_load": (c_int, [c_int, c_int, c_int, c_int]),
    "cine_unet3d_ws_bytes": (c_size_t, [c_int] * 8),
    "cine_unet3d_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "cine_normunet3d_pack": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_normunet3d_unpack": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "cine_mwcnn_pad": (c_int, [c_int, c_int, P, P]),
    "cine_xpd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cine_xpd_pack": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_xpd_unpack": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_chanlast_to_planes": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_planes_to_chanlast": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_extract_complex": (c_int, [P, P, c_long, c_int, c_int, c_int, P]),
    "cine_repeat_complex": (c_int, [P, P, c_long, c_int, P]),
    "cine_dot_ws_bytes": (c_size_t, []),
    "cine_cg_ws_bytes": (c_size_t, []),
    "cine_cg_step": (c_int, [P, P, P, P, c_long, P, P, P, P]),
    "cine_cg_step_pd": (c_int, [P, P, P, P, c_long, P, P, P, P]),
    "cine_cg_step_pd2": (c_int, [P, P, P, P, c_long, P, P, P, P, P]),
    "cine_cg_fused_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cine_normal_op_cg_fused_t": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P, c_size_t, P]),
    "cine_conj_grad_ws_bytes": (c_size_t, [c_int] * 5),
    "cine_conj_grad": (c_int, [P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_conj_grad_rec": (c_int, [P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P, P, P, P]),
    "cine_sens_tile_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cine_sens_tile_pack": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "cine_image_dc_t": (c_int, [P, P, P, P, P, P, c_float, c_float, c_float, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_normal_op_t": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_normal_op_cg_fused": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P, c_size_t, P]),
    "cine_cg_adjoint_part_floats": (c_size_t, []),
    "cine_cg_adjoint_step": (c_int, [P, P, P, P, P, c_long, P, P, P, P, P]),
    "cine_cg_adjoint_finish": (c_int, [P, P, P, c_int, P, P]),
    "cine_normal_op_pd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_dot": (c_int, [P, P, c_long, P, P, P]),
    "cine_axpby_dev": (c_int, [P, P, P, c_long, P, P, P, c_float, P]),
    "cine_crop_select": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_gauss_axis": (c_int, [P, P, c_long, c_int, c_long, c_double, P]),
    "cine_combine_target": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_espirit_lag_kernels": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "cine_espirit_eig": (c_int, [P, P, P, c_int, c_long, c_int, c_float, P]),
    "cine_conv3x3_dgrad_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_tconv2x2_dgrad_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_conv1x1_dgrad_packed_floats": (c_size_t, [c_int, c_int]),
    "cine_pack_conv3x3_dgrad": (c_int, [P, P, c_int, c_int, P]),
    "cine_pack_tconv2x2_dgrad": (c_int, [P, P, c_int, c_int, P]),
    "cine_pack_conv1x1_dgrad": (c_int, [P, P, c_int, c_int, P]),
    "cine_pack_desc_bytes": (c_size_t, []),
    "cine_pack_desc": (c_int, [P, c_int, P, P, c_int, c_int]),
    "cine_pack_batch": (c_int, [P, c_int, c_long, P]),
    "cine_conv3x3_dgrad": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_tconv2x2_dgrad": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_conv1x1_dgrad": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_unet2d_train_ws_bytes": (c_size_t, [c_int] * 7),
    "cine_unet2d_forward_train": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "cine_complex_mul": (c_int, [P, P, P, c_int, P, P, P, P]),
    "cine_complex_conj": (c_int, [P, P, c_long, P]),
    "cine_complex_abs_sq": (c_int, [P, P, c_long, P]),
    "cine_rss": (c_int, [P, P, c_long, c_int, c_long, c_int, P]),
    "cine_roll": (c_int, [P, P, c_long, c_int, c_long, c_int, P]),
    "cine_pad2d": (c_int, [P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_set_side_stream": (c_int, [P]),
    "cine_set_conv_plane": (c_int, [c_int]),
    "cine_unet3d_train_ws_bytes": (c_size_t, [c_int] * 8),
    "cine_unet3d_forward_train": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "cine_unet3d_backward_ws_bytes": (c_size_t, [c_int] * 8),
    "cine_unet3d_backward": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P, c_size_t, P, P]),
    "cine_unet2d_branch_ws_bytes": (c_size_t, [c_int] * 10),
    "cine_unet2d_forward_branches": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P, P, c_int, c_int, P]),
    "cine_unet2d_drop_floats": (c_size_t, [c_int, c_int, c_int]),
    "cine_unet3d_forward_train_drop": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P, P]),
    "cine_unet3d_backward_drop": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P, c_size_t, P, P, P]),
    "cine_unet2d_backward_drop": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P, c_size_t, P, P, P]),
    "cine_diag_counter": (c_long, [c_int, c_int]),
    "cine_spin": (c_int, [c_int, P]),
    "cine_unet2d_backward_ws_bytes": (c_size_t, [c_int] * 7),
    "cine_unet2d_backward": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P, c_size_t, P, P]),
    "cine_mwcnn_train_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int]),
    "cine_mwcnn_forward_train": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "cine_mwcnn_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int]),
    "cine_mwcnn_backward": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_float, P, c_size_t, P, c_size_t, P, P]),
    "cine_relu_mask": (c_int, [P, P, c_long, P]),
    "cine_conv3x3_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cine_conv3x3_wgrad": (c_int, [P, c_int, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_conv1x1_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cine_conv1x1_wgrad": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "cine_in_lrelu_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cine_in_lrelu_bwd": (c_int, [P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P, c_size_t, P]),
    "cine_xpd_unpack_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_xpd_pack_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_normunet_unpack_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "cine_normunet_pack_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "cine_xfyf_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cine_xfyf_unpack_bwd": (c_int, [P] * 10 + [c_int] * 5 + [P, c_size_t, P]),
    "cine_xfyf_pack_bwd": (c_int, [P] * 10 + [c_int] * 5 + [P, c_size_t, P]),
    "cine_image_dc_sens_grad": (c_int, [P, P, P, P, P, c_float, c_float, P, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_coil_accum": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "cine_rss_normalise_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "cine_complex_abs_bwd": (c_int, [P, P, P, c_long, P]),
    "cine_axpby_lam": (c_int, [P, P, P, c_long, P, c_int, c_float, P]),
    "cine_ssim_loss_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cine_ssim_loss": (c_int, [P, P, c_int, c_int, c_int, c_int, c_double, c_double, P, P, c_size_t, P]),
    "cine_ssim_loss_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_size_t, P, P]),
    "cine_profile_begin": (c_int, []),
    "cine_profile_end": (c_int, [P, P, c_int]),
    "cine_profile_families": (c_int, []),
    "cine_profile_family_name": (c_char_p, [c_int]),
}

_lib = None


def declared_symbols(header_path: str = HEADER_PATH):
    """Every function name include/cine_hip.h declares (used by the symbol-export test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cine_[a-z0-9_]+)\s*\(", text)))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CineHipError(
                f"{LIB_PATH} not found: build it with `make -C deep-cine-cardiac-mri_amd/csrc` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = lib().cine_last_error().decode(errors="replace")
        raise CineHipError(f"{what or 'cine_hip'} failed ({code}): {msg}")
