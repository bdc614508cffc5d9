"""ctypes binding of libcplxamd.so (the C ABI declared in include/cplxamd.h).

The library is the product: there is NO fallback.  If it is missing, was built for another
ABI version, or a kernel is asked to run on a non-HIP tensor, this module raises.
"""
import ctypes
import os
import threading
from ctypes import c_double, c_float, c_int, c_int64, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPLXAMD_LIB") or os.path.join(_HERE, "libcplxamd.so")   # env: A/B builds
ABI_VERSION = 24

F32, BF16, F16 = 0, 1, 2
KL_KINDS = {"real_vd": 0, "real_ard": 1, "cplx_vd": 2, "cplx_ard": 3, "cplx_vd_approx": 4,
            "cplx_vd_scalefree": 5, "cplx_vd_bogus": 6}

_P, _I, _L, _U, _F, _D = c_void_p, c_int, c_int64, c_uint64, c_float, c_double

# name -> argtypes; restype is int unless listed in _RESTYPES
SIGNATURES = {
    "cplxamd_abi_version": [],
    "cplxamd_vd_kl_ws_bytes": [],
    "cplxamd_vd_kl_fwd": [_P, _P, _P, _I, _P, _P, _P, _L, _P],
    "cplxamd_vd_kl_bwd": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _P],
    "cplxamd_vd_kl_fwd_bwd": [_P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _L, _P],
    "cplxamd_vd_log_alpha": [_P, _P, _P, _P, _L, _P],
    "cplxamd_vd_log_alpha_bwd": [_P, _P, _P, _P, _P, _L, _P],
    "cplxamd_cplx_abs_fwd": [_P, _P, _P, _L, _I, _P],
    "cplxamd_cplx_abs_bwd": [_P, _P, _P, _P, _P, _L, _I, _P],
    "cplxamd_mask_mul": [_P, _P, _P, _P, _P, _L, _I, _I, _P],
    "cplxamd_vd_mask": [_P, _P, _P, _F, _P, _P, _P, _L, _P],
    "cplxamd_expi_fwd": [_P, _P, _L, _P],
    "cplxamd_expi_bwd": [_P, _P, _P, _L, _P],
    "cplxamd_lrt_reparam_fwd": [_P, _P, _P, _P, _P, _U, _U, _P, _P, _P, _L, _I, _P],
    "cplxamd_lrt_reparam_bwd": [_P, _P, _P, _P, _P, _U, _U, _P, _P, _L, _I, _I, _P],
    "cplxamd_lrt_reparam_fwd_ex": [_P, _P, _P, _P, _P, _U, _U, _P, _P, _P, _L, _I, _I, _P],
    "cplxamd_lrt_reparam_bwd_ex": [_P, _P, _P, _P, _P, _U, _U, _P, _P, _L, _I, _I, _I, _P],
    "cplxamd_lrt_reparam_bwd_cols_ws_bytes": [_L, _I],
    "cplxamd_lrt_reparam_bwd_cols": [_P, _P, _P, _P, _P, _U, _U, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _L, _P],
    "cplxamd_philox_advance": [_P, _P, _P],
    "cplxamd_philox_normal": [_P, _P, _U, _U, _L, _P],
    "cplxamd_cgemm": [_P, _P, _L, _L, _P, _P, _L, _L, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I,
                      _I, _I, _I, _P, _L, _P],
    "cplxamd_cgemm_ex": [_P, _P, _L, _L, _P, _P, _L, _L, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I,
                         _I, _I, _P, _I, _P, _L, _P],
    "cplxamd_rgemm_ex": [_P, _L, _L, _P, _L, _L, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P],
    "cplxamd_vd_prep_kl": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P],
    "cplxamd_cgemm_lrt_dx": [_P, _P, _L, _L, _P, _P, _L, _L, _P, _P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P],
    "cplxamd_rgemm_lrt_dx": [_P, _L, _L, _P, _L, _L, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "cplxamd_cgemm_batched": [_P, _P, _L, _L, _L, _P, _P, _L, _L, _L, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "cplxamd_rgemm": [_P, _L, _L, _P, _L, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "cplxamd_gemm_ws_bytes": [_I, _I, _I, _I, _I, _I],
    "cplxamd_cplx_maxpool2d_fwd": [_P, _P, _P, _P, _P, _P, _I, _P],
    "cplxamd_cplx_maxpool2d_bwd": [_P, _P, _P, _P, _P, _P, _I, _P],
    "cplxamd_cplx_maxpool2d_fwd_cl": [_P, _P, _P, _P, _P, _P, _I, _P],
    "cplxamd_cplx_maxpool2d_bwd_cl": [_P, _P, _P, _P, _P, _P, _I, _P],
    "cplxamd_bilinear_reduce_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "cplxamd_bilinear_reduce_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "cplxamd_deinterleave": [_P, _P, _P, _L, _I, _P],
    "cplxamd_interleave": [_P, _P, _P, _L, _I, _P],
    "cplxamd_cplx_mul": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "cplxamd_split_relu": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "cplxamd_modrelu_fwd": [_P, _P, _P, _F, _I, _P, _P, _L, _I, _P],
    "cplxamd_modrelu_bwd": [_P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _L, _I, _P],
    "cplxamd_cplx_dropout": [_P, _P, _P, _P, _D, _U, _U, _P, _L, _I, _P],
    "cplxamd_nhwc_pad": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "cplxamd_nhwc_pad_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "cplxamd_conv2d_nhwc_f32": [_P, _P, _P, _P, _P, _P, _P, _P] + [_I] * 10 + [_L] + [_I] * 4 + [_P],
    "cplxamd_conv2d_nhwc_wgrad_ws_bytes": [_I] * 8,
    "cplxamd_conv2d_nhwc_wgrad_f32_ws_bytes": [_I] * 8,
    "cplxamd_conv2d_nhwc_wgrad_f32": [_P, _P, _P, _P, _P, _P, _P] + [_I] * 9 + [_P, _L, _P],
    "cplxamd_conv2d_nhwc_wgrad": [_P, _P, _P, _P, _P, _P, _P] + [_I] * 9 + [_P, _L, _P],
    "cplxamd_conv2d_nhwc": [_P, _P, _P, _P, _P, _P, _P, _P] + [_I] * 10 + [_L] + [_I] * 5 + [_P],
    "cplxamd_cl_to_nchw": [_P, _P, _L, _I, _L, _P],
    "cplxamd_conv2d_clr_pack_bytes": [_I, _I, _I, _I],
    "cplxamd_conv2d_clr_ws_bytes": [_I],
    "cplxamd_conv2d_clr_pack": [_P, _P, _I, _I, _I, _I, _I, _P],
    "cplxamd_conv2d_clr": [_P, _P, _P, _P, _L] + [_I] * 11 + [_P, _L, _P],
    "cplxamd_conv2d_clr_wgrad_ws_bytes": [_L, _I, _I, _I, _I],
    "cplxamd_conv2d_clr_wgrad": [_P, _P, _P, _I, _P, _L] + [_I] * 10 + [_P, _L, _P],
    "cplxamd_conv2d_cl_pack_bytes": [_I, _I, _I, _I],
    "cplxamd_conv2d_cl_ws_bytes": [_I],
    "cplxamd_conv2d_cl_pack": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cplxamd_conv2d_cl": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 11 + [_P, _L, _P],
    "cplxamd_conv2d_cl2": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 11 + [_P, _L, _P],
    "cplxamd_chansum2": [_P, _P, _P, _P, _L, _I, _L, _I, _P, _P],
    "cplxamd_conv2d_cl2_lrt_dx": [_P] * 8 + [_L] + [_I] * 6 + [_P, _L, _P],
    "cplxamd_conv2d_cl2_mom_chunks": [_L] + [_I] * 10,
    "cplxamd_conv2d_cl2_mom": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P, _L, _P, _L, _P],
    "cplxamd_conv2d_cl_wgrad_ws_bytes": [_L, _I, _I, _I, _I],
    "cplxamd_conv2d_cl_wgrad": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P, _L, _P],
    "cplxamd_cgemm3m_ws_bytes": [_I, _I, _I],
    "cplxamd_abs2": [_P, _P, _P, _L, _I, _I, _P],
    "cplxamd_modulus": [_P, _P, _P, _L, _P],
    "cplxamd_exp": [_P, _P, _L, _I, _P],
    "cplxamd_cast": [_P, _P, _L, _I, _I, _P],
    "cplxamd_split3": [_P, _P, _L, _P, _L, _L, _L, _I, _I, _I, _P],
    "cplxamd_absmax_ws_bytes": [],
    "cplxamd_absmax_scale": [_P, _P, _L, _L, _I, _I, _P, _P, _P],
    "cplxamd_split2h": [_P, _P, _L, _P, _L, _L, _L, _I, _I, _I, _P, _P],
    "cplxamd_cgemm_sc_fl": [_P, _P, _L, _L, _P, _P, _L, _L, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L,
                            _I, _P],
    "cplxamd_rgemm_sc_fl": [_P, _L, _L, _P, _L, _L, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _P],
    "cplxamd_transpose": [_P, _L, _P, _L, _I, _I, _I, _P],
    "cplxamd_colsum_ws_bytes": [_I],
    "cplxamd_colsum": [_P, _L, _P, _I, _I, _I, _P, _P],
    "cplxamd_colsum2": [_P, _P, _L, _P, _P, _I, _I, _I, _P, _P],
    "cplxamd_lrt_dx_accum": [_P, _P, _P, _P, _P, _L, _I, _I, _P],
    "cplxamd_conv2d_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "cplxamd_conv2d_dgrad": [_P, _P, _P, _P, _P, _P, _P, _I, _P],
    "cplxamd_conv2d_wgrad_splits": [_P],
    "cplxamd_conv2d_wgrad_ws_bytes": [_P, _I],
    "cplxamd_conv2d_wgrad": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _P],
    "cplxamd_conv2d_wgrad_bias": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _P],
    "cplxamd_conv2d_out_shape": [_P, _P, _P],
    "cplxamd_conv2d_ktab_size": [_P, _I],
    "cplxamd_conv2d_ktab_fill": [_P, _I, _P],
    "cplxamd_conv2d_bf16_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "cplxamd_conv2d_bf16_dgrad": [_P, _P, _P, _P, _P, _P, _P, _P, _P],
    "cplxamd_conv2d_bf16_wgrad_ws_bytes": [_P, _I],
    "cplxamd_conv2d_bf16_wgrad": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P],
    "cplxamd_chansum": [_P, _P, _L, _I, _L, _I, _P, _P],
    "cplxamd_bn_ws_bytes": [_I],
    "cplxamd_bn_fwd": [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _L, _P],
    "cplxamd_bn_fwd_ex": [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _L, _P],
    "cplxamd_bn_bwd": [_P, _P, _P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _I, _I, _P, _L, _P],
    "cplxamd_bn_bwd_sums": [_P, _P, _P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _I, _I, _P, _P, _L, _P],
    "cplxamd_bn_rows_path": [_L, _I, _L],
    "cplxamd_gemm_set_persistent": [_I],
    "cplxamd_gemm_set_family": [_I],
    # ABI 19: per-call launch policy (trailing `flags` in front of the stream) + the dispatch as a pure function
    "cplxamd_cgemm_fl": [_P, _P, _L, _L, _P, _P, _L, _L, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I,
                         _I, _I, _P, _I, _P, _L, _I, _P],
    "cplxamd_rgemm_fl": [_P, _L, _L, _P, _L, _L, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _L, _I, _P],
    "cplxamd_cgemm_lrt_dx_fl": [_P, _P, _L, _L, _P, _P, _L, _L, _P, _P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "cplxamd_rgemm_lrt_dx_fl": [_P, _L, _L, _P, _L, _L, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    "cplxamd_gemm_plan": [_I] * 10,
    "cplxamd_conv2d_cl_fl": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 11 + [_P, _L, _I, _P],
    "cplxamd_conv2d_cl2_fl": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 11 + [_P, _L, _I, _P],
    "cplxamd_conv2d_cl2_lrt_dx_fl": [_P] * 8 + [_L] + [_I] * 6 + [_P, _L, _I, _P],
    "cplxamd_conv2d_cl2_mom_chunks_fl": [_L] + [_I] * 11,
    "cplxamd_conv2d_cl2_mom_fl": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P, _L, _P, _L, _I, _P],
    "cplxamd_conv2d_cl_wgrad_fl": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P, _L, _I, _P],
    # ABI 24: batch-norm backward without its apply pass + the weight gradient that forms dX while staging it
    "cplxamd_bn_bwd_sums_amax": [_P, _P, _P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _I, _I, _P, _P, _P, _L, _P],
    "cplxamd_absmax_scale_partials": [_P, _I, _P, _P],
    "cplxamd_bn_bwd_coef": [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _I, _I, _P, _P, _P, _L, _P],
    "cplxamd_conv2d_cl_wgrad_bn_fl": [_P] * 11 + [_L] + [_I] * 10 + [_P, _L, _I, _P],
    # ABI 23: float64 contractions (parity mode)
    "cplxamd_gemm_f64": [_P, _P, _L, _L, _L, _P, _P, _L, _L, _L, _P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _P],
    "cplxamd_conv2d_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "cplxamd_expi_f64": [_P, _P, _L, _P],
    # ABI 22: the channels-last convolutions on IEEE-half pieces (float32 out)
    "cplxamd_conv2d_cl2h_fl": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _L] + [_I] * 7 + [_P, _L, _I, _P],
    "cplxamd_conv2d_cl2h_wrap_fl": [_P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _L] + [_I] * 7 + [_P, _L, _I, _P],
    "cplxamd_conv2d_clh_wgrad_ws_bytes": [_L, _I, _I, _I, _I],
    "cplxamd_conv2d_clh_wgrad": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P, _L, _P],
    "cplxamd_conv2d_clh_wgrad_fl": [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P, _L, _I, _P],
    "cplxamd_conv2d_clh_wgrad_skip_fl": [_P, _P, _P, _P, _P, _P, _L] + [_I] * 12 + [_P, _L, _I, _P],
    "cplxamd_conv2d_clr_fl": [_P, _P, _P, _P, _L] + [_I] * 11 + [_P, _L, _I, _P],
    "cplxamd_conv2d_clr_wgrad_fl": [_P, _P, _P, _I, _P, _L] + [_I] * 10 + [_P, _L, _I, _P],
    "cplxamd_bn_moments": [_P, _P, _P, _P, _P, _L, _I, _L, _I, _P, _P, _L, _P],
    "cplxamd_bn_fwd_sync": [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _L, _P],
    "cplxamd_bn_fwd_partials": [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _I, _P, _L, _P],
    "cplxamd_bn_bwd_sync": [_P, _P, _P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _P],
}
_RESTYPES = {"cplxamd_absmax_ws_bytes": c_int64, "cplxamd_conv2d_cl2_mom_chunks": c_int64, "cplxamd_conv2d_cl2_mom_chunks_fl": c_int64, "cplxamd_vd_kl_ws_bytes": c_int64, "cplxamd_lrt_reparam_bwd_cols_ws_bytes": c_int64, "cplxamd_bn_ws_bytes": c_int64,
             "cplxamd_conv2d_wgrad_ws_bytes": c_int64, "cplxamd_conv2d_bf16_wgrad_ws_bytes": c_int64, "cplxamd_colsum_ws_bytes": c_int64, "cplxamd_gemm_ws_bytes": c_int64,
             "cplxamd_cgemm3m_ws_bytes": c_int64,
             "cplxamd_conv2d_nhwc_wgrad_ws_bytes": c_int64,
             "cplxamd_conv2d_nhwc_wgrad_f32_ws_bytes": c_int64,
             "cplxamd_conv2d_cl_pack_bytes": c_int64, "cplxamd_conv2d_cl_ws_bytes": c_int64,
             "cplxamd_conv2d_cl_wgrad_ws_bytes": c_int64, "cplxamd_conv2d_clh_wgrad_ws_bytes": c_int64,

             "cplxamd_conv2d_clr_pack_bytes": c_int64,
             "cplxamd_conv2d_clr_ws_bytes": c_int64, "cplxamd_conv2d_clr_wgrad_ws_bytes": c_int64}

_lib = None

# ---- per-call launch policy (include/cplxamd.h CPLXAMD_LAUNCH_*) ----------------------------------------------------
# The C ABI keeps no mutable launch state: every GEMM / channels-last convolution call carries its flags.  WHO decides
# lives here, on the host: (1) an override registered for the STREAM the call is launched on (`with launch_policy(flags):`
# -- tests, A/B runs, a serving stream that knows it shares the chip; keyed by stream, not by thread, because autograd
# runs a layer's backward on an engine thread but on the forward's stream), else (2) LAUNCH_SHARED while any
# data-parallel hook of this process has collectives in flight (`shared_chip_enter` / `_leave`:
# cplxmodule_amd.dp.BucketHook; the window is a property of the CHIP, so every model launching into it shares), else
# (3) 0 = the library's defaults.  A hipGraph capture bakes in the flags of its capture pass, exactly as it bakes in every
# other launch argument.
LAUNCH_DEFAULT, LAUNCH_SHARED, LAUNCH_EXCLUSIVE = 0, 1, 2


def LAUNCH_FAMILY(mask):
    return 0x100 | ((int(mask) & 0xff) << 16)


_policy = {}               # (device index, raw stream handle) -> flags
_sharing = set()           # id() of the hooks whose collectives are in flight
_policy_lock = threading.Lock()


def _stream_key():
    if not torch.cuda.is_available():
        return (-1, 0)
    idx = torch.cuda.current_device()
    return (idx, _current_stream_handle(idx))


def launch_flags():
    """Flags of the next GEMM / convolution launch on the current stream."""
    if _policy:
        f = _policy.get(_stream_key())
        if f is not None:
            return f
    return LAUNCH_SHARED if _sharing else LAUNCH_DEFAULT


class launch_policy:
    """`with launch_policy(LAUNCH_SHARED):` -- launches on the stream that is current at entry carry these flags until
    exit (forward AND backward: autograd replays a node's backward on its forward's stream)."""

    def __init__(self, flags):
        self.flags = int(flags)

    def __enter__(self):
        self.key = _stream_key()
        with _policy_lock:
            self.prev = _policy.get(self.key)
            _policy[self.key] = self.flags
        return self

    def __exit__(self, *exc):
        with _policy_lock:
            if self.prev is None:
                _policy.pop(self.key, None)
            else:
                _policy[self.key] = self.prev
        return False


def shared_chip_enter(owner):
    with _policy_lock:
        _sharing.add(id(owner))


def shared_chip_leave(owner):
    with _policy_lock:
        _sharing.discard(id(owner))


class CplxAmdError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises CplxAmdError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CplxAmdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or cplxmodule_amd/csrc/build.sh).  cplxmodule_amd has no CPU or "
            "pure-torch fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise CplxAmdError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CplxAmdError(f"{LIB_PATH} does not export `{name}` (stale build?)") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.cplxamd_abi_version() != ABI_VERSION:
        raise CplxAmdError("libcplxamd.so ABI version mismatch: rebuild the library")
    _lib = lib
    return lib


_ERRORS = {-1: "invalid argument", -2: "misaligned pointer / leading dimension",
           -3: "unsupported shape", -4: "workspace too small"}


def try_call(name, *args):
    """Like `call`, but returns False when the entry point declines the shape
    (CPLXAMD_ESHAPE = "use the generic kernel"); every other failure raises."""
    rc = getattr(load(), name)(*args)
    if rc == -3:
        return False
    if rc != 0:
        what = _ERRORS.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise CplxAmdError(f"{name} failed: {what}")
    return True


def call(name, *args):
    """Invoke an entry point; non-zero return codes become exceptions."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        what = _ERRORS.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise CplxAmdError(f"{name} failed: {what}")


_empty_anchor = {}


def ptr(t):
    """Device pointer of a tensor for the C ABI (None -> NULL).  An EMPTY tensor has no storage
    (data_ptr() == 0), which the library would reject as a missing argument; it gets the address
    of a small per-device anchor instead -- with zero elements nothing is read or written there."""
    if t is None:
        return None
    p = t.data_ptr()
    if p == 0 and t.numel() == 0 and t.is_cuda:
        key = t.device.index
        if key not in _empty_anchor:
            _empty_anchor[key] = torch.zeros(64, dtype=torch.float32, device=t.device)
        p = _empty_anchor[key].data_ptr()
    return c_void_p(p)


# torch.cuda.current_stream() builds a Stream object per call (~8 us: a third of a millisecond per eager step of a
# 125-launch model); the raw handle is one C call
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_stream_handle(idx=None):
    if idx is None:
        idx = torch.cuda.current_device()
    if _raw_stream is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(idx).cuda_stream


def stream_ptr():
    return c_void_p(_current_stream_handle())


def scratch_key(device):
    """Key of the per-(device, stream) scratch caches (split-K slabs, BN / KL partial sums, conv slabs): kernels
    launched on two streams must not share a workspace.  During hipGraph capture the capturing stream's key is
    used like any other, so a cache grown inside a capture belongs to that capture's pool only."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return (dev.type, idx, _current_stream_handle(idx) if dev.type == "cuda" else 0)


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise CplxAmdError(f"unsupported dtype {t.dtype}: this kernel takes float32 and bfloat16 (float64 is offered for the "
                       "linear / convolution / batch-norm / relevance layers only: cplxmodule_amd/f64.py)")


def require_device(*tensors):
    """Every kernel argument must live on one HIP device; anything else is an error."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise CplxAmdError(
                "cplxmodule_amd runs on MI355X only: got a tensor on "
                f"'{t.device}'.  Move the module and its inputs to 'cuda' (there is no CPU path).")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise CplxAmdError(f"tensors on different devices: {dev} vs {t.device}")
    return dev
