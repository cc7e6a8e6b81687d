"""ctypes binding of libcreamfl_hip.so (C ABI declared in include/creamfl_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call
returns non-zero, this module raises.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C creamfl_amd/csrc``.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_uint, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libcreamfl_hip.so')

_P = c_void_p

# name -> (restype, argtypes); mirrors include/creamfl_hip.h one to one
SIGNATURES = {
    'cfl_version': (c_int, []),
    'cfl_arch': (c_char_p, []),
    'cfl_num_kernels': (c_int, []),
    'cfl_kernel_name': (c_char_p, [c_int]),
    'cfl_prof_enable': (c_int, [c_int]),
    'cfl_prof_select': (c_int, [c_int]),
    'cfl_prof_reset': (c_int, []),
    'cfl_prof_query': (c_int, [c_int, POINTER(c_longlong), POINTER(c_double)]),
    'cfl_pair_loss_ws_bytes': (c_size_t, [c_int, c_int]),
    'cfl_pair_loss_fwd': (c_int, [_P, _P, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P]),
    'cfl_pair_loss_bwd': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    'cfl_bank_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cfl_bank_lse_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P]),
    'cfl_bank_lse_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    'cfl_get_exact_gemm': (c_int, []),
    'cfl_set_exact_gemm': (c_int, [c_int]),
    'cfl_client_contrast_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P]),
    'cfl_bank_image_bytes': (c_size_t, [c_int, c_int]),
    'cfl_bank_image_build': (c_int, [_P, c_int, c_int, _P, _P]),
    'cfl_bank_gsplit_supported': (c_int, [c_int, c_int, c_int]),
    'cfl_bank_gsplit_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'cfl_client_contrast_img_fwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int,
                                            _P, _P, _P, _P, _P, _P, _P, _P]),
    'cfl_intra_ws_bytes': (c_size_t, [c_int]),
    'cfl_intra_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    'cfl_kd_mse': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    'cfl_gemm_bf16_nt': (c_int, [_P, c_longlong, _P, c_longlong, _P, c_longlong, c_int, c_int, c_int, c_int, _P]),
    'cfl_gemm_bf16_nt_join': (c_int, [_P, c_longlong, _P, c_longlong, _P, _P, _P, c_int, c_int, c_int, _P]),
    'cfl_gemm_bf16_bres_min_m': (c_int, [c_int]),
    'cfl_gemm_bf16_nt_stats_nblk': (c_int, [c_int, c_int, c_int]),
    'cfl_gemm_bf16_nt_stats': (c_int, [_P, c_longlong, _P, c_longlong, _P, c_int, c_int, c_int, _P, _P]),
    'cfl_transpose_bf16': (c_int, [_P, c_int, c_int, _P, _P]),
    'cfl_transpose_bf16_multi': (c_int, [_P, c_int, c_int, _P]),
    'cfl_daln_ws_bytes': (c_size_t, [c_int, c_int]),
    'cfl_daln_fwd': (c_int, [_P, _P, c_int, _P, _P, _P, c_int, c_int, c_float, c_float, c_uint, _P, _P, _P, _P, _P]),
    'cfl_daln_bwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_uint, _P, _P, _P, _P, c_int, _P, _P]),
    'cfl_preln_bwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    'cfl_bias_gelu_ws_bytes': (c_size_t, [c_longlong, c_int]),
    'cfl_bias_gelu_fwd': (c_int, [_P, _P, c_int, c_longlong, c_int, _P, _P]),
    'cfl_bias_gelu_bwd': (c_int, [_P, _P, c_int, _P, c_longlong, c_int, _P, _P, c_int, _P, _P]),
    'cfl_dropout_mask': (c_int, [c_uint, c_float, c_longlong, _P, _P]),
    'cfl_set_dropout_tick': (c_int, [_P]),
    'cfl_attn_small_fwd': (c_int, [_P, _P, _P, c_longlong, c_longlong, _P, c_int, c_int, c_int, c_int, _P, c_longlong, c_longlong, _P]),
    'cfl_attn_small_bwd': (c_int, [_P, _P, _P, c_longlong, c_longlong, _P, c_int, c_int, c_int, c_int, _P, c_longlong, c_longlong,
                                   _P, _P, _P, c_longlong, c_longlong, _P]),
    'cfl_attn_small_fwd_varlen': (c_int, [_P, _P, _P, c_longlong, _P, c_int, c_int, c_int, _P, c_longlong, _P]),
    'cfl_attn_small_bwd_varlen': (c_int, [_P, _P, _P, c_longlong, _P, c_int, c_int, c_int, _P, c_longlong, _P, _P, _P, c_longlong, _P]),
    'cfl_stem_s2d': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    'cfl_maxpool3s2_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'cfl_maxpool3s2_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    'cfl_sup_ws_bytes': (c_size_t, [c_int, c_int]),
    'cfl_sup_glue_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_float, _P, _P, _P]),
    'cfl_sup_glue_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P]),
    'cfl_gru_supported': (c_int, [c_int]),
    'cfl_gru_streams_weights': (c_int, [c_int]),
    'cfl_gru_fwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'cfl_gru_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'cfl_gru_cell0_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    'cfl_gru_cell0_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    'cfl_conw_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cfl_conw_logprob': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'cfl_conw_combine': (c_int, [POINTER(c_void_p), _P, c_int, c_int, c_int, _P, _P, _P]),
    'cfl_conw_img_supported': (c_int, [c_int, c_int, c_int]),
    'cfl_conw_img_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cfl_conw_logprob_img': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'cfl_pie_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'cfl_pie_pool_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'cfl_pie_pool_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'cfl_pie_fused_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'cfl_pie_head_fwd': (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    'cfl_pie_head_bwd': (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'cfl_pie_epilogue_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P]),
    'cfl_pie_epilogue_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'cfl_l2norm_fwd': (c_int, [_P, c_int, c_int, _P, _P, _P]),
    'cfl_l2norm_bwd': (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    'cfl_rank_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cfl_rank_count': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    'cfl_bn_sliced': (c_int, [c_int]),
    'cfl_conv1x1_wgrad_supported': (c_int, [c_longlong, c_int, c_int]),
    'cfl_conv1x1_wgrad_ws_bytes': (c_size_t, [c_longlong, c_int, c_int]),
    'cfl_conv1x1_wgrad': (c_int, [_P, _P, c_longlong, c_int, c_int, _P, _P, _P]),
    'cfl_conv1x1_wgrad_workgroups': (c_int, [c_int]),
    'cfl_conv3x3_x3_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'cfl_conv3x3_x3_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    'cfl_conv3x3_x3_rot_weight': (c_int, [_P, c_int, c_int, _P, _P]),
    'cfl_conv3x3_x3_wimage_bytes': (c_size_t, [c_int, c_int]),
    'cfl_conv3x3_x3_wimage': (c_int, [_P, c_int, c_int, _P, _P]),
    'cfl_conv3x3_x3_wimage_rot': (c_int, [_P, c_int, c_int, _P, _P]),
    'cfl_conv3x3_x3_fwd_img': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    'cfl_conv3x3_x3_fwd_img_s2': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    'cfl_conv3x3_x3_wgrad_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'cfl_conv3x3_x3_wgrad_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'cfl_conv3x3_x3_wgrad': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'cfl_conv3x3_x3_wgrad_splits': (c_int, [c_int]),
    'cfl_conv3x3_wgrad_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'cfl_conv3x3_wgrad_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'cfl_conv3x3_wgrad': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'cfl_conv3x3_wgrad_splits': (c_int, [c_int]),
    'cfl_conv3x3_wgrad_debug': (c_int, [c_int]),
    'cfl_bn_bwd_wgrad_supported': (c_int, [c_longlong, c_int, c_int]),
    'cfl_bn_bwd_wgrad_ws_bytes': (c_size_t, [c_longlong, c_int, c_int]),
    'cfl_bn_bwd_wgrad': (c_int, [_P, _P, _P, c_int, _P, _P, _P, c_longlong, c_int, _P, _P, _P, _P, _P, _P]),
    'cfl_bn_ws_bytes': (c_size_t, [c_longlong, c_int]),
    'cfl_bn_fwd': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P]),
    'cfl_bn_fwd_pre': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, c_int, _P]),
    'cfl_bn_apply': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P, _P]),
    'cfl_bn_fwd_f32': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P]),
    'cfl_bn_apply_f32': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P, _P]),
    'cfl_bn_bwd_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'cfl_bn_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'cfl_bn_pool_fwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    'cfl_bn_pool_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'cfl_bn_pool_fwd_f32': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    'cfl_bn_pool_bwd_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'cfl_grad_clip_coef': (c_int, [_P, _P, c_int, c_float, _P, _P, _P]),
    'cfl_adamp_step': (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, _P, c_float, c_float, c_float, c_float, c_float,
                               c_float, c_float, c_int, c_int, _P, _P]),
    'cfl_adamp_step_counted': (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, _P, c_float, c_float, c_float, c_float, c_float,
                                       c_float, c_float, c_int, _P, _P, _P]),
}

_lib = None


class CreamflHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CreamflHipError(
            f'{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; '
            'run `make -C creamfl_amd/csrc` (hipcc --offload-arch=gfx950) first.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: 'CFL_EINVAL (bad size / null pointer)', -2: 'CFL_EALIGN', -3: 'CFL_ELIMIT (size out of range)'}
        raise CreamflHipError(f'{what} failed: {kind.get(rc, "hipError_t " + str(rc))}')


def kernel_names():
    lib = load()
    return [lib.cfl_kernel_name(i).decode() for i in range(lib.cfl_num_kernels())]


def prof_enable(on=True):
    check(load().cfl_prof_enable(1 if on else 0), 'cfl_prof_enable')


def prof_select(kernel_name=None):
    """Restrict event timing to one kernel (by name), or to all kernels with None."""
    kid = kernel_names().index(kernel_name) if kernel_name else -1
    check(load().cfl_prof_select(kid), 'cfl_prof_select')


def prof_reset():
    check(load().cfl_prof_reset(), 'cfl_prof_reset')


def prof_query():
    """{kernel_name: (launches, total_ms)} for every kernel launched since the last reset."""
    lib = load()
    out = {}
    for i, name in enumerate(kernel_names()):
        n, ms = c_longlong(0), c_double(0.0)
        check(lib.cfl_prof_query(i, ctypes.byref(n), ctypes.byref(ms)), 'cfl_prof_query')
        if n.value:
            out[name] = (n.value, ms.value)
    return out
