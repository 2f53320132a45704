"""ctypes binding of libpfn_hip.so (the C ABI declared in include/pfn_hip.h).

PyTorch-ROCm is only plumbing here: it owns device memory and streams; every arithmetic step of
the hot path runs in the hand-written gfx950 kernels behind this boundary.  There is NO fallback:
if the shared library is missing (or was built for another ABI version) every product entry
point raises, it never silently routes to PyTorch ops.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpfn_hip.so')
ABI_VERSION = 8

PREC_BF16 = 0
PREC_F32 = 1
PREC_FP16 = 2      # fp16 MFMA operands under a device-side loss scale (include/pfn_hip.h): the timed path that holds the north star's 1e-3
PRECISIONS = {'bf16': PREC_BF16, 'f32': PREC_F32, 'fp32': PREC_F32, 'fp16': PREC_FP16, 'f16': PREC_FP16}
SCHED_TOP_LAYER_ALL_ROWS, SCHED_FUSE_LN_WIDE, SCHED_SEPARATE_LNBWD, SCHED_DETERMINISTIC, SCHED_NO_KEY_CENTERING, SCHED_FUSE_Q_PROJECTION, SCHED_KEY_CENTERING, SCHED_F32_RESIDUAL = 1, 2, 4, 8, 16, 32, 64, 128     # pfn_model_desc.schedule bits (include/pfn_hip.h)

# GEMM epilogue flags (csrc/pfn_kernels.h)
EPI_BIAS, EPI_GELU, EPI_GELU_BWD, EPI_RESID, EPI_OUT_F32, EPI_OUT_T, EPI_OUT2_T, EPI_ACCUM, EPI_RESID_T = 1, 2, 4, 8, 16, 32, 64, 128, 256


class HipExtensionError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [('num_features', ctypes.c_int32), ('emsize', ctypes.c_int32), ('nhead', ctypes.c_int32),
                ('nhid', ctypes.c_int32), ('nlayers', ctypes.c_int32), ('n_out', ctypes.c_int32),
                ('precision', ctypes.c_int32), ('ln_eps', ctypes.c_float), ('dropout', ctypes.c_float), ('schedule', ctypes.c_int32)]

    def key(self):
        return (self.num_features, self.emsize, self.nhead, self.nhid, self.nlayers, self.n_out, self.precision, self.ln_eps, self.dropout, self.schedule)


HOST_CALLBACK = ctypes.CFUNCTYPE(None, ctypes.c_void_p)     # pfn_host_callback

_lib = None

_c = ctypes
_P, _I, _L, _F, _U64 = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_uint64
_D = _c.POINTER(ModelDesc)

# name -> (restype, argtypes).  This table is the Python mirror of include/pfn_hip.h; the not-gpu
# test suite checks that the library exports every one of these symbols.
SIGNATURES = {
    'pfn_abi_version': (_I, []),
    'pfn_last_error_string': (_c.c_char_p, []),
    'pfn_set_tuning': (_I, [_I, _I]),
    'pfn_default_schedule': (_I, []),
    'pfn_profile_enable': (_I, [_I]),
    'pfn_profile_read': (_I, [_I, _c.POINTER(_c.c_double), _c.POINTER(_L)]),
    'pfn_param_layout': (_I, [_D, _c.POINTER(_L), _c.POINTER(_L), _I]),
    'pfn_param_count': (_L, [_D]),
    'pfn_shadow_bytes': (_L, [_D]),
    'pfn_prepare_params': (_I, [_D, _P, _P, _P]),
    'pfn_workspace_bytes': (_L, [_D, _I, _I]),
    'pfn_top_layer_rows': (_L, [_D, _I, _I, _I, _I]),
    'pfn_stack_forward': (_I, [_D, _P, _P, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _P, _L, _P, _P]),
    'pfn_stack_backward': (_I, [_D, _P, _P, _P, _L, _L, _P, _L, _L, _I, _I, _I, _P, _L, _P, _P, _P, _P]),
    'pfn_stack_forward_dropout': (_I, [_D, _P, _P, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _P, _L, _P, _P, _U64]),
    'pfn_stack_backward_split': (_I, [_D, _P, _P, _P, _L, _L, _P, _L, _L, _I, _I, _I, _P, _L, _P, _P, _P, _P, _I, _P, _P, _I, _U64]),
    # (d, params, shadow, x, x_st, x_sb, y, y_st, y_sb, B, S, sep_of, row_off, sep_min, sep_max, test_rows, ws, ws_bytes, logits, stream, use_dropout, seed)
    'pfn_stack_forward_ragged': (_I, [_D, _P, _P, _P, _L, _L, _P, _L, _L, _I, _I, _P, _P, _I, _I, _L, _P, _L, _P, _P, _I, _U64]),
    # (..., ws, ws_bytes, dlogits, grads, stream, first_group_layers, callback, user, use_dropout, seed)
    'pfn_stack_backward_ragged': (_I, [_D, _P, _P, _P, _L, _L, _P, _L, _L, _I, _I, _P, _P, _I, _I, _L, _P, _L, _P, _P, _P, _I, _P, _P, _I, _U64]),
    'pfn_bar_nll_forward': (_I, [_P, _L, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    'pfn_bar_nll_backward': (_I, [_P, _L, _P, _P, _P, _L, _I, _P, _P]),
    'pfn_bar_mean': (_I, [_P, _L, _P, _L, _I, _I, _P, _P]),
    'pfn_clip_adam_step': (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _I, _I, _P, _P]),
    'pfn_gp_workspace_bytes': (_L, [_I, _I]),
    'pfn_gp_prior_sample': (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _U64, _U64, _P, _P]),
    'pfn_gp_posterior': (_I, [_P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'pfn_mlp_prior_forward': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _U64, _U64, _P]),
    'pfn_op_gemm_nt': (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _P, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _I, _P]),
    'pfn_op_gemm_tn': (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P]),
    'pfn_op_gemm_tn_group': (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'pfn_op_gemm_ln': (_I, [_P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _P]),
    'pfn_op_gemm_lnbwd': (_I, [_P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'pfn_op_attention_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'pfn_op_attention_bwd_ws_bytes': (_L, [_I, _I, _I, _I]),
    'pfn_op_attention_fwd_from': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pfn_op_attention_bwd_from': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pfn_op_gather_rows': (_I, [_P, _P, _I, _I, _L, _I, _P]),
    'pfn_op_scatter_rows': (_I, [_P, _P, _I, _I, _L, _I, _I, _P]),
    'pfn_op_attention_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pfn_op_layernorm_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P]),
    'pfn_op_layernorm_bwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'pfn_op_cast': (_I, [_P, _P, _L, _I, _P]),
    'pfn_op_qkv_projection': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
}


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libpfn_hip.so (in-tree). hipcc cross-compiles without a GPU."""
    script = os.path.join(_HERE, 'csrc', 'build.sh')
    res = subprocess.run(['bash', script], capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise HipExtensionError('building libpfn_hip.so failed:\n' + res.stdout[-4000:] + res.stderr[-8000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def lib():
    """The loaded library (import torch first so libamdhip64 is already resident)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f'{LIB_PATH} not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or transformerscandobayesianinference_amd/csrc/build.sh). There is no CPU / PyTorch fallback for the hot path.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise HipExtensionError(f'libpfn_hip.so does not export {name}') from e
            fn.restype = res
            fn.argtypes = args
        if handle.pfn_abi_version() != ABI_VERSION:
            raise HipExtensionError(f'libpfn_hip.so ABI {handle.pfn_abi_version()} != binding ABI {ABI_VERSION}; rebuild')
        _lib = handle
    return _lib


def check(rc, what):
    if rc < 0:
        msg = lib().pfn_last_error_string()
        raise HipExtensionError(f'{what} failed with code {rc}: {msg.decode() if msg else ""}')
    return rc


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_gpu_tensor(t, name):
    if not t.is_cuda:
        raise HipExtensionError(
            f'{name} lives on {t.device}: the PFN hot path only runs on an MI355X through libpfn_hip.so; '
            f'move the model and data to a cuda device (there is no CPU fallback in the product path).')
