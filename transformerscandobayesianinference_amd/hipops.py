"""Tensor-level wrappers around the single-op C-ABI entry points (`pfn_op_*`, include/pfn_hip.h): one kernel launch per call on the
current stream.  Used by the per-kernel parity tests (tests/test_gpu_ops.py), by bench.py's kernel table and by tools/."""
import torch

from transformerscandobayesianinference_amd import _hip

TDT = {_hip.PREC_BF16: torch.bfloat16, _hip.PREC_F32: torch.float32, _hip.PREC_FP16: torch.float16}
PREC_OF = {v: k for k, v in TDT.items()}      # operand precision of a tensor's dtype
MANGLED_OPERAND = {_hip.PREC_BF16: 'DF16b', _hip.PREC_FP16: 'DF16_', _hip.PREC_F32: 'f'}      # Itanium mangling of the kernels' operand-type template argument


def sp():
    return _hip.stream_ptr()


def gemm_nt(A, B, flags, prec, bias=None, aux=None, resid=None, out_f32=None, out_t=None, out2_t=None):
    M, K = A.shape
    N = B.shape[0]
    p = _hip.ptr
    ld = lambda t: 0 if t is None else t.stride(0)
    _hip.check(_hip.lib().pfn_op_gemm_nt(p(A), A.stride(0), p(B), B.stride(0), M, N, K, flags, p(bias), p(aux), ld(aux),
                                         p(resid), ld(resid), p(out_f32), ld(out_f32), p(out_t), ld(out_t), p(out2_t), ld(out2_t),
                                         prec, sp()), 'pfn_op_gemm_nt')


def gemm_tn(A, B, C, prec, atomic=1):
    M, P = A.shape
    Q = B.shape[1]
    _hip.check(_hip.lib().pfn_op_gemm_tn(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), M, P, Q,
                                         atomic, prec, sp()), 'pfn_op_gemm_tn')


def gemm_tn_group(problems, splits=0):
    """problems: list of (A[M,P], B[M,Q], C[P,Q], colsum[P] or None), all operands of ONE 16-bit dtype (bf16 / fp16) with the same M."""
    import ctypes
    n = len(problems)
    M = problems[0][0].shape[0]
    VP, L, I = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
    A = VP(*[p[0].data_ptr() for p in problems]); lda = L(*[p[0].stride(0) for p in problems])
    B = VP(*[p[1].data_ptr() for p in problems]); ldb = L(*[p[1].stride(0) for p in problems])
    C = VP(*[p[2].data_ptr() for p in problems]); ldc = L(*[p[2].stride(0) for p in problems])
    P = I(*[p[0].shape[1] for p in problems]); Q = I(*[p[1].shape[1] for p in problems])
    cs = VP(*[(p[3].data_ptr() if p[3] is not None else None) for p in problems])
    _hip.check(_hip.lib().pfn_op_gemm_tn_group(n, A, lda, B, ldb, C, ldc, P, Q, cs, M, splits, PREC_OF[problems[0][0].dtype], sp()), 'pfn_op_gemm_tn_group')


SUMS_16BIT = 256      # include/pfn_hip.h PFN_OP_SUMS_16BIT


def gemm_ln(A, B, bias, gamma, beta, eps, resid=None, prev=None, out=None, sums16=False):
    """prev = (ry, rmean, rrstd, rgamma, rbeta) when the residual is the previous LayerNorm's (recomputed) output.
    out = (y[M+2,N], x_t, mean, rstd) pre-allocated buffers (timing loops).
    sums16 (fp16 operands): y leaves -- and prev's ry arrives -- in operand precision (PFN_OP_SUMS_16BIT: what the stack does for fp16 models)."""
    M, K = A.shape
    N = B.shape[0]
    dev = A.device
    if out is None:
        y = torch.full((M + 2, N), float('nan'), dtype=A.dtype if sums16 else torch.float32, device=dev)
        x_t = torch.empty(M, N, dtype=A.dtype, device=dev)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    else:
        y, x_t, mean, rstd = out
    p = _hip.ptr
    pv = prev if prev is not None else (None,) * 5
    assert y.dtype == (A.dtype if sums16 else torch.float32) and (pv[0] is None or pv[0].dtype == y.dtype)
    _hip.check(_hip.lib().pfn_op_gemm_ln(p(A), A.stride(0), p(B), B.stride(0), M, N, K, p(bias), p(resid), p(pv[0]), p(pv[1]), p(pv[2]), p(pv[3]), p(pv[4]),
                                         p(gamma), p(beta), eps, p(y), p(mean), p(rstd), p(x_t), PREC_OF[A.dtype] | (SUMS_16BIT if sums16 else 0), sp()), 'pfn_op_gemm_ln')
    if out is None:
        assert torch.isnan(y[M:]).all()
    return y[:M], x_t, mean, rstd


def attention_fwd(qkv, H, sep, prec, q_begin=0, out=None):
    """q_begin > 0: the queries below it (rounded down to a multiple of 256) are skipped -- their ctx / lse rows keep the NaN fill
    (or whatever `out` = (ctx, lse) held: timing loops pass buffers so that no fill kernel runs between the launches)."""
    B, S, E3 = qkv.shape
    E = E3 // 3
    if q_begin:
        if out is not None:
            ctx, lse = out
        else:
            ctx = torch.full((B, S, E), float('nan'), dtype=qkv.dtype, device=qkv.device)
            lse = torch.full((B, H, S), float('nan'), dtype=torch.float32, device=qkv.device)
        _hip.check(_hip.lib().pfn_op_attention_fwd_from(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, S, E, H, sep, q_begin, prec, sp()), 'attn fwd')
        return ctx, lse
    ctx = torch.empty(B, S, E, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=qkv.device)
    _hip.check(_hip.lib().pfn_op_attention_fwd(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, S, E, H, sep, prec, sp()), 'attn fwd')
    return ctx, lse


def gather_rows(src, sep, out=None):
    """[B, S, W] -> the compact test rows [(S - sep) * B, W] in the decoder's order (row (t - sep) * B + b)."""
    B, S, W = src.shape
    dst = out if out is not None else torch.empty((S - sep) * B, W, dtype=src.dtype, device=src.device)
    _hip.check(_hip.lib().pfn_op_gather_rows(src.data_ptr(), dst.data_ptr(), B, S, W * src.element_size(), sep, sp()), 'gather rows')
    return dst


def scatter_rows(src, B, S, sep, zero_from, fill=float('nan'), out=None):
    """the inverse: rows >= sep from the compact rows, zeros in [zero_from, sep), `fill` (untouched) below"""
    W = src.shape[1]
    dst = out if out is not None else torch.full((B, S, W), fill, dtype=src.dtype, device=src.device)
    _hip.check(_hip.lib().pfn_op_scatter_rows(src.data_ptr(), dst.data_ptr(), B, S, W * src.element_size(), sep, zero_from, sp()), 'scatter rows')
    return dst


def gemm_lnbwd(A, B, aux, y, mean, rstd, gamma, out=None):
    """dx_t, dgamma, dbeta of  v = A . B^T + aux  pushed back through the LayerNorm (y, mean, rstd, gamma)  (pfn_op_gemm_lnbwd).
    y in operand precision (fp16) selects PFN_OP_SUMS_16BIT."""
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = (torch.full((M + 2, N), float('nan'), dtype=A.dtype, device=A.device), torch.zeros(N, device=A.device), torch.zeros(N, device=A.device))
    dx_t, dgamma, dbeta = out
    p = _hip.ptr
    _hip.check(_hip.lib().pfn_op_gemm_lnbwd(p(A), A.stride(0), p(B), B.stride(0), M, N, K, p(aux), p(y), p(mean), p(rstd), p(gamma),
                                            p(dx_t), p(dgamma), p(dbeta), PREC_OF[A.dtype] | (SUMS_16BIT if y.dtype == A.dtype else 0), sp()), 'pfn_op_gemm_lnbwd')
    return dx_t, dgamma, dbeta


# (name, rocprofv3 kernel name, `parts` bit, algorithmic product units, executed product units) of every launch of the attention
# backward; one unit = one [S x keys x head-dim] product over all heads (the backward's algorithmic work is 4: dV, dP, dK, dQ;
# it executes 5, the forward's S being recomputed once)
ATTENTION_BWD_PARTS = [
    # (rocprofv3 prints these kernels by their mangled names: its demangler does not know the __bf16 template argument)
    ('attn_bwd: delta = rowsum(dO * O)', '_ZN3pfn17attn_delta_kernelI{T}EEvNS_8AttnArgsEi', 1, 0.0, 0.0),
    ('attn_bwd: key-block pass (S, dP, dV, dK products; stores dS^T)', '_ZN3pfn18attn_bwd_kv_kernelI{T}Li{D}ELi0ELb0', 2, 3.0, 4.0),
    ('attn_bwd: query-block pass (dQ = dS K from the stored dS^T)', '_ZN3pfn18attn_bwd_dq_kernelI{T}Li{D}ELb0', 4, 1.0, 1.0),      # (...ELb0: the variant without the dropout masks)
]
ATTENTION_FWD_ROCPROF = '_ZN3pfn15attn_fwd_kernelI{T}Li{D}ELb0'
# (round 2: head dim 256 ran the key-block pass as two launches with one more S product -- {256: 5.0}; round 3: one pass everywhere)
ATTENTION_BWD_KV_EXECUTED_UNITS = {}
_bwd_scratch = {}


def attention_bwd(qkv, ctx, lse, dctx, H, sep, prec, parts=0, q_begin=0):
    """parts = 0: the whole backward (outputs start as NaN so a skipped element shows).  parts != 0 (timing): only the selected
    launches, into cached scratch buffers (the skipped launches' products must exist from an earlier full call with the same
    shapes for the numbers to mean anything)."""
    B, S, E3 = qkv.shape
    E = E3 // 3
    key = (tuple(qkv.shape), qkv.dtype, qkv.device, H)
    if key not in _bwd_scratch:
        _bwd_scratch.clear()
        ws = _hip.check(_hip.lib().pfn_op_attention_bwd_ws_bytes(B, S, H, prec), 'attn bwd ws')
        _bwd_scratch[key] = (torch.zeros_like(qkv), torch.zeros(2, B, H, S, dtype=torch.float32, device=qkv.device),     # [delta | lse in log2 units]
                             torch.empty(ws, dtype=torch.uint8, device=qkv.device))
    dqkv, delta, ds = _bwd_scratch[key]
    if not parts:
        dqkv = torch.full_like(qkv, float('nan'))
        ds.fill_(0xff)                                 # NaN patterns: a dS^T element the key-block pass skipped would show in dQ
    if q_begin:
        _hip.check(_hip.lib().pfn_op_attention_bwd_from(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dqkv.data_ptr(),
                                                        delta.data_ptr(), ds.data_ptr(), B, S, E, H, sep, q_begin, prec, parts, sp()), 'attn bwd')
        return dqkv
    _hip.check(_hip.lib().pfn_op_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dqkv.data_ptr(),
                                               delta.data_ptr(), ds.data_ptr(), B, S, E, H, sep, prec, parts, sp()), 'attn bwd')
    return dqkv


def layernorm_fwd(x, gamma, beta, eps, prec):
    rows, E = x.shape
    y32 = torch.empty_like(x)
    yt = torch.empty(rows, E, dtype=TDT[prec], device=x.device)
    mean = torch.empty(rows, device=x.device)
    rstd = torch.empty(rows, device=x.device)
    _hip.check(_hip.lib().pfn_op_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y32.data_ptr(), yt.data_ptr(),
                                               mean.data_ptr(), rstd.data_ptr(), rows, E, eps, prec, sp()), 'ln fwd')
    return y32, yt, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, prec, want_f32=True):
    """dy: f32, or the operand dtype of `prec` (the form the backward schedule uses; then usually want_f32=False)."""
    rows, E = x.shape
    dy_is_t = int(dy.dtype != torch.float32)
    dx32 = torch.empty_like(x) if want_f32 else None
    dxt = torch.empty(rows, E, dtype=TDT[prec], device=x.device)
    dg = torch.zeros(E, device=x.device)
    db = torch.zeros(E, device=x.device)
    dbias = torch.zeros(E, device=x.device)
    _hip.check(_hip.lib().pfn_op_layernorm_bwd(dy.data_ptr(), dy_is_t, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               dx32.data_ptr() if want_f32 else 0, dxt.data_ptr(), dg.data_ptr(), db.data_ptr(), dbias.data_ptr(),
                                               rows, E, prec, sp()), 'ln bwd')
    return dx32, dxt, dg, db, dbias


def qkv_projection(x, w_in, b_in, sep, center=True, sep_of=None):
    """x [B, S, E], w_in [3E, E] (one 16-bit dtype), b_in [3E] f32 -> (qkv [B, S, 3E], kshift [B, E] f32): the encoder layer's packed projection with the keys of every
    dataset centred on a sample mean of its train rows (pfn_op_qkv_projection; center=False: the plain projection)."""
    B, S, E = x.shape
    qkv = torch.full((B, S, 3 * E), float('nan'), dtype=x.dtype, device=x.device)
    ks = torch.full((B, E), float('nan'), dtype=torch.float32, device=x.device)
    _hip.check(_hip.lib().pfn_op_qkv_projection(x.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), qkv.data_ptr(), ks.data_ptr(), B, S, E, sep,
                                                _hip.ptr(sep_of), int(center), PREC_OF[x.dtype], sp()), 'pfn_op_qkv_projection')
    return qkv, ks
