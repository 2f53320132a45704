"""GPU unit tests of the individual HIP kernels against plain PyTorch references of the same op.

bf16 cases are compared against an f64 reference computed from the SAME bf16-rounded inputs, so
the tolerance only has to cover accumulation-order effects and the bf16 rounding of the outputs;
the exact-f32 MFMA path is held to f32 round-off.
"""
import math

import pytest
import torch

from transformerscandobayesianinference_amd import _hip
from transformerscandobayesianinference_amd import hipops

pytestmark = pytest.mark.gpu
BF, F32, FP16 = _hip.PREC_BF16, _hip.PREC_F32, _hip.PREC_FP16
PRECS = [pytest.param(BF, id='bf16'), pytest.param(FP16, id='fp16'), pytest.param(F32, id='f32')]


def tol(prec, bf16, fp16, f32):
    """bound by operand format: bf16 rounds to 8 significand bits, fp16 to 11 (the bounds sit a factor 8 apart), f32 MFMA is exact f32 arithmetic"""
    return {BF: bf16, FP16: fp16, F32: f32}[prec]


@pytest.fixture(params=[BF, FP16], ids=['bf16', 'fp16'])
def op16(request):
    """the two 16-bit operand formats of the LDS-DMA kernels (same kernels, same tiles; v_mfma_f32_32x32x16_bf16 / _f16)"""
    return request.param


def dev():
    return torch.device('cuda:0')


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=None):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xffff))
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


def relerr(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def maxerr(a, b):
    return (a.double() - b.double()).abs().max().item()


GEMM_SHAPES = [(128, 128, 64), (256, 512, 512), (300, 200, 136), (4000, 1536, 512), (77, 1000, 1024), (1024, 8, 200), (130, 1, 64)]


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
def test_gemm_nt_plain(M, N, K, prec):
    dt = hipops.TDT[prec]
    A, B = rnd(M, K, dtype=dt, seed=1), rnd(N, K, dtype=dt, seed=2)
    out = torch.full((M, N), float('nan'), device=dev())
    hipops.gemm_nt(A, B, _hip.EPI_OUT_F32, prec, out_f32=out)
    ref = A.double() @ B.double().t()
    tol = 1e-5
    assert relerr(out, ref) < tol, relerr(out, ref)


@pytest.mark.parametrize('prec', PRECS)
def test_gemm_nt_asymmetric_identity(prec):
    """A = I against an asymmetric B catches a transposed output layout (guide G9)."""
    dt = hipops.TDT[prec]
    A = torch.eye(128, device=dev()).to(dt)
    B = (torch.arange(256 * 128, device=dev()).float().view(256, 128) % 251 / 16).to(dt)
    out = torch.empty(128, 256, device=dev())
    hipops.gemm_nt(A, B, _hip.EPI_OUT_F32, prec, out_f32=out)
    assert torch.equal(out, B.float().t())


def gelu_grad(x):
    x = x.double()
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize('prec', PRECS)
def test_gemm_nt_epilogues(prec):
    dt = hipops.TDT[prec]
    M, N, K = 500, 264, 192
    A, B = rnd(M, K, dtype=dt, seed=3), rnd(N, K, dtype=dt, seed=4, scale=0.1)
    bias, resid = rnd(N, seed=5), rnd(M, N, seed=6)
    aux = rnd(M, N, dtype=dt, seed=7)
    base = A.double() @ B.double().t()
    tol_t = tol(prec, 4e-3, 5e-4, 1e-6)
    # bias + gelu, both outputs
    out_t = torch.empty(M, N, dtype=dt, device=dev()); out2 = torch.empty_like(out_t)
    hipops.gemm_nt(A, B, _hip.EPI_BIAS | _hip.EPI_GELU | _hip.EPI_OUT_T | _hip.EPI_OUT2_T, prec, bias=bias, out_t=out_t, out2_t=out2)
    pre = base + bias.double()
    assert relerr(out2, gelu_grad(pre)) < tol_t          # the second output of a GELU GEMM is gelu'(pre-activation)
    assert relerr(out_t, torch.nn.functional.gelu(pre)) < tol_t
    # bias + residual -> f32
    out = torch.empty(M, N, device=dev())
    hipops.gemm_nt(A, B, _hip.EPI_BIAS | _hip.EPI_RESID | _hip.EPI_OUT_F32, prec, bias=bias, resid=resid, out_f32=out)
    assert relerr(out, pre + resid.double()) < 2e-6
    # gelu backward multiply
    hipops.gemm_nt(A, B, _hip.EPI_GELU_BWD | _hip.EPI_OUT_T, prec, aux=aux, out_t=out_t)      # multiplies by the stored derivative
    assert relerr(out_t, base * aux.double()) < tol_t
    # accumulate
    out.fill_(1.0)
    hipops.gemm_nt(A, B, _hip.EPI_OUT_F32 | _hip.EPI_ACCUM, prec, out_f32=out)
    assert relerr(out, base + 1.0) < 2e-6


@pytest.fixture(params=[2, 3], ids=['256x256', '128x256'])
def big_gemm(request):
    """Force one of the LDS-DMA NT kernels (pfn_set_tuning) for the duration of a test."""
    _hip.check(_hip.lib().pfn_set_tuning(0, request.param), 'pfn_set_tuning')
    yield
    _hip.check(_hip.lib().pfn_set_tuning(0, 0), 'pfn_set_tuning')


@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (256, 512, 512), (300, 200, 128), (4000, 1536, 512), (77, 1000, 1024), (1023, 516, 192), (2048, 512, 1536)])
def test_gemm_nt_big_plain(M, N, K, big_gemm, op16):
    A, B = rnd(M, K, dtype=hipops.TDT[op16], seed=1), rnd(N, K, dtype=hipops.TDT[op16], seed=2)
    out = torch.full((M + 3, N), float('nan'), device=dev())
    hipops.gemm_nt(A, B, _hip.EPI_OUT_F32, op16, out_f32=out[:M])
    ref = A.double() @ B.double().t()
    assert relerr(out[:M], ref) < 1e-5, relerr(out[:M], ref)
    assert torch.isnan(out[M:]).all()          # nothing written past the last row
    # same stage order and MFMA shape as the 128x128 kernel: the two must agree to accumulation order
    _hip.lib().pfn_set_tuning(0, 1)
    out_small = torch.empty(M, N, device=dev())
    hipops.gemm_nt(A, B, _hip.EPI_OUT_F32, op16, out_f32=out_small)
    assert relerr(out[:M], out_small) < 1e-6


def test_gemm_nt_big_asymmetric_identity(big_gemm, op16):
    A = torch.eye(256, device=dev()).to(hipops.TDT[op16])
    B = (torch.arange(512 * 256, device=dev()).float().view(512, 256) % 251 / 16).to(hipops.TDT[op16])
    out = torch.empty(256, 512, device=dev())
    hipops.gemm_nt(A, B, _hip.EPI_OUT_F32, op16, out_f32=out)
    assert torch.equal(out, B.float().t())


def test_gemm_nt_big_epilogues(big_gemm, op16):
    dt = hipops.TDT[op16]
    M, N, K = 700, 520, 192
    A, B = rnd(M, K, dtype=dt, seed=3), rnd(N, K, dtype=dt, seed=4, scale=0.1)
    bias, resid = rnd(N, seed=5), rnd(M, N, seed=6)
    aux = rnd(M, N, dtype=dt, seed=7)
    base = A.double() @ B.double().t()
    tol_t = tol(op16, 4e-3, 5e-4, 0)
    out_t = torch.empty(M, N, dtype=dt, device=dev()); out2 = torch.empty_like(out_t)
    hipops.gemm_nt(A, B, _hip.EPI_BIAS | _hip.EPI_GELU | _hip.EPI_OUT_T | _hip.EPI_OUT2_T, op16, bias=bias, out_t=out_t, out2_t=out2)
    pre = base + bias.double()
    assert relerr(out2, gelu_grad(pre)) < tol_t
    assert relerr(out_t, torch.nn.functional.gelu(pre)) < tol_t
    out = torch.empty(M, N, device=dev())
    hipops.gemm_nt(A, B, _hip.EPI_BIAS | _hip.EPI_RESID | _hip.EPI_OUT_F32, op16, bias=bias, resid=resid, out_f32=out)
    assert relerr(out, pre + resid.double()) < 1e-5
    hipops.gemm_nt(A, B, _hip.EPI_RESID | _hip.EPI_OUT_F32, op16, resid=resid, out_f32=out)
    assert relerr(out, base + resid.double()) < 1e-5
    hipops.gemm_nt(A, B, _hip.EPI_RESID_T | _hip.EPI_OUT_F32, op16, aux=aux, out_f32=out)
    assert relerr(out, base + aux.double()) < 1e-5
    hipops.gemm_nt(A, B, _hip.EPI_BIAS | _hip.EPI_OUT_T, op16, bias=bias, out_t=out_t)
    assert relerr(out_t, pre) < tol_t
    hipops.gemm_nt(A, B, _hip.EPI_OUT_T, op16, out_t=out_t)
    assert relerr(out_t, base) < tol_t
    hipops.gemm_nt(A, B, _hip.EPI_GELU_BWD | _hip.EPI_OUT_T, op16, aux=aux, out_t=out_t)
    assert relerr(out_t, base * aux.double()) < tol_t


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('M,P,Q', [(64, 128, 128), (1000, 512, 1024), (4096, 1000, 200), (333, 1, 72), (16000, 1536, 512)])
def test_gemm_tn(M, P, Q, prec):
    dt = hipops.TDT[prec]
    # A may carry a padded leading dimension (decoder dlogits)
    ldp = (P + 7) // 8 * 8
    Afull = torch.zeros(M, ldp, dtype=dt, device=dev())
    Afull[:, :P] = rnd(M, P, dtype=dt, seed=8)
    A = Afull[:, :P]
    B = rnd(M, Q, dtype=dt, seed=9)
    C = torch.ones(P, Q, device=dev())
    hipops.gemm_tn(A, B, C, prec)
    ref = A.double().t() @ B.double() + 1.0
    assert relerr(C, ref) < 3e-6, relerr(C, ref)


@pytest.mark.parametrize('gemm_mode', [0, 1], ids=['lds-dma-kernel', '128x128-kernel'])
@pytest.mark.parametrize('B,S,E,seps', [(3, 700, 256, [600, 600, 600]), (2, 2000, 512, [1755, 1755]), (4, 300, 128, [257, 0, 1, 40]), (2, 130, 64, [130, 5])])
def test_qkv_projection_with_centred_keys(B, S, E, seps, gemm_mode, op16):
    """Key centring (pfn_op_qkv_projection, csrc/pfn_kernels.h launch_key_shift): the K block leaves the projection minus W_k . xbar[b], xbar the mean of <= 64
    evenly spaced TRAIN rows of the dataset; q and v are the plain projection; a dataset without train rows is not shifted; per-dataset eval positions (ragged
    batch) and the uniform one agree; the shifted keys give the SAME attention output (softmax is invariant to one vector subtracted from every key)."""
    dt = hipops.TDT[op16]
    x = rnd(B, S, E, dtype=dt, seed=60) + 1.5            # a common component, as in a trained model
    w = rnd(3 * E, E, dtype=dt, seed=61, scale=0.1)
    b = rnd(3 * E, seed=62)
    uniform = len(set(seps)) == 1
    sep_of = None if uniform else torch.tensor(seps, dtype=torch.int32, device=dev())
    _hip.check(_hip.lib().pfn_set_tuning(0, gemm_mode), 'pfn_set_tuning')
    try:
        plain, _ = hipops.qkv_projection(x, w, b, max(seps), center=False)
        got, ks = hipops.qkv_projection(x, w, b, max(seps), center=True, sep_of=sep_of)
    finally:
        _hip.lib().pfn_set_tuning(0, 0)
    assert not torch.isnan(got.float()).any() and not torch.isnan(ks).any()
    ref = x.double() @ w.double().t() + b.double()
    tol_t = tol(op16, 4e-3, 5e-4, 0)
    assert relerr(plain, ref) < tol_t
    want_ks = torch.zeros(B, E, dtype=torch.float64, device=dev())
    for i, sep in enumerate(seps):
        if sep > 0:
            ns = min(64, sep); st = sep // ns
            want_ks[i] = x[i, 0:ns * st:st].double().mean(0) @ w[E:2 * E].double().t()
    assert relerr(ks, want_ks) < 1e-5, relerr(ks, want_ks)
    ref_c = ref.clone()
    ref_c[:, :, E:2 * E] -= want_ks[:, None, :]
    assert torch.equal(got[:, :, :E], plain[:, :, :E]) and torch.equal(got[:, :, 2 * E:], plain[:, :, 2 * E:])      # q and v untouched
    # relative to the CENTRED keys' own norm -- the point of the exercise: the rounding is now relative to the part of k that differs between keys
    assert relerr(got[:, :, E:2 * E], ref_c[:, :, E:2 * E]) < tol_t
    if uniform:
        H = max(1, E // 64)
        ctx_p, lse_p = hipops.attention_fwd(plain, H, seps[0], op16)
        ctx_c, lse_c = hipops.attention_fwd(got, H, seps[0], op16)
        assert relerr(ctx_c, ctx_p) < 2.5 * tol_t


@pytest.fixture(params=[0, 1], ids=['128-row-tiles', '64-row-tiles'])
def ln_tile_rows(request):
    """PFN_TUNE_GEMM_LN_ROWS: the LayerNorm-fused GEMMs at N = 512 on 128-row tiles (default) or on 64-row tiles, two workgroups per CU."""
    _hip.check(_hip.lib().pfn_set_tuning(7, request.param), 'pfn_set_tuning')
    yield request.param
    _hip.check(_hip.lib().pfn_set_tuning(7, 0), 'pfn_set_tuning')


@pytest.mark.parametrize('M,N,K', [(128, 512, 512), (1000, 512, 1024), (333, 128, 96), (2050, 256, 64), (77, 512, 32), (300, 1024, 1024), (130, 1024, 2048), (64, 1024, 96), (1, 1024, 64)])
def test_gemm_ln_fused(M, N, K, ln_tile_rows, op16):
    """linear + bias + residual + LayerNorm in one kernel, both residual forms, against f64 PyTorch."""
    if ln_tile_rows and N != 512:
        pytest.skip('the 64-row tiles exist at N = 512 only')
    dt = hipops.TDT[op16]
    tol_t = tol(op16, 4e-3, 5e-4, 0)
    A, B = rnd(M, K, dtype=dt, seed=40), rnd(N, K, dtype=dt, seed=41, scale=0.1)
    bias, gamma, beta = rnd(N, seed=42), rnd(N, seed=43) + 1, rnd(N, seed=44)
    resid = rnd(M, N, seed=45)
    v = A.double() @ B.double().t() + bias.double() + resid.double()
    y, x_t, mean, rstd = hipops.gemm_ln(A, B, bias, gamma, beta, 1e-5, resid=resid)
    ref = torch.nn.functional.layer_norm(v, (N,), gamma.double(), beta.double(), 1e-5)
    assert relerr(y, v) < 1e-5 and relerr(x_t, ref) < tol_t
    assert maxerr(mean, v.mean(1)) < 1e-5 and relerr(rstd, 1 / torch.sqrt(v.var(1, unbiased=False) + 1e-5)) < 1e-5
    # residual = previous LayerNorm output, recomputed from its pre-LN sums and statistics
    ry = rnd(M, N, seed=46) * 2 + 0.3
    rg, rb = rnd(N, seed=47) + 1, rnd(N, seed=48)
    rmean, rvar = ry.double().mean(1), ry.double().var(1, unbiased=False)
    rrstd = 1 / torch.sqrt(rvar + 1e-5)
    r2 = (ry.double() - rmean[:, None]) * rrstd[:, None] * rg.double() + rb.double()
    v2 = A.double() @ B.double().t() + bias.double() + r2
    y2, x2, _, _ = hipops.gemm_ln(A, B, bias, gamma, beta, 1e-5, prev=(ry, rmean.float(), rrstd.float(), rg, rb))
    assert relerr(y2, v2) < 1e-5
    assert relerr(x2, torch.nn.functional.layer_norm(v2, (N,), gamma.double(), beta.double(), 1e-5)) < tol_t


@pytest.mark.parametrize('M,N,K', [(128, 512, 512), (1000, 512, 1024), (333, 128, 96), (2050, 256, 64), (77, 512, 1536),
                                   (200, 1024, 2048), (77, 1024, 3072), (64, 1024, 32)])      # N = 1024: the 64-row tiles (emsize 1024)
def test_gemm_lnbwd_fused(M, N, K, ln_tile_rows, op16):
    """dgrad GEMM + residual-branch gradient + LayerNorm backward in one kernel, against f64 autograd of the LayerNorm."""
    if ln_tile_rows and N != 512:
        pytest.skip('the 64-row tiles exist at N = 512 only')
    dt = hipops.TDT[op16]
    A, B = rnd(M, K, dtype=dt, seed=50), rnd(N, K, dtype=dt, seed=51, scale=0.1)
    aux = rnd(M, N, dtype=dt, seed=52)
    y = rnd(M, N, seed=53) * 2 + 0.3
    gamma = rnd(N, seed=54) + 1
    yd = y.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = torch.zeros(N, dtype=torch.float64, device=dev(), requires_grad=True)
    v = A.double() @ B.double().t() + aux.double()
    torch.nn.functional.layer_norm(yd, (N,), gd, bd, 1e-5).backward(v)
    mean, var = y.double().mean(1), y.double().var(1, unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    dx_t, dgamma, dbeta = hipops.gemm_lnbwd(A, B, aux, y, mean.float(), rstd.float(), gamma)
    assert torch.isnan(dx_t[M:].float()).all()            # nothing written past row M
    assert relerr(dx_t[:M], yd.grad) < tol(op16, 4e-3, 5e-4, 0), relerr(dx_t[:M], yd.grad)
    # (xhat waits in LDS in operand precision between its two uses: that rounding is all the dgamma sum sees)
    assert relerr(dgamma, gd.grad) < tol(op16, 2e-3, 2.5e-4, 0) and relerr(dbeta, bd.grad) < 1e-4, (relerr(dgamma, gd.grad), relerr(dbeta, bd.grad))
    # accumulation: a second launch doubles the parameter gradients
    hipops.gemm_lnbwd(A, B, aux, y, mean.float(), rstd.float(), gamma, out=(dx_t, dgamma, dbeta))
    assert relerr(dgamma, 2 * gd.grad) < tol(op16, 2e-3, 2.5e-4, 0) and relerr(dbeta, 2 * bd.grad) < 1e-4


@pytest.mark.parametrize('M,N,K', [(128, 512, 512), (1000, 512, 1024), (333, 128, 96), (2050, 256, 64), (300, 1024, 1024), (1, 1024, 64)])
def test_gemm_ln_and_lnbwd_with_sums_in_operand_precision(M, N, K, ln_tile_rows):
    """PFN_OP_SUMS_16BIT (fp16 operands): the LayerNorm-fused GEMM writes its pre-LN sums, and reads the previous LayerNorm's, as fp16 rows -- statistics, the
    LayerNorm and its operand copy still come from the f32 values in the registers; the LayerNorm-backward GEMM reads such rows.  bf16 refuses the flag."""
    if ln_tile_rows and N != 512:
        pytest.skip('the 64-row tiles exist at N = 512 only')
    dt = torch.float16
    A, B = rnd(M, K, dtype=dt, seed=40), rnd(N, K, dtype=dt, seed=41, scale=0.1)
    bias, gamma, beta = rnd(N, seed=42), rnd(N, seed=43) + 1, rnd(N, seed=44)
    resid = rnd(M, N, seed=45)
    v = A.double() @ B.double().t() + bias.double() + resid.double()
    y, x_t, mean, rstd = hipops.gemm_ln(A, B, bias, gamma, beta, 1e-5, resid=resid, sums16=True)
    y32, x32, mean32, rstd32 = hipops.gemm_ln(A, B, bias, gamma, beta, 1e-5, resid=resid)
    assert y.dtype == dt and torch.equal(y, y32.to(dt))                        # the stored sums ARE the f32 sums, rounded once
    assert torch.equal(x_t, x32) and torch.equal(mean, mean32) and torch.equal(rstd, rstd32)     # everything else is untouched by the flag
    # residual = the previous LayerNorm's output recomputed from ITS 16-bit sums and (f32) statistics
    ry32 = rnd(M, N, seed=46) * 2 + 0.3
    ry = ry32.to(dt)
    rg, rb = rnd(N, seed=47) + 1, rnd(N, seed=48)
    rmean, rrstd = ry32.double().mean(1), 1 / torch.sqrt(ry32.double().var(1, unbiased=False) + 1e-5)      # statistics of the unrounded sums, as the producing kernel leaves them
    r2 = (ry.double() - rmean[:, None]) * rrstd[:, None] * rg.double() + rb.double()
    v2 = A.double() @ B.double().t() + bias.double() + r2
    y2, x2, _, _ = hipops.gemm_ln(A, B, bias, gamma, beta, 1e-5, prev=(ry, rmean.float(), rrstd.float(), rg, rb), sums16=True)
    assert relerr(y2, v2) < 5e-4 and relerr(x2, torch.nn.functional.layer_norm(v2, (N,), gamma.double(), beta.double(), 1e-5)) < 5e-4
    with pytest.raises(_hip.HipExtensionError):
        hipops.gemm_ln(A.to(torch.bfloat16), B.to(torch.bfloat16), bias, gamma, beta, 1e-5, resid=resid, sums16=True)
    # LayerNorm backward inside the data-gradient GEMM, from 16-bit rows
    aux = rnd(M, N, dtype=dt, seed=52)
    yd = ry.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = torch.zeros(N, dtype=torch.float64, device=dev(), requires_grad=True)
    Kb = K
    Ab, Bb = rnd(M, Kb, dtype=dt, seed=50), rnd(N, Kb, dtype=dt, seed=51, scale=0.1)
    vb = Ab.double() @ Bb.double().t() + aux.double()
    m16, r16 = ry.double().mean(1), 1 / torch.sqrt(ry.double().var(1, unbiased=False) + 1e-5)
    torch.nn.functional.layer_norm(yd, (N,), gd, bd, 1e-5).backward(vb)
    dx_t, dgamma, dbeta = hipops.gemm_lnbwd(Ab, Bb, aux, ry, m16.float(), r16.float(), gamma)
    assert torch.isnan(dx_t[M:].float()).all()
    assert relerr(dx_t[:M], yd.grad) < 5e-4 and relerr(dgamma, gd.grad) < 2.5e-4 and relerr(dbeta, bd.grad) < 1e-4
    dx32, dg32, db32 = hipops.gemm_lnbwd(Ab, Bb, aux, ry.float(), m16.float(), r16.float(), gamma)      # the same rows handed over as f32: the same arithmetic
    assert torch.equal(dx32[:M], dx_t[:M])


@pytest.fixture(params=[8, 4], ids=['8-waves-128x64', '4-waves-128x128'])
def wgrad_waves(request):
    """PFN_TUNE_WGRAD_WAVES: the grouped weight-gradient launch as eight waves of 128 x 64 (gemm_tn_big_kernel) or four of 128 x 128 (gemm_tn_wide_kernel)."""
    _hip.check(_hip.lib().pfn_set_tuning(14, request.param), 'pfn_set_tuning')
    yield request.param
    _hip.check(_hip.lib().pfn_set_tuning(14, 8), 'pfn_set_tuning')


@pytest.mark.parametrize('M,splits', [(64, 1), (1000, 1), (1000, 0), (4100, 3), (37, 1), (130, 1)])
def test_gemm_tn_group(M, splits, op16, wgrad_waves):
    """Grouped 256x256 weight-gradient kernel: several problems in one launch, ragged token tail, fused bias gradient."""
    shapes = [(512, 256), (256, 768), (256, 256)]
    probs, refs = [], []
    for i, (P, Q) in enumerate(shapes):
        A, B = rnd(M, P, dtype=hipops.TDT[op16], seed=20 + i), rnd(M, Q, dtype=hipops.TDT[op16], seed=30 + i)
        C = torch.ones(P, Q, device=dev())
        cs = torch.full((P,), 2.0, device=dev()) if i != 1 else None
        probs.append((A, B, C, cs))
        refs.append((A.double().t() @ B.double() + 1.0, A.double().sum(0) + 2.0))
    hipops.gemm_tn_group(probs, splits)
    for (A, B, C, cs), (rc, rs) in zip(probs, refs):
        assert relerr(C, rc) < 3e-6, relerr(C, rc)
        if cs is not None:
            assert relerr(cs, rs) < 3e-6, relerr(cs, rs)


def test_gemm_tn_group_asymmetric(op16, wgrad_waves):
    M = 256
    A = torch.zeros(M, 256, dtype=hipops.TDT[op16], device=dev()); A[torch.arange(M), torch.arange(M)] = 1
    B = (torch.arange(M * 512, device=dev()).float().view(M, 512) % 127 / 8).to(hipops.TDT[op16])
    C = torch.zeros(256, 512, device=dev())
    hipops.gemm_tn_group([(A, B, C, None)], 1)
    assert torch.equal(C, A.float().t() @ B.float())


def test_gemm_tn_asymmetric(op16):
    M = 64
    A = torch.zeros(M, 128, dtype=hipops.TDT[op16], device=dev()); A[torch.arange(M), torch.arange(M)] = 1
    B = (torch.arange(M * 128, device=dev()).float().view(M, 128) % 127 / 8).to(hipops.TDT[op16])
    C = torch.zeros(128, 128, device=dev())
    hipops.gemm_tn(A, B, C, op16)
    ref = A.float().t() @ B.float()
    assert torch.equal(C, ref)


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('rows,E', [(1000, 512), (37, 128), (513, 1024), (64, 200)])
def test_layernorm(rows, E, prec):
    x = rnd(rows, E, seed=10) * 2 + 0.5
    gamma, beta = rnd(E, seed=11) + 1, rnd(E, seed=12)
    y32, yt, mean, rstd = hipops.layernorm_fwd(x, gamma, beta, 1e-5, prec)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (E,), gd, bd, 1e-5)
    assert maxerr(y32, ref) < 1e-5
    assert relerr(yt, ref) < tol(prec, 4e-3, 5e-4, 1e-6)
    dy = rnd(rows, E, seed=13)
    ref.backward(dy.double())
    dx32, dxt, dg, db, dbias = hipops.layernorm_bwd(dy, x, gamma, mean, rstd, prec)
    assert relerr(dx32, xd.grad) < 1e-5
    assert relerr(dg, gd.grad) < 1e-5 and relerr(db, bd.grad) < 1e-5
    assert relerr(dbias, xd.grad.sum(0)) < 1e-4
    # the form the backward schedule uses: upstream gradient in operand precision, operand-precision dx only
    dy_t = dy.to(hipops.TDT[prec])
    xd2 = x.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd2, (E,), gamma.double(), beta.double(), 1e-5).backward(dy_t.double())
    none32, dxt2, dg2, db2, _ = hipops.layernorm_bwd(dy_t, x, gamma, mean, rstd, prec, want_f32=False)
    assert none32 is None and relerr(dxt2, xd2.grad) < tol(prec, 4e-3, 5e-4, 1e-5)
    assert relerr(db2, dy_t.double().sum(0)) < 1e-5


def attention_reference(qkv, H, sep):
    """Dense masked attention in f64 with the mask of generate_D_q_matrix (transformer.py:34-41)."""
    B, S, E3 = qkv.shape
    E = E3 // 3
    D = E // H
    q, k, v = [t.view(B, S, H, D).transpose(1, 2) for t in qkv.double().split(E, dim=-1)]  # [B,H,S,D]
    scores = q @ k.transpose(-1, -2) / math.sqrt(D)
    allowed = torch.zeros(S, S, dtype=torch.bool, device=qkv.device)
    allowed[:, :sep] = True
    allowed |= torch.eye(S, dtype=torch.bool, device=qkv.device)
    scores = scores.masked_fill(~allowed, float('-inf'))
    lse = torch.logsumexp(scores, -1)
    out = torch.softmax(scores, -1) @ v
    return out.transpose(1, 2).reshape(B, S, E), lse


ATTN_CASES = [  # B, S, E, H, sep
    (2, 100, 128, 4, 70), (1, 256, 256, 2, 256), (2, 130, 128, 2, 0), (1, 333, 512, 4, 301), (2, 200, 128, 2, 1),
    (1, 2000, 512, 4, 1755), (1, 257, 64, 2, 64), (1, 300, 512, 2, 129),
    (1, 4000, 1024, 4, 3549), (1, 4000, 1024, 16, 3549),     # BASELINE config 5 length: head dim 256 (nhead 4) and 64 (nhead 16)
]


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('B,S,E,H,sep', ATTN_CASES)
def test_attention_forward_backward(B, S, E, H, sep, prec):
    dt = hipops.TDT[prec]
    qkv = rnd(B, S, 3 * E, dtype=dt, seed=20)
    ctx, lse = hipops.attention_fwd(qkv, H, sep, prec)
    qd = qkv.double().requires_grad_(True)
    ref, ref_lse = attention_reference(qd, H, sep)
    assert relerr(ctx, ref) < tol(prec, 6e-3, 8e-4, 2e-5), relerr(ctx, ref)
    assert maxerr(lse, ref_lse) < tol(prec, 2e-2, 2.5e-3, 1e-4)
    dctx = rnd(B, S, E, dtype=dt, seed=21)
    ref.backward(dctx.double())
    dqkv = hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, prec)
    assert not torch.isnan(dqkv.float()).any()
    floor = 1e-2 * dctx.double().norm().item()  # dq is exactly 0 when sep == 0 (one key per row)
    for name, got, want in zip('qkv', dqkv.float().split(E, -1), qd.grad.split(E, -1)):
        err = (got.double() - want).norm().item() / max(want.norm().item(), floor)
        assert err < tol(prec, 1.5e-2, 2e-3, 5e-5), (name, err)


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('B,S,E,H,sep', [(2, 700, 256, 2, 600), (3, 600, 128, 4, 300), (1, 1100, 512, 2, 1000), (2, 520, 64, 2, 512), (2, 300, 128, 2, 200)])
def test_attention_from_a_query_block(B, S, E, H, sep, prec):
    """q_begin (the top encoder layer, whose train rows feed nothing): the launches that skip the query blocks below sep return, on the rows they
    do write, exactly what the full launches return when d(ctx) is zero on the train rows -- and never read the skipped rows (NaN there)."""
    dt = hipops.TDT[prec]
    q0 = sep // 256 * 256
    qkv = rnd(B, S, 3 * E, dtype=dt, seed=23)
    ctx, lse = hipops.attention_fwd(qkv, H, sep, prec)
    ctx_f, lse_f = hipops.attention_fwd(qkv, H, sep, prec, q_begin=sep)
    assert torch.equal(ctx_f[:, q0:], ctx[:, q0:]) and torch.equal(lse_f[:, :, q0:], lse[:, :, q0:])
    assert torch.isnan(ctx_f[:, :q0].float()).all() and torch.isnan(lse_f[:, :, :q0]).all()      # skipped rows: not written
    dctx = rnd(B, S, E, dtype=dt, seed=24)
    dctx[:, :sep] = 0
    want = hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, prec)
    poisoned = dctx.clone()
    poisoned[:, :q0] = float('nan')                                                                # ... and not read
    got = hipops.attention_bwd(qkv, ctx_f, lse_f, poisoned, H, sep, prec, q_begin=sep)
    assert not torch.isnan(got.float()).any()
    assert (got[:, :q0, :E] == 0).all()
    assert relerr(got, want) < 1e-6, relerr(got, want)
    # against the dense reference too
    qd = qkv.double().requires_grad_(True)
    ref, _ = attention_reference(qd, H, sep)
    ref.backward(dctx.double())
    floor = 1e-2 * dctx.double().norm().item()
    for name, g, w in zip('qkv', got.float().split(E, -1), qd.grad.split(E, -1)):
        err = (g.double() - w).norm().item() / max(w.norm().item(), floor)
        assert err < tol(prec, 1.5e-2, 2e-3, 5e-5), (name, err)


@pytest.mark.parametrize('dtype,W', [(torch.float32, 64), (torch.bfloat16, 40), (torch.float32, 1), (torch.bfloat16, 6), (torch.float16, 40)])
def test_gather_and_scatter_test_rows(dtype, W):
    B, S, sep = 3, 37, 21
    src = rnd(B, S, W, dtype=dtype, seed=25)
    got = hipops.gather_rows(src, sep)
    want = src[:, sep:].transpose(0, 1).reshape((S - sep) * B, W)      # row (t - sep) * B + b
    assert torch.equal(got, want)
    back = hipops.scatter_rows(got, B, S, sep, zero_from=16, fill=7.0)
    assert torch.equal(back[:, sep:], src[:, sep:]) and (back[:, 16:sep] == 0).all() and (back[:, :16] == 7.0).all()


def test_attention_online_softmax_rescale():
    """A late, very large score forces the running-max rescale branch (guide 5.4 rule 26)."""
    B, S, E, H, sep = 1, 256, 128, 1, 256
    qkv = rnd(B, S, 3 * E, seed=22) * 0.2
    qkv[0, 5, :E] *= 40.0
    qkv[0, 200, E:2 * E] = qkv[0, 5, :E] / 8
    ctx, lse = hipops.attention_fwd(qkv, H, sep, F32)
    ref, ref_lse = attention_reference(qkv, H, sep)
    assert relerr(ctx, ref) < 2e-5 and maxerr(lse, ref_lse) < 1e-3


def test_bar_distribution_kernels():
    from transformerscandobayesianinference_amd import bar_distribution as bd
    torch.manual_seed(0)
    for nb, R, full in [(100, 257, True), (1000, 500, True), (10, 33, False), (7, 64, True)]:
        borders = torch.sort(torch.randn(nb + 1))[0].to(dev())
        crit = (bd.FullSupportBarDistribution if full else bd.BarDistribution)(borders)
        logits = rnd(R, nb, seed=30).requires_grad_(True)
        lo, hi = borders[0].item(), borders[-1].item()
        y = torch.empty(R, device=dev()).uniform_(lo - (1.0 if full else 0.0), hi + (1.0 if full else 0.0))
        if not full:
            y = y.clamp(lo, hi)
        y[0], y[1] = borders[0], borders[-1]
        y[2] = borders[3]
        nll = crit(logits, y)
        # PyTorch statement of the reference math (bar_distribution.py:19-33, 89-108)
        ld = logits.detach().double().requires_grad_(True)
        bw = (borders[1:] - borders[:-1]).double()
        t = torch.searchsorted(borders, y) - 1
        t[y == borders[0]] = 0
        t[y == borders[-1]] = nb - 1
        t = t.clamp(0, nb - 1)
        lp = (torch.log_softmax(ld, -1) - torch.log(bw)).gather(-1, t[:, None]).squeeze(-1)
        if full:
            icdf = torch.distributions.HalfNormal(torch.tensor(1., dtype=torch.float64)).icdf(torch.tensor(.5, dtype=torch.float64))
            hn0 = torch.distributions.HalfNormal((bw[0] / icdf).to(dev()))
            hn1 = torch.distributions.HalfNormal((bw[-1] / icdf).to(dev()))
            m0, m1 = t == 0, t == nb - 1
            lp = lp + m0 * (hn0.log_prob((borders[1] - y).clamp(min=1e-8).double()) + torch.log(bw[0]))
            lp = lp + m1 * (hn1.log_prob((y - borders[-2]).clamp(min=0).double()) + torch.log(bw[-1]))
        assert maxerr(nll, -lp) < 2e-4, maxerr(nll, -lp)
        w = rnd(R, seed=31)
        (nll * w).sum().backward()
        (-lp * w.double()).sum().backward()
        assert relerr(logits.grad, ld.grad) < 1e-5
        means = crit.mean(logits)
        ref_mean = torch.softmax(ld.detach(), -1) @ crit.bucket_means().double()
        assert maxerr(means, ref_mean) < 1e-4


def test_clip_adam_matches_torch():
    n = 100_000
    p0, g0 = rnd(n, seed=40), rnd(n, seed=41) * 0.05
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    p, m, v = p0.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    scratch = torch.zeros(2048, device=dev())
    for step in range(1, 4):
        g = g0 * step
        ref_p.grad = g.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        gg = g.clone()
        _hip.check(_hip.lib().pfn_clip_adam_step(p.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8,
                                                 1.0, 1.0, step, 1, scratch.data_ptr(), _hip.stream_ptr()), 'adam')
        assert abs(scratch[0].item() - norm_ref.item()) < 1e-3 * norm_ref.item()
        assert gg.abs().max().item() == 0.0
        assert maxerr(p, ref_p.detach()) < 2e-6


def test_clip_adam_skips_a_step_whose_gradient_is_not_finite():
    """pfn_clip_adam_step: an inf / NaN anywhere in the gradient leaves parameters and moments untouched, clears the gradient when asked and counts the step in
    scratch[1025] (GradScaler's found_inf behaviour; include/pfn_hip.h) -- the next finite step runs as if the skipped ones had not happened."""
    n = 70_000
    p0, g0 = rnd(n, seed=42), rnd(n, seed=43) * 0.05
    p, m, v = p0.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    scratch = torch.zeros(2048, device=dev())

    def step(pp, g, mm, vv, k, sc):
        _hip.check(_hip.lib().pfn_clip_adam_step(pp.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 1.0, 1.0, k, 1,
                                                 sc.data_ptr(), _hip.stream_ptr()), 'adam')
    step(p, g0.clone(), m, v, 1, scratch)
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    assert scratch[1025].item() == 0 and not torch.equal(p1, p0)
    for bad in (float('inf'), float('-inf'), float('nan')):
        g = g0.clone()
        g[n // 2 + 3] = bad
        step(p, g, m, v, 2, scratch)
        assert torch.equal(p, p1) and torch.equal(m, m1) and torch.equal(v, v1)
        assert g.abs().max().item() == 0.0 and not math.isfinite(scratch[0].item())
    assert scratch[1025].item() == 3
    ref, rm, rv = p1.clone(), m1.clone(), v1.clone()       # the same finite step on a copy that never saw the bad ones
    step(ref, g0 * 2, rm, rv, 2, torch.zeros(2048, device=dev()))
    step(p, g0 * 2, m, v, 2, scratch)
    assert torch.equal(p, ref) and torch.equal(m, rm) and torch.equal(v, rv) and scratch[1025].item() == 3
