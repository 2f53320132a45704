"""The fused dgrad + LayerNorm-backward GEMM against the pair of kernels it replaces (gemm_nt with the bf16 residual epilogue, then
layernorm_bwd) at the north-star micro-batch shape:  python tools/bench_gemm_lnbwd.py [--batch 16]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
H = _hip
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
w = bench.WORKLOAD
E, F = w['emsize'], w['nhid']
M = a.batch * w['bptt']
dev = torch.device('cuda'); bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)
f = lambda *s: torch.randn(*s, device=dev)
for name, K in (('dx1 = dh.W1 + dy2 -> LN1 backward', F), ('dx = dqkv.Win + dy1 -> LN2 backward', 3 * E)):
    A, B, aux = r(M, K), r(E, K), r(M, E)
    if os.environ.get('PFN_A_RESIDENT') == '1':      # experiment: every row of A is the same 2-3 KB (row stride 0): the A operand stream hits in cache instead of coming from HBM
        A = A[:1].expand(M, K)
    y, gamma = f(M, E), f(E)
    mean, rstd = y.mean(1), 1 / torch.sqrt(y.var(1, unbiased=False) + 1e-5)
    out = (torch.empty(M, E, dtype=bf, device=dev), torch.zeros(E, device=dev), torch.zeros(E, device=dev))
    gA = torch.empty(M, E, dtype=bf, device=dev)
    for _ in range(2):
        t_f = bench.time_kernel(lambda: hipops.gemm_lnbwd(A, B, aux, y, mean, rstd, gamma, out=out))
        t_g = bench.time_kernel(lambda: hipops.gemm_nt(A, B, H.EPI_RESID_T | H.EPI_OUT_T, H.PREC_BF16, aux=aux, out_t=gA))
        t_l = bench.time_kernel(lambda: hipops.layernorm_bwd(gA, y, gamma, mean, rstd, H.PREC_BF16, want_f32=False))
    print(f'{name:40s} {M}x{E}x{K}: fused {t_f * 1e6:7.1f} us | gemm_nt {t_g * 1e6:7.1f} + layernorm_bwd {t_l * 1e6:7.1f} = {(t_g + t_l) * 1e6:7.1f} us')
