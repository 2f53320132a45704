"""Fused linear + residual + LayerNorm kernel vs the unfused pair, north-star shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
H = _hip
M, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 16) * 2000, 512
dev = torch.device('cuda'); bf = torch.bfloat16
for K in (512, 1024):
    A, B = (torch.randn(M, K, device=dev) * .5).to(bf), (torch.randn(N, K, device=dev) * .1).to(bf)
    if os.environ.get('PFN_A_RESIDENT') == '1':      # experiment: every row of A is the same 2-3 KB (row stride 0): the A operand stream hits in cache instead of coming from HBM
        A = A[:1].expand(M, K)
    bias, gamma, beta = torch.randn(N, device=dev), torch.randn(N, device=dev) + 1, torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev)
    bufs = (torch.empty(M + 2, N, device=dev), torch.empty(M, N, dtype=bf, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev))
    t_f = bench.time_kernel(lambda: hipops.gemm_ln(A, B, bias, gamma, beta, 1e-5, resid=resid, out=bufs))
    out = torch.empty(M, N, device=dev)
    H.lib().pfn_set_tuning(0, 2)
    t_g = bench.time_kernel(lambda: hipops.gemm_nt(A, B, H.EPI_BIAS | H.EPI_RESID | H.EPI_OUT_F32, H.PREC_BF16, bias=bias, resid=resid, out_f32=out))
    H.lib().pfn_set_tuning(0, 0)
    t_l = bench.time_kernel(lambda: hipops.layernorm_fwd(out, gamma, beta, 1e-5, H.PREC_BF16))
    print(f'K={K}: fused {t_f*1e6:.1f} us | gemm {t_g*1e6:.1f} + ln {t_l*1e6:.1f} = {(t_g+t_l)*1e6:.1f} us')
