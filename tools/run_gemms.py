"""Launch the grouped weight-gradient GEMM and two NT GEMMs a few times at the north-star micro-batch shape (for rocprofv3 --pmc
passes; tools/pmc_gemm.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
H = _hip
M, E, F, L = 32000, 512, 1024, 6
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device='cuda') * 0.5).to(bf)
group = []
for _ in range(L):
    for P, Q in ((3 * E, E), (E, E), (F, E), (E, F)):
        group.append((r(M, P), r(M, Q), torch.zeros(P, Q, device='cuda'), torch.zeros(P, device='cuda')))
A, B = r(M, E), r(3 * E, E)
qkv = torch.empty(M, 3 * E, dtype=bf, device='cuda')
bias = torch.randn(3 * E, device='cuda')
for _ in range(3):
    hipops.gemm_tn_group(group, 0)
    hipops.gemm_nt(A, B, H.EPI_BIAS | H.EPI_OUT_T, H.PREC_BF16, bias=bias, out_t=qkv)
torch.cuda.synchronize()
