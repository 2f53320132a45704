"""What would fusing the Q projection into the attention kernel buy (north_star wording: "QKV projection + attention as one fused
kernel")?  K and V must exist for every key before any query block can run, so only Q can move into the attention prologue.
This times, at the north-star micro-batch shape, the packed projection as it is (N = 3E), the K|V part alone (N = 2E) and the Q
part alone (N = E): (N = 3E) - (N = 2E) is what the GEMM side would save, and the Q-only GEMM -- a 256-row tile of it is what an
attention workgroup would have to compute itself before its first key tile -- is the work that would move into the prologue of
every attention workgroup, at that kernel's efficiency instead of the GEMM's."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
M, E = 32000, 512
bf = torch.bfloat16
A = (torch.randn(M, E, device='cuda') * 0.5).to(bf)
for name, N in (('q|k|v (as shipped)', 3 * E), ('k|v only', 2 * E), ('q only', E)):
    W = (torch.randn(N, E, device='cuda') * 0.05).to(bf)
    bias = torch.randn(N, device='cuda')
    out = torch.empty(M, N, dtype=bf, device='cuda')
    t = bench.time_kernel(lambda: hipops.gemm_nt(A, W, _hip.EPI_BIAS | _hip.EPI_OUT_T, _hip.PREC_BF16, bias=bias, out_t=out), iters=30)
    print(f'{name:22s} N={N:5d}: {t * 1e6:7.1f} us  {2.0 * M * N * E / t / 1e12:7.1f} TF/s  output {M * N * 2 / 1e6:6.1f} MB')
