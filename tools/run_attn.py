"""Launch the attention kernels a few times at the north-star shape (for rocprofv3 --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
B, S, E, H, sep = 16, 2000, 512, 4, 1604
qkv = (torch.randn(B, S, 3 * E, device='cuda') * 0.5).to(torch.bfloat16)
dctx = (torch.randn(B, S, E, device='cuda') * 0.5).to(torch.bfloat16)
for _ in range(3):
    ctx, lse = hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16)
    hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16)
torch.cuda.synchronize()
