"""Time the attention kernels at the north-star micro-batch shape with alternative builds of the library
(ablation builds from csrc with -DPFN_ATTN_ABLATE=n, see attention.hip): one subprocess per library.

    python tools/exp_attn_variants.py [lib.so ...]      ('' = the in-tree library)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, torch
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import _hip
if LIB: _hip.LIB_PATH = LIB
from transformerscandobayesianinference_amd import hipops
B, S, E, H, sep = 16, 2000, 512, 4, 1603
qkv = (torch.randn(B, S, 3 * E, device='cuda') * 0.5).to(torch.bfloat16)
dctx = (torch.randn(B, S, E, device='cuda') * 0.5).to(torch.bfloat16)
ctx, lse = hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16)
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
f = t(lambda: hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16))
bw = t(lambda: hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16))
print(f'{os.path.basename(LIB) if LIB else "in-tree":24s} fwd {f:7.1f} us   bwd {bw:7.1f} us')
'''
for lib in (sys.argv[1:] or ['']):
    code = f'ROOT = {ROOT!r}\nLIB = {os.path.abspath(lib) if lib else ""!r}\n' + CHILD
    subprocess.run([sys.executable, '-c', code], check=False)
