"""LayerNorm backward at the north-star micro-batch shape (32 000 tokens x 512): the form the backward schedule runs
(operand-precision dY in, operand-precision dX out) and the f32 form."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
rows, E = 32000, 512
x = torch.randn(rows, E, device='cuda')
gamma, beta = torch.rand(E, device='cuda') + .5, torch.randn(E, device='cuda')
_, _, mean, rstd = hipops.layernorm_fwd(x, gamma, beta, 1e-5, _hip.PREC_BF16)
dy = torch.randn(rows, E, device='cuda')
dyt = dy.bfloat16()
def t(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
lib = _hip.lib()
dxt = torch.empty(rows, E, dtype=torch.bfloat16, device='cuda'); dx32 = torch.empty_like(x)
dg, db = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
sp = hipops.sp()
call = lambda dyp, is_t, d32: lib.pfn_op_layernorm_bwd(dyp.data_ptr(), is_t, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), d32, dxt.data_ptr(),
                                                       dg.data_ptr(), db.data_ptr(), 0, rows, E, _hip.PREC_BF16, sp)
us = t(lambda: call(dyt, 1, 0))
print(f'ln_bwd bf16 dY -> bf16 dX : {us:6.1f} us   {(rows * E * (4 + 2 + 2)) / us / 1e6:5.2f} TB/s')
us = t(lambda: call(dy, 0, dx32.data_ptr()))
print(f'ln_bwd f32 dY -> f32+bf16 : {us:6.1f} us   {(rows * E * (4 + 4 + 4 + 2)) / us / 1e6:5.2f} TB/s')
