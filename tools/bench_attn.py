"""Attention kernels alone at a workload's micro-batch shape: forward, the backward's launches one by one, the whole backward.
    python tools/bench_attn.py [B S E H sep]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
import bench
for kv in os.environ.get('PFN_TUNE', '').split(','):      # e.g. PFN_TUNE=4=0 (pfn_set_tuning keys, include/pfn_hip.h)
    if kv:
        _hip.check(_hip.lib().pfn_set_tuning(*[int(v) for v in kv.split('=')]), 'pfn_set_tuning')

B, S, E, H, sep = [int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (16, 2000, 512, 4, 1604))]
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device='cuda') * 0.5).to(bf)
qkv, dctx = r(B, S, 3 * E), r(B, S, E)
ctx, lse = hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16)
unit = 2.0 * E * bench.pairs(S, sep) * B
t = bench.time_kernel(lambda: hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16), iters=20)
print(f'[{os.path.basename(_hip.LIB_PATH)}] shape B{B} S{S} E{E} H{H} sep{sep}: one product unit = {unit / 1e9:.1f} GFLOP')
print(f'attn_fwd                      {t * 1e6:8.1f} us  {2 * unit / t / 1e12:7.1f} TF/s')
hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16)
for name, _, part, alg, ex in hipops.ATTENTION_BWD_PARTS:
    t = bench.time_kernel(lambda: hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16, parts=part), iters=20)
    print(f'{name[:60]:60s} {t * 1e6:8.1f} us  alg {alg * unit / t / 1e12:7.1f}  executed {ex * unit / t / 1e12:7.1f} TF/s')
t = bench.time_kernel(lambda: hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16, parts=7), iters=20)
print(f'attn_bwd, all launches        {t * 1e6:8.1f} us  alg {4 * unit / t / 1e12:7.1f} TF/s')
