"""The attention backward's launches under several builds of libpfn_hip.so in ONE process (tools/build_variants.sh -> _variants/libpfn_<name>.so):
    python tools/bench_kv_variants.py name[@key=value,...] [name ...] [-- B S E H sep]
Each launch is timed inside the backward's sequence (bench.time_sequence).  Ablation builds return garbage; only the durations mean anything."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import _hip, hipops
import bench

args = sys.argv[1:]
shape = (32, 2000, 512, 4, 1604)
if '--' in args:
    shape = tuple(int(v) for v in args[args.index('--') + 1:])
    args = args[:args.index('--')]
B, S, E, H, sep = shape
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device='cuda') * 0.5).to(bf)
qkv, dctx = r(B, S, 3 * E), r(B, S, E)
unit = 2.0 * E * bench.pairs(S, sep) * B
print(f'shape B{B} S{S} E{E} H{H} sep{sep}: one product unit = {unit / 1e9:.1f} GFLOP')
for name in args:
    lib_name, _, tune = name.partition('@')           # name@key=value[,key=value]: pfn_set_tuning keys for this run
    _hip._lib = None
    _hip.LIB_PATH = os.path.join(ROOT, 'transformerscandobayesianinference_amd', '_variants', f'libpfn_{lib_name}.so')
    for kv in filter(None, tune.split(',')):
        _hip.check(_hip.lib().pfn_set_tuning(*[int(v) for v in kv.split('=')]), 'pfn_set_tuning')
    ctx, lse = hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16)
    hipops._bwd_scratch.clear()
    hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16)
    tf = bench.time_kernel(lambda: hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16), iters=10)
    seq = bench.time_sequence([(lambda part=part: hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16, parts=part))
                               for _, _, part, _, _ in hipops.ATTENTION_BWD_PARTS], iters=10)
    print(f'{name:14s} fwd {tf * 1e6:7.1f}   delta {seq[0] * 1e6:6.1f}   key-block pass {seq[1] * 1e6:7.1f} us ({3 * unit / seq[1] / 1e12:6.1f} TF/s alg)   query-block pass {seq[2] * 1e6:7.1f}')
