"""The encoder layer's packed q|k|v projection alone at the micro-batch shape, plain and with the keys centred (pfn_op_qkv_projection: key_shift kernel + the GEMM's
EPI_ROWSHIFT epilogue), and the k|v-only GEMM of the fused-Q schedule.   python tools/bench_qkv.py [B S E sep]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import hipops, _hip
import bench
B, S, E, sep = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 2000, 512, 1604))]
for dt in (torch.bfloat16, torch.float16):
    x = (torch.randn(B, S, E, device='cuda') * 0.5).to(dt)
    w = (torch.randn(3 * E, E, device='cuda') * 0.05).to(dt)
    b = torch.randn(3 * E, device='cuda')
    qkv = torch.empty(B, S, 3 * E, dtype=dt, device='cuda')
    ks = torch.empty(B, E, device='cuda')
    prec = hipops.PREC_OF[dt]
    lib = _hip.lib()
    sp = hipops.sp
    for name, center in (('plain', 0), ('keys centred', 1)):
        f = lambda: _hip.check(lib.pfn_op_qkv_projection(x.data_ptr(), w.data_ptr(), b.data_ptr(), qkv.data_ptr(), ks.data_ptr(), B, S, E, sep, 0, center, prec, sp()), 'qkv')
        t = bench.time_kernel(f, iters=20)
        print(f'{str(dt):16s} q|k|v projection {B * S}x{3 * E}x{E}, {name:14s} {t * 1e6:8.1f} us  {2.0 * B * S * 3 * E * E / t / 1e12:7.1f} TF/s')
    kv = torch.empty(B * S, 2 * E, dtype=dt, device='cuda')
    t = bench.time_kernel(lambda: hipops.gemm_nt(x.view(B * S, E), w[E:], _hip.EPI_BIAS | _hip.EPI_OUT_T, prec, bias=b[E:], out_t=kv), iters=20)
    print(f'{str(dt):16s} k|v projection only {B * S}x{2 * E}x{E}               {t * 1e6:8.1f} us  {2.0 * B * S * 2 * E * E / t / 1e12:7.1f} TF/s')
