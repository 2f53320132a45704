"""The LayerNorm-fused GEMMs of the encoder (out_proj + LN, linear2 + LN, dy1 = LN1 backward of dh.W1 + dy2, dy2 = LN2 backward of dqkv.Win + dy1) at the
north-star micro-batch, on 128-row tiles (one workgroup per CU) and on 64-row tiles (two per CU): PFN_TUNE_GEMM_LN_ROWS 0 / 1, interleaved A/B in one process.
    python tools/bench_ln_tiles.py [datasets per micro-batch = 32]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import hipops, _hip
H = _hip
M, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 32) * 2000, 512
dev = torch.device('cuda'); bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)
f32 = lambda *s: torch.randn(*s, device=dev)
rows = []
for name, K, kind in (('out_proj + LN', 512, 'ln'), ('linear2 + LN', 1024, 'ln'), ('dy1 (lnbwd, K 1024)', 1024, 'lnb'), ('dy2 (lnbwd, K 1536)', 1536, 'lnb')):
    A, B_ = r(M, K), (torch.randn(N, K, device=dev) * .1).to(bf)
    if kind == 'ln':
        bias, gamma, beta, resid = f32(N), f32(N) + 1, f32(N), f32(M, N)
        bufs = (torch.empty(M + 2, N, device=dev), torch.empty(M, N, dtype=bf, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev))
        fn = lambda: hipops.gemm_ln(A, B_, bias, gamma, beta, 1e-5, resid=resid, out=bufs)
        nbytes = sum(v.numel() * v.element_size() for v in (A, B_, resid) + bufs)
    else:
        aux, y, gamma = r(M, N), f32(M, N), f32(N)
        mean, rstd = y.mean(1), 1 / torch.sqrt(y.var(1, unbiased=False) + 1e-5)
        bufs = (torch.empty(M, N, dtype=bf, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev))
        fn = lambda: hipops.gemm_lnbwd(A, B_, aux, y, mean, rstd, gamma, out=bufs)
        nbytes = sum(v.numel() * v.element_size() for v in (A, B_, aux, y, mean, rstd, bufs[0]))
    t = {0: [], 1: []}
    for rep in range(3):
        for mode in (0, 1):
            H.check(H.lib().pfn_set_tuning(7, mode), 'tuning')
            t[mode].append(bench.time_kernel(fn, iters=20, warm=3) * 1e6)
    H.lib().pfn_set_tuning(7, 0)
    floor = nbytes / 6.3e12 * 1e6
    print(f'{name:24s} M={M} K={K}: 128-row {min(t[0]):7.1f} us ({min(t[0]) / floor:.2f} x HBM floor {floor:.1f}) | 64-row x 2/CU {min(t[1]):7.1f} us ({min(t[1]) / floor:.2f} x) | all {[round(v, 1) for v in t[0]]} {[round(v, 1) for v in t[1]]}')
