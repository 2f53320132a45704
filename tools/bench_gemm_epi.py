"""The encoder's NT GEMMs with their REAL epilogues at the north-star shape, per kernel-selection mode
(pfn_set_tuning key 0): python tools/bench_gemm_epi.py [--batch 16] [--modes 2,3]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
H = _hip
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=16); ap.add_argument('--modes', default='2,3')
ap.add_argument('--k', type=int, default=0, help='override the contraction length (64: one stage = prologue + epilogue only)')
ap.add_argument('--persist', type=int, default=0, help='PFN_TUNE_GEMM_PERSIST value: workgroups of the persistent kernel (0 = off)')
ap.add_argument('--lib', default=None, help='alternative build of libpfn_hip.so (experiment variants under _build/exp)')
a = ap.parse_args()
if a.lib:
    _hip.LIB_PATH = os.path.abspath(a.lib)
w = bench.WORKLOAD
E, F, L = w['emsize'], w['nhid'], w['nlayers']
M = a.batch * w['bptt']
dev = torch.device('cuda'); bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)
f = lambda *s: torch.randn(*s, device=dev)
cases = [  # name, N, K, flags, count per step
    ('qkv', 3 * E, E, H.EPI_BIAS | H.EPI_OUT_T, L),
    ('out_proj', E, E, H.EPI_BIAS | H.EPI_RESID | H.EPI_OUT_F32, L),
    ('linear1+gelu', F, E, H.EPI_BIAS | H.EPI_GELU | H.EPI_OUT_T | H.EPI_OUT2_T, L),
    ('linear2', E, F, H.EPI_BIAS | H.EPI_RESID | H.EPI_OUT_F32, L),
    ('d(hpre)', F, E, H.EPI_GELU_BWD | H.EPI_OUT_T, L),
    ('dx1', E, F, H.EPI_RESID | H.EPI_OUT_F32, L),
    ('d(ctx)', E, E, H.EPI_OUT_T, L),
    ('dx', E, 3 * E, H.EPI_RESID | H.EPI_OUT_F32, L),
]
H.check(H.lib().pfn_set_tuning(3, a.persist), 'tuning')
tot = {}
for name, N, K, flags, cnt in cases:
    K = a.k or K
    A, B = r(M, K), r(N, K)
    if os.environ.get('PFN_A_RESIDENT') == '1':      # experiment: every row of A is the same 2-3 KB (row stride 0): the A operand stream hits in cache instead of coming from HBM
        A = A[:1].expand(M, K)
    kw = {}
    if flags & H.EPI_BIAS: kw['bias'] = f(N)
    if flags & H.EPI_RESID: kw['resid'] = f(M, N)
    if flags & H.EPI_GELU_BWD: kw['aux'] = r(M, N)
    if flags & H.EPI_OUT_F32: kw['out_f32'] = torch.empty(M, N, device=dev)
    if flags & H.EPI_OUT_T: kw['out_t'] = torch.empty(M, N, dtype=bf, device=dev)
    if flags & H.EPI_OUT2_T: kw['out2_t'] = torch.empty(M, N, dtype=bf, device=dev)
    line = f'{name:14s} {M}x{N}x{K}'
    for mode in [int(m) for m in a.modes.split(',')]:
        H.check(H.lib().pfn_set_tuning(0, mode), 'tuning')
        t = bench.time_kernel(lambda: hipops.gemm_nt(A, B, flags, H.PREC_BF16, **kw))
        line += f' | mode{mode}: {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF'
        tot[mode] = tot.get(mode, 0.0) + t * cnt
    print(line)
H.lib().pfn_set_tuning(0, 0)
print('per step:', {m: f'{v * 1e3:.3f} ms' for m, v in tot.items()})
