"""The grouped weight-gradient launch alone at the north-star micro-batch shape (24 problems, 32000 tokens):
    [PFN_LIB=<variant .so>] python tools/bench_wgrad.py [--batch 16]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import hipops
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--tune', default='', help='key=value,... for pfn_set_tuning (14=4: four waves of 128 x 128; 11=N: token splits)')
a = ap.parse_args()
for kv in a.tune.split(','):
    if kv:
        _hip.check(_hip.lib().pfn_set_tuning(*[int(v) for v in kv.split('=')]), 'pfn_set_tuning')
w = bench.WORKLOAD
E, F, L = w['emsize'], w['nhid'], w['nlayers']
M = a.batch * w['bptt']
r = lambda *s: (torch.randn(*s, device='cuda') * 0.5).to(torch.bfloat16)
probs, refs = [], []
for _ in range(L):
    for P, Q, cs in ((E, F, False), (F, E, True), (E, E, False), (3 * E, E, True)):
        probs.append((r(M, P), r(M, Q), torch.zeros(P, Q, device='cuda'), torch.zeros(P, device='cuda') if cs else None))
hipops.gemm_tn_group(probs, 0)
A, B, C, cs = probs[1]
ref = A.float().t() @ B.float()
err = ((C - ref).norm() / ref.norm()).item()
cerr = ((cs - A.float().sum(0)).norm() / A.float().sum(0).norm()).item()
for _ in range(3):
    t = bench.time_kernel(lambda: hipops.gemm_tn_group(probs, 0), iters=5, warm=2)
flops = 2.0 * M * L * (2 * E * F + 4 * E * E)
print(f'[{os.path.basename(_hip.LIB_PATH)} tune={a.tune or "-"}] grouped weight gradients: {t * 1e6:8.1f} us  {flops / t / 1e12:6.0f} TF/s   (rel. error of one product {err:.2e}, of its column sums {cerr:.2e})')
