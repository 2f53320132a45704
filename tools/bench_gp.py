"""GP prior sampler alone (run on the GPU box): python tools/bench_gp.py [--batch 64]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd.priors import fast_gp
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
for kv in os.environ.get('PFN_TUNE', '').split(','):      # e.g. PFN_TUNE=8=0: pfn_set_tuning keys (include/pfn_hip.h)
    if kv:
        _hip.check(_hip.lib().pfn_set_tuning(*[int(v) for v in kv.split('=')]), 'pfn_set_tuning')
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=64); ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--no-check', action='store_true', help='leave the failure flags alone (timing of ablation builds whose factorisations fail by construction)')
a = ap.parse_args()
w = bench.WORKLOAD
f = lambda: fast_gp.get_batch(a.batch, w['bptt'], w['num_features'], device='cuda', hyperparameters=w['hyperparameters'])
if a.no_check:
    hp = w['hyperparameters']; noise, osc, ls = hp['noise'], hp['outputscale'], hp['lengthscale']
    f = lambda: fast_gp.gp_sample(a.batch, w['bptt'], w['num_features'], 'cuda', ls, osc, noise, check=False)
for _ in range(2): f()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(a.iters): f()
torch.cuda.synchronize(); t = (time.time() - t0) / a.iters
print(f'[{os.path.basename(_hip.LIB_PATH)} PFN_TUNE={os.environ.get("PFN_TUNE", "")}] gp draw B={a.batch}: {t * 1e3:.3f} ms = {t / a.batch * 1e6:.1f} us per dataset ({a.batch / t:.0f} datasets/s)')
