"""Accuracy of the HIP GP sampler (y = chol(K) z in f32 on the device) against the f64 oracle on the same (x, z), for the library
selected by PFN_LIB -- used to compare the split-bf16 trailing update with the exact-f32 MFMA one.
    [PFN_LIB=<variant .so>] python tools/exp_gp_accuracy.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
from transformerscandobayesianinference_amd.priors import fast_gp
from oracle import pfn_oracle
for kv in os.environ.get('PFN_TUNE', '').split(','):      # e.g. PFN_TUNE=8=0: pfn_set_tuning keys (include/pfn_hip.h)
    if kv:
        _hip.check(_hip.lib().pfn_set_tuning(*[int(v) for v in kv.split('=')]), 'pfn_set_tuning')

g = torch.Generator().manual_seed(3)
for (B, T, F, hp, kernel) in [(4, 2000, 18, (1e-4, 1.0, 0.6), 'rbf'), (4, 2000, 5, (1e-4, 1.0, 0.6), 'rbf'), (2, 4000, 10, (1e-3, 1.0, 0.5), 'matern')]:
    x = torch.rand(B, T, F, generator=g)
    z = torch.randn(B, T, generator=g)
    noise, os_, ls = hp
    want = pfn_oracle.gp_sample(x, z, ls, os_, noise, kernel)
    _, got, _, info = fast_gp.gp_sample(B, T, F, torch.device('cuda'), ls, os_, noise, fast_gp.KERNEL_RBF if kernel == 'rbf' else fast_gp.KERNEL_MATERN52, x=x, z=z)
    got = got.double().cpu().t() if got.shape != want.shape else got.double().cpu()
    err = ((got - want).norm() / want.norm()).item()
    print(f'[{os.path.basename(_hip.LIB_PATH)}] B={B} T={T} F={F} {kernel} noise={noise}: rel. L2 error of y vs f64 = {err:.3e}, max abs {((got - want).abs().max()).item():.3e}, failed factorisations {int(info.abs().sum())}')
