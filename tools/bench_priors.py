"""The prior samplers alone (run on the GPU box), as records for profiles/: python tools/bench_priors.py
  * GP prior (configs 1-3): datasets/s of one sampler call of 128 datasets at bptt 2000 and its f32 rate against the 157 TF/s roof;
  * GP-mixture prior (config 5): the same at bptt 4000;
  * BNN prior (config 4): one get_batch of 64 datasets at bptt 1000, 60 features, split into parameter draws (torch), the batched
    network forward (HIP) and the post-processing tail."""
import os, sys, time, random
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd.priors import fast_gp, fast_gp_mix, mlp


def timed(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


def gp_flops(S, nf):   # Gram S^2 (2 nf + 4) + Cholesky S^3 / 3 + L z S^2
    return S * S * (2 * nf + 4) + S ** 3 / 3 + S * S


torch.manual_seed(0); random.seed(0); np.random.seed(0)
w = bench.CONFIGS[2]
for B in (32, 128):
    t = timed(lambda: fast_gp.get_batch(B, w['bptt'], w['num_features'], device='cuda', hyperparameters=w['hyperparameters']))
    print(f'fast_gp   bptt 2000 B={B:4d}: {t * 1e3:8.3f} ms/call  {B / t:9.0f} datasets/s  {t / B * 1e6:7.1f} us/dataset  '
          f'{gp_flops(2000, 18) * B / t / 1e12:6.1f} TF/s f32 = {gp_flops(2000, 18) * B / t / 157.3e12:.2f} of the f32 MFMA roof')
w5 = bench.CONFIGS[5]
for B in (8, 32):
    t = timed(lambda: fast_gp_mix.get_batch(B, w5['bptt'], w5['num_features'], device='cuda', hyperparameters={}), iters=3, warm=1)
    print(f'fast_gp_mix bptt 4000 B={B:4d}: {t * 1e3:8.3f} ms/call  {B / t:9.0f} datasets/s  {t / B * 1e6:7.1f} us/dataset  '
          f'{gp_flops(4000, 18) * B / t / 1e12:6.1f} TF/s f32 = {gp_flops(4000, 18) * B / t / 157.3e12:.2f} of the f32 MFMA roof')
w4 = bench.CONFIGS[4]
kw = {k: v for k, v in bench.prior_kwargs(w4).items() if k != 'num_features'}
t_all = timed(lambda: mlp.get_batch(64, w4['bptt'], w4['num_features'], device='cuda', **kw))
# parts, on the tensors of one call
orig_fwd, orig_post = mlp.forward_networks, mlp.postprocess
rec = {}
def fwd(*a, **k):
    rec['fwd'] = (a, k); return orig_fwd(*a, **k)
def post(*a, **k):
    rec['post'] = (a, k); return orig_post(*a, **k)
mlp.forward_networks, mlp.postprocess = fwd, post
mlp.get_batch(64, w4['bptt'], w4['num_features'], device='cuda', **kw)
mlp.forward_networks, mlp.postprocess = orig_fwd, orig_post
t_fwd = timed(lambda: orig_fwd(*rec['fwd'][0], **rec['fwd'][1]))
t_post = timed(lambda: orig_post(*rec['post'][0], **rec['post'][1]))
print(f'priors.mlp bptt 1000, 60 features, B=64: {t_all * 1e3:8.3f} ms/call ({64 / t_all:7.0f} datasets/s): network forward (HIP) {t_fwd * 1e3:.3f} ms, '
      f'post-processing {t_post * 1e3:.3f} ms, parameter draws + host samplers {max(t_all - t_fwd - t_post, 0) * 1e3:.3f} ms')
