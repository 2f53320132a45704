"""A few forward + loss + backward passes of one micro-batch of the benchmarked model on one stream (for rocprofv3 runs that compare
builds of a small kernel inside the step):  [PFN_LIB=<variant .so>] python tools/run_step.py [--batch 32] [--steps 3]"""
import argparse, contextlib, io, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import _hip
if os.environ.get('PFN_LIB'):
    _hip.LIB_PATH = os.path.abspath(os.environ['PFN_LIB'])
import bench
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--steps', type=int, default=3)
a = ap.parse_args()
w = bench.WORKLOAD
dev = torch.device('cuda')
with contextlib.redirect_stdout(io.StringIO()):
    model = bench.build_model(dev, 'bf16', w)
model.train()
S, nf, sep = w['bptt'], w['num_features'], 1604
x, y = torch.rand(S, a.batch, nf, device=dev), torch.randn(S, a.batch, device=dev)
for _ in range(a.steps):
    out = model((x, y), single_eval_pos=sep)
    loss = model.criterion(out.reshape(-1, w['num_bars']), y[sep:].reshape(-1)).mean()
    loss.backward()
torch.cuda.synchronize()
print('loss', loss.item())
