"""Where the step time goes (run on the GPU box): the GP prior draw alone, the training step on a
fixed batch (forward / backward / optimizer separately and together), at the north-star shape."""
import argparse
import contextlib
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transformerscandobayesianinference_amd.optim import FusedClipAdam  # noqa: E402
from transformerscandobayesianinference_amd.priors import fast_gp  # noqa: E402


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--sep', type=int, default=1755)
    args = ap.parse_args()
    w = bench.WORKLOAD
    dev = torch.device('cuda')
    S, nf, O, B, sep = w['bptt'], w['num_features'], w['num_bars'], args.batch, args.sep
    with contextlib.redirect_stdout(io.StringIO()):
        model = bench.build_model(dev, 'bf16')
    model.train()
    opt = FusedClipAdam(model, lr=1e-4, max_grad_norm=1.0)
    t_gp = timed(lambda: fast_gp.get_batch(B, S, nf, device=dev, hyperparameters=w['hyperparameters']))
    print(f'gp draw        B={B}: {t_gp * 1e3:8.3f} ms  ({B / t_gp:8.1f} datasets/s)')
    x, y, target = fast_gp.get_batch(B, S, nf, device=dev, hyperparameters=w['hyperparameters'])

    def fwd():
        return model((x, y), single_eval_pos=sep)

    def fwd_loss():
        logits = fwd()
        return model.criterion(logits.reshape(-1, O), target[sep:].reshape(-1)).mean()

    def fwd_bwd():
        fwd_loss().backward()

    def full():
        fwd_bwd()
        opt.step(zero_grad=True)

    with torch.no_grad():
        t_f = timed(fwd)
    t_fl = timed(fwd_loss)
    t_fb = timed(fwd_bwd)
    t_full = timed(full)
    print(f'forward (no grad)   : {t_f * 1e3:8.3f} ms')
    print(f'forward + loss      : {t_fl * 1e3:8.3f} ms')
    print(f'fwd + loss + bwd    : {t_fb * 1e3:8.3f} ms')
    print(f'full step, fixed batch: {t_full * 1e3:8.3f} ms  ({B / t_full:8.1f} datasets/s)')
    # host-side cost of enqueueing a step (no sync inside): time to return from the Python calls
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        full()
    t_host = (time.time() - t0) / 5
    torch.cuda.synchronize()
    print(f'host enqueue time per step: {t_host * 1e3:8.3f} ms')


if __name__ == '__main__':
    main()
