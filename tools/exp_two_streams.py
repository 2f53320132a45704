"""Experiment: one optimizer step as two half-batches on two HIP streams (kernels of the two halves can
fill each other's memory-bound prologues / epilogues) vs one full batch on one stream."""
import contextlib, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import _hip
from transformerscandobayesianinference_amd.optim import FusedClipAdam
from transformerscandobayesianinference_amd.priors import fast_gp

w = bench.WORKLOAD
dev = torch.device('cuda')
S, nf, O, sep = w['bptt'], w['num_features'], w['num_bars'], 1755
with contextlib.redirect_stdout(io.StringIO()):
    model = bench.build_model(dev, 'bf16')
model.train()
opt = FusedClipAdam(model, lr=1e-4, max_grad_norm=1.0)


def timed(fn, iters=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / iters


for B, NS in ((32, (2,)), (48, (2, 3)), (64, (2, 4))):
    x, y, target = fast_gp.get_batch(B, S, nf, device=dev, hyperparameters=w['hyperparameters'])

    def one():
        logits = model((x, y), single_eval_pos=sep)
        model.criterion(logits.reshape(-1, O), target[sep:].reshape(-1)).mean().backward()
        opt.step(zero_grad=True)

    for nstream in NS:
        streams = [torch.cuda.Stream() for _ in range(nstream)]
        h = B // nstream

        def split():
            main = torch.cuda.current_stream()
            model.flat_parameters()
            model._refresh_shadow(_hip.stream_ptr(dev))
            for i, s in enumerate(streams):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    xs, ys, ts = x[:, i * h:(i + 1) * h], y[:, i * h:(i + 1) * h], target[:, i * h:(i + 1) * h]
                    logits = model((xs, ys), single_eval_pos=sep)
                    (model.criterion(logits.reshape(-1, O), ts[sep:].reshape(-1)).mean() / nstream).backward()
            for s in streams:
                main.wait_stream(s)
            opt.step(zero_grad=True)

        t1 = timed(one)
        t2 = timed(split)
        print(f'B={B}: one stream {t1*1e3:.3f} ms ({B/t1:.0f}/s) | {nstream} streams {t2*1e3:.3f} ms ({B/t2:.0f}/s)')
