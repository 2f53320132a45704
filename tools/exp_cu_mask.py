"""Experiment: the two micro-batch streams of a step confined to disjoint halves of the chip
(hipExtStreamCreateWithCUMask) -- does keeping each micro-batch on its own CUs / L2s beat letting both
spread over all 256 CUs?  Mask layouts tried: none, contiguous halves of the 256-bit mask, interleaved bits."""
import contextlib, ctypes, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transformerscandobayesianinference_amd import _hip
from transformerscandobayesianinference_amd.optim import FusedClipAdam
from transformerscandobayesianinference_amd.priors import fast_gp

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


w = bench.WORKLOAD
dev = torch.device('cuda')
S, nf, O, sep, B = w['bptt'], w['num_features'], w['num_bars'], 1603, 32
with contextlib.redirect_stdout(io.StringIO()):
    model = bench.build_model(dev, 'bf16')
model.train()
opt = FusedClipAdam(model, lr=1e-4, max_grad_norm=1.0)
x, y, target = fast_gp.get_batch(B, S, nf, device=dev, hyperparameters=w['hyperparameters'])


def timed(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / iters


def run(streams):
    h = B // len(streams)

    def split():
        main = torch.cuda.current_stream()
        model.flat_parameters()
        model._refresh_shadow(_hip.stream_ptr(dev))
        for i, s in enumerate(streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                xs, ys, ts = x[:, i * h:(i + 1) * h], y[:, i * h:(i + 1) * h], target[:, i * h:(i + 1) * h]
                logits = model((xs, ys), single_eval_pos=sep)
                (model.criterion(logits.reshape(-1, O), ts[sep:].reshape(-1)).mean() / len(streams)).backward()
        for s in streams:
            main.wait_stream(s)
        opt.step(zero_grad=True)
    return timed(split)


FULL = 0xffffffff
layouts = {
    'no mask': [torch.cuda.Stream(), torch.cuda.Stream()],
    'halves (bits 0-127 / 128-255)': [masked_stream([FULL] * 4 + [0] * 4), masked_stream([0] * 4 + [FULL] * 4)],
    'interleaved (even / odd bits)': [masked_stream([0x55555555] * 8), masked_stream([0xaaaaaaaa] * 8)],
    'interleaved by 4 (0x0f0f / 0xf0f0)': [masked_stream([0x0f0f0f0f] * 8), masked_stream([0xf0f0f0f0] * 8)],
    'both full masks': [masked_stream([FULL] * 8), masked_stream([FULL] * 8)],
}
for name, st in layouts.items():
    t = run(st)
    print(f'{name:40s} {t * 1e3:8.3f} ms   {B / t:7.0f} datasets/s')
