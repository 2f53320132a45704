"""Error anatomy of the product precision at the benchmarked shape (BASELINE.json configs[1]): the HIP path in bf16 and
in exact-f32 mode against the f64 CPU oracle on the same fixed-seed GP draw and the SAME state dict (weights and bar
borders), optionally after `--train N` optimizer steps of the bf16 model (an untrained PFN predicts the prior mean ~ 0 for
every point, so errors relative to its means say nothing).

    python tools/parity_probe.py [--train N] [--lr LR] [--layers L ...]
"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import pfn_oracle  # noqa: E402


def train(model, w, steps, lr, batch=32):
    from transformerscandobayesianinference_amd.optim import FusedClipAdam
    from transformerscandobayesianinference_amd.priors import fast_gp
    from transformerscandobayesianinference_amd.streams import MicroBatchStreams
    from transformerscandobayesianinference_amd.utils import get_weighted_single_eval_pos_sampler
    import contextlib, io
    dev = next(model.parameters()).device
    model.train()
    opt = FusedClipAdam(model, lr=lr, max_grad_norm=1.0)
    sampler = get_weighted_single_eval_pos_sampler(w['bptt'])
    with contextlib.redirect_stdout(io.StringIO()):
        dl = iter(fast_gp.DataLoader(num_steps=steps, batch_size=batch, seq_len=w['bptt'], num_features=w['num_features'],
                                     hyperparameters=w['hyperparameters'], device=dev))
    micro = MicroBatchStreams(2)
    O = w['num_bars']
    t0 = time.time()
    for it in range(steps):
        sep = sampler()
        for g in opt.param_groups:
            g['lr'] = lr * min(1.0, (it + 1) / 50)
        (x, y), target = next(dl)
        losses = micro.forward_backward(model, (x, y), target, sep,
                                        lambda out, tg: model.criterion(out.reshape(-1, O), tg[sep:].reshape(-1)).view(out.shape[0], -1))
        opt.step(zero_grad=True)
        if it % 100 == 0 or it == steps - 1:
            print(json.dumps(dict(train_step=it, loss=losses.mean().item(), sep=sep, seconds=time.time() - t0)), flush=True)
    model.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train', type=int, default=0)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--layers', type=int, nargs='*', default=[6])
    ap.add_argument('--sep', type=int, default=1755)
    args = ap.parse_args()
    dev = torch.device('cuda')
    w = dict(bench.WORKLOAD)
    B, S, sep = 2, w['bptt'], args.sep
    gen = torch.Generator().manual_seed(1234)
    x, y, _ = pfn_oracle.get_batch_fast_gp(B, S, w['num_features'], w['hyperparameters'], gen, dtype=torch.float64)
    for L in args.layers:
        w['nlayers'] = L
        random.seed(1)
        torch.manual_seed(1)
        model16 = bench.build_model(dev, 'bf16', w)
        if args.train:
            train(model16, w, args.train, args.lr)
        model16.eval()
        sd = {k: v.detach().cpu().clone() for k, v in model16.state_dict().items()}
        model32 = bench.build_model(dev, 'f32', w)
        model32.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        model32.criterion.load_state_dict(model16.criterion.state_dict())
        model32.eval()
        borders = sd['criterion.borders'].double()
        t0 = time.time()
        lo = pfn_oracle.forward(sd, x, y, sep, w['nhead'], dtype=torch.float64)
        nll_rows_o = pfn_oracle.bar_nll(lo.reshape(-1, w['num_bars']), y[sep:].reshape(-1), borders)
        nll_o = nll_rows_o.mean().item()
        mean_o = pfn_oracle.bar_mean(lo, borders)
        secs = time.time() - t0
        for prec, model in (('f32', model32), ('bf16', model16)):
            with torch.no_grad():
                lg = model((x.to(dev).float(), y.to(dev).float()), single_eval_pos=sep)
                nll_rows = model.criterion(lg.reshape(-1, w['num_bars']), y[sep:].to(dev).float().reshape(-1)).double().cpu()
                mean = model.criterion.mean(lg).double().cpu()
            nll = nll_rows.mean().item()
            lg = lg.double().cpu()
            d = mean - mean_o
            rec = dict(L=L, trained_steps=args.train, precision=prec, logits_std=lo.std().item(), logits_rel_l2=((lg - lo).norm() / lo.norm()).item(),
                       nll_hip=nll, nll_oracle=nll_o, nll_rel=abs(nll - nll_o) / abs(nll_o),
                       nll_row_absdiff_max=(nll_rows - nll_rows_o).abs().max().item(),
                       mean_rel_l2=(d.norm() / mean_o.norm()).item(),
                       mean_max_over_range=(d.abs().max() / (mean_o.max() - mean_o.min())).item(),
                       mean_max_over_y_range=(d.abs().max() / (y.max() - y.min())).item(),
                       mean_abs_max=d.abs().max().item(), mean_ref_rms=mean_o.pow(2).mean().sqrt().item(),
                       mean_ref_range=(mean_o.max() - mean_o.min()).item(),
                       mean_vs_y_rmse=(mean_o - y[sep:].double()).pow(2).mean().sqrt().item(), y_std=y.std().item(), oracle_s=secs)
            print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
