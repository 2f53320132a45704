"""Gradient parity AT THE BENCHMARKED SHAPES, numbers on record: the HIP backward in bf16 (and exact-f32
mode where it exists) against `oracle.loss_and_grads` in f64 on the same inputs and weights.

    python tools/grad_parity.py [--case config2 config5slice config4] --out gpurun_out/r03_grad_parity.json

  config2      : BASELINE configs[1] -- bptt 2000, 18 features, emsize 512, 4 heads, 6 layers, 1000 bars; B 2, eval position 1755
  config5slice : configs[4] width -- bptt 4000, emsize 1024, 4 heads (head dim 256), nhid 2048, 2 layers; B 1, eval position 3549
  config4      : configs[3] -- bptt 1000, 60 features, BCE head, 6 layers; B 2, eval positions 437 and 500 (the bench's parity_sep)
Per case: loss, logits, the global gradient and every parameter tensor's gradient (relative L2), sorted worst first.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pfn_oracle  # noqa: E402
from transformerscandobayesianinference_amd import bar_distribution, encoders  # noqa: E402
from transformerscandobayesianinference_amd.transformer import TransformerModel  # noqa: E402

DEV = 'cuda:0'
CASES = {
    'config2': dict(T=2000, B=2, F=18, E=512, H=4, nhid=1024, L=6, nbars=1000, seps=[1755], precisions=['bf16', 'f32']),
    'config5slice': dict(T=4000, B=1, F=18, E=1024, H=4, nhid=2048, L=2, nbars=1000, seps=[3549], precisions=['bf16']),
    'config4': dict(T=1000, B=2, F=60, E=512, H=4, nhid=1024, L=6, nbars=1, seps=[437, 500], precisions=['bf16', 'f32'], bce=True),
}


def make(cfg, precision, seed):
    torch.manual_seed(seed)
    m = TransformerModel(encoders.Linear(cfg['F'], cfg['E']), cfg['nbars'], cfg['E'], cfg['H'], cfg['nhid'], cfg['L'], 0.0,
                         y_encoder=encoders.Linear(1, cfg['E']), precision=precision, eval_precision=precision)
    if not cfg.get('bce'):
        m.criterion = bar_distribution.FullSupportBarDistribution(torch.sort(torch.randn(cfg['nbars'] + 1) * 1.5)[0])
    with torch.no_grad():
        for layer in m.transformer_encoder.layers:          # un-zero the residual branches
            for t in (layer.linear2.weight, layer.self_attn.out_proj.weight):
                t.normal_(0, 0.03)
    return m


def oracle_bce(sd, x, y, sep, H):
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items() if not k.startswith('criterion.')}
    logits = pfn_oracle.forward(leaves, x, y, sep, H, torch.float64)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.squeeze(-1), y[sep:].double())
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in leaves.items()}


def run_case(name, cfg, log):
    out = dict(shape={k: v for k, v in cfg.items() if k not in ('seps', 'precisions')}, results=[])
    gen = torch.Generator().manual_seed(17)
    if cfg.get('bce'):
        x = torch.randn(cfg['T'], cfg['B'], cfg['F'], generator=gen)
        y = (torch.rand(cfg['T'], cfg['B'], generator=gen) > 0.5).float()
    else:
        x, y, _ = pfn_oracle.get_batch_fast_gp(cfg['B'], cfg['T'], cfg['F'], {'noise': 1e-4, 'outputscale': 1., 'lengthscale': .6}, gen)
    for sep in cfg['seps']:
        sd = {k: v.clone() for k, v in make(cfg, 'bf16', seed=9).state_dict().items()}
        t0 = time.time()
        if cfg.get('bce'):
            loss_o, logits_o, grads_o = oracle_bce(sd, x, y, sep, cfg['H'])
        else:
            loss_o, logits_o, grads_o = pfn_oracle.loss_and_grads(sd, x, y, y, sep, cfg['H'], sd['criterion.borders'])
        oracle_s = time.time() - t0
        tot = math.sqrt(sum((g ** 2).sum().item() for g in grads_o.values()))
        for prec in cfg['precisions']:
            model = make(cfg, prec, seed=9).to(DEV).train()
            model.zero_grad()
            logits = model((x.to(DEV), y.to(DEV)), single_eval_pos=sep)
            if cfg.get('bce'):
                loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.squeeze(-1), y[sep:].to(DEV))
            else:
                loss = model.criterion(logits.reshape(-1, cfg['nbars']), y[sep:].to(DEV).flatten()).mean()
            loss.backward()
            per = {}
            err2 = 0.0
            for k, p in model.named_parameters():
                g, go = p.grad.double().cpu(), grads_o[k]
                e = (g - go).norm().item()
                err2 += e * e
                per[k] = dict(rel=e / max(go.norm().item(), 1e-30), norm=go.norm().item(), share_of_global_error=None)
            for k in per:
                per[k]['share_of_global_error'] = (per[k]['rel'] * per[k]['norm']) ** 2 / max(err2, 1e-300)
            worst = sorted(per.items(), key=lambda kv: -kv[1]['rel'])
            rec = dict(case=name, precision=prec, sep=sep, oracle_seconds=oracle_s, loss_hip=loss.item(), loss_oracle=loss_o.item(),
                       loss_rel=abs(loss.item() - loss_o.item()) / abs(loss_o.item()),
                       logits_rel_l2=((logits.double().cpu() - logits_o).norm() / logits_o.norm()).item(),
                       grad_global_rel_l2=math.sqrt(err2) / tot, grad_norm_oracle=tot,
                       per_tensor_rel_l2={k: v['rel'] for k, v in worst},
                       per_tensor_share_of_global_error={k: v['share_of_global_error'] for k, v in sorted(per.items(), key=lambda kv: -kv[1]['share_of_global_error'])[:8]})
            out['results'].append(rec)
            log(f"{name} {prec} sep {sep}: loss_rel {rec['loss_rel']:.2e} logits {rec['logits_rel_l2']:.2e} grad global {rec['grad_global_rel_l2']:.2e} "
                f"worst tensors {[(k.replace('transformer_encoder.layers.', 'L'), round(v['rel'], 4)) for k, v in worst[:4]]} (oracle {oracle_s:.0f}s)")
            del model
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--case', nargs='+', default=['config2', 'config5slice', 'config4'], choices=sorted(CASES))
    ap.add_argument('--out', default='gpurun_out/r03_grad_parity.json')
    args = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    log = lambda s: print(s, flush=True)
    result = json.load(open(args.out)) if os.path.exists(args.out) else {}
    for c in args.case:
        result[c] = run_case(c, CASES[c], log)
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(result, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
