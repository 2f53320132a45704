"""CPU experiment (analysis infrastructure, imports the oracle): the dynamic range of the BACKWARD's 16-bit tensors against max|dlogits| -- what the fp16 loss-scale
target (csrc/pfn_device.h loss_scale_exp, PFN_TUNE_LOSS_SCALE_TARGET) has to leave room for.

The f64 oracle forward of tools/sim_operand_formats.py is run with autograd on the trained configs[0]-shaped checkpoint (tests/golden/trained_config1.pt) and on
untrained weights at the same shape; for every tensor class the product path stores in 16 bits on the way back (d logits, d GELU output, d layer input, dq / dk / dv,
d attention output) it reports max|grad| / max|dlogits| per eval position, and the share of the tensor's squared norm carried by elements that would be subnormal
(< 2^-14) or flushed (< 2^-25) under a given target.  fp16 holds up to 65504 = 2^16: target t leaves 16 - t binades above max|dlogits|.

    python tools/sim_gradient_range.py --out profiles/r06_gradient_range_simulation.json
"""
import argparse, json, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pfn_oracle as O


def forward_with_taps(sd, x, y, sep, nhead, taps):
    dt = torch.float64
    p = {k: v.detach().to(dt) for k, v in sd.items() if not k.startswith('criterion.')}
    x, y = x.to(dt), y.to(dt)
    T, B, _ = x.shape

    def tap(name, t):
        t.retain_grad()
        taps.setdefault(name, []).append(t)
        return t
    emb = O._linear(x, p['encoder.weight'], p['encoder.bias'])
    h = torch.cat([emb[:sep] + O._linear(y.unsqueeze(-1), p['y_encoder.weight'], p['y_encoder.bias'])[:sep], emb[sep:]], 0).requires_grad_(True)
    E = h.shape[-1]; D = E // nhead
    mask = O.d_q_mask(T, sep, dt, h.device)
    L = 1 + max(int(k.split('.')[2]) for k in p if k.startswith('transformer_encoder.layers.'))
    for l in range(L):
        pre = f'transformer_encoder.layers.{l}.'
        h = tap('layer input (dX)', h * 1.)
        qkv = O._linear(h, p[pre + 'self_attn.in_proj_weight'], p[pre + 'self_attn.in_proj_bias'])
        q, k, v = qkv.split(E, -1)
        q, k, v = tap('dq', q * 1.), tap('dk', k * 1.), tap('dv', v * 1.)
        q, k, v = [t.reshape(T, B, nhead, D).permute(1, 2, 0, 3) for t in (q, k, v)]
        s = q @ k.transpose(-1, -2) / math.sqrt(D) + mask
        ctx = torch.softmax(s, -1) @ v
        ctx = tap('d attention output', ctx.permute(2, 0, 1, 3).reshape(T, B, E) * 1.)
        h = O._layer_norm(h + O._linear(ctx, p[pre + 'self_attn.out_proj.weight'], p[pre + 'self_attn.out_proj.bias']), p[pre + 'norm1.weight'], p[pre + 'norm1.bias'])
        h = tap('FFN input (dX)', h * 1.)
        act = tap('d GELU output', O._gelu(tap('d FFN pre-activation', O._linear(h, p[pre + 'linear1.weight'], p[pre + 'linear1.bias']))) * 1.)
        h = O._layer_norm(h + O._linear(act, p[pre + 'linear2.weight'], p[pre + 'linear2.bias']), p[pre + 'norm2.weight'], p[pre + 'norm2.bias'])
    d = tap('d GELU output', O._gelu(tap('d FFN pre-activation', O._linear(tap('layer input (dX)', h[sep:] * 1.), p['decoder.0.weight'], p['decoder.0.bias']))) * 1.)
    return tap('d logits', O._linear(d, p['decoder.2.weight'], p['decoder.2.bias']) * 1.)


def analyse(name, sd, x, y, seps, nhead, borders, nbars, targets, rows):
    for sep in seps:
        taps = {}
        lg = forward_with_taps(sd, x, y, sep, nhead, taps)
        O.bar_nll(lg.reshape(-1, nbars), y[sep:].reshape(-1).double(), borders).mean().backward()
        amax = taps['d logits'][0].grad.abs().max().item()
        for cls, ts in taps.items():
            g = torch.cat([t.grad.flatten() for t in ts]).abs() / amax
            r = dict(model=name, sep=sep, tensor=cls, max_over_dlogits_max=g.max().item(), log2_max=math.log2(g.max().item()), rms_over_dlogits_max=g.pow(2).mean().sqrt().item())
            tot = g.pow(2).sum()
            for t in targets:
                sc = g * 2. ** t
                r[f'target{t}'] = dict(overflowing_elements=int((sc > 65504.).sum()), sqnorm_share_subnormal=(sc[sc < 2. ** -14].pow(2).sum() / tot).item(),
                                       sqnorm_share_flushed=(sc[sc < 2. ** -25].pow(2).sum() / tot).item())
            rows.append(r)
            print(json.dumps(r), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    rows = []
    sd, _ = torch.load(os.path.join(ROOT, 'tests', 'golden', 'trained_config1.pt'))
    gen = torch.Generator().manual_seed(2024)
    x, y, _ = O.get_batch_fast_gp(8, 100, 5, {'noise': 1e-4, 'outputscale': 1., 'lengthscale': .6}, gen)
    targets = (6, 2, 0, -2)
    analyse('trained_config1.pt (bptt 100, emsize 128, 2 layers, 9.6 M datasets)', sd, x, y, (1, 5, 20, 50, 81, 99), 4, sd['criterion.borders'].double(), 100, targets, rows)
    if a.out:
        json.dump(dict(rows=rows, note='f64 oracle + autograd; ratios are |gradient element| / max|dlogits| of the same backward; targets: log2 of where max|dlogits| is scaled to'),
                  open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
