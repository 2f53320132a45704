"""Where does the product precision lose accuracy?  bf16 vs exact-f32 HIP paths (the f32 path matches the f64 oracle to
1e-6, tools/parity_probe.py) on the same fixed-seed GP draw at the benchmarked shape: per-row statistics of the logit and
bar-NLL differences, for several depths."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import pfn_oracle  # noqa: E402


def main():
    dev = torch.device('cuda')
    w = dict(bench.WORKLOAD)
    B, S, sep = 2, w['bptt'], 1755
    gen = torch.Generator().manual_seed(1234)
    x, y, _ = pfn_oracle.get_batch_fast_gp(B, S, w['num_features'], w['hyperparameters'], gen, dtype=torch.float64)
    x, y = x.float().to(dev), y.float().to(dev)
    for L in [int(a) for a in (sys.argv[1:] or ['0', '1', '6'])]:
        w['nlayers'] = L
        res = {}
        first = bench.build_model(dev, 'f32', w)
        sd = {k: v.detach().clone() for k, v in first.state_dict().items()}   # ONE state dict (weights and bar borders) for both precisions
        for prec in ('f32', 'bf16'):
            model = bench.build_model(dev, prec, w)
            model.load_state_dict(sd)
            model.eval()
            with torch.no_grad():
                lg = model((x, y), single_eval_pos=sep)
                nll = model.criterion(lg.reshape(-1, w['num_bars']), y[sep:].reshape(-1))
            res[prec] = (lg.double(), nll.double())
        (l32, n32), (l16, n16) = res['f32'], res['bf16']
        d = (l16 - l32)
        rows = d.reshape(-1, w['num_bars'])
        ref = l32.reshape(-1, w['num_bars'])
        cen = ref - ref.mean(1, keepdim=True)
        dcen = rows - rows.mean(1, keepdim=True)
        dn = n16 - n32
        rec = dict(L=L, logits_std=ref.std().item(), logits_absmean=ref.abs().mean().item(), logits_centered_std=cen.std().item(),
                   dlogits_rms=rows.pow(2).mean().sqrt().item(), dlogits_centered_rms=dcen.pow(2).mean().sqrt().item(),
                   dlogits_rowmean_rms=rows.mean(1).pow(2).mean().sqrt().item(),
                   dlogits_colmean_rms=rows.mean(0).pow(2).mean().sqrt().item(),     # common to all rows (coherent part)
                   row_rel_max=(rows.norm(dim=1) / ref.norm(dim=1)).max().item(),
                   nll32=n32.mean().item(), nll16=n16.mean().item(), dnll_mean=dn.mean().item(), dnll_std=dn.std().item(),
                   dnll_absmax=dn.abs().max().item(), n_rows=dn.numel(), rows_over_0p05=int((dn.abs() > 0.05).sum()))
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
