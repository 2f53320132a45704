"""Reference-side control for "it trains" (weakness 6 / item 4): the reference's OWN `train.train` (imported from /root/reference, CPU) and this
repo's `train()` (HIP stack, bf16) run on IDENTICAL batches and eval positions -- regenerated on both sides from seeds -- for the GP prior of BASELINE configs[1]
(noise 1e-4, outputscale 1, lengthscale 0.6) with 5 features (the notebook's recipe) and with 18 features (configs[1] as written), at a CPU-feasible size.

    python tools/reference_curve.py --reference      # build container only: runs the reference, writes profiles/r04_reference_vs_hip_curves.json
    python tools/reference_curve.py --hip            # GPU box: replays the same stream through the HIP stack, adds its curves to that file

Test infrastructure / measurement tool; nothing here is on the product path."""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'profiles', 'r04_reference_vs_hip_curves.json')
CFG = dict(bptt=200, batch_size=8, emsize=128, nhid=256, nlayers=3, nhead=4, num_bars=100, epochs=24, steps_per_epoch=50, warmup_epochs=4, lr=1e-3,
           hyperparameters=dict(noise=1e-4, outputscale=1.0, lengthscale=0.6))


def gp_batch(seed, index, B, T, F, hp):
    """(x[T,B,F], y[T,B]) of batch `index`: a pure function of (seed, index) -- f64 Cholesky on the host (priors/fast_gp.py:41-58 restated)."""
    g = torch.Generator().manual_seed(seed * 1000003 + index)
    x = torch.rand(B, T, F, generator=g)
    z = torch.randn(B, T, generator=g)
    xs = x.double() / hp['lengthscale']
    d2 = (xs.unsqueeze(2) - xs.unsqueeze(1)).pow(2).sum(-1)
    K = hp['outputscale'] * torch.exp(-0.5 * d2) + hp['noise'] * torch.eye(T, dtype=torch.float64)
    y = (torch.linalg.cholesky(K) @ z.double().unsqueeze(-1)).squeeze(-1).float()
    return x.transpose(0, 1).contiguous(), y.transpose(0, 1).contiguous()


class SeededLoader:
    num_outputs = 1
    fuse_x_y = False

    def __init__(self, num_steps, batch_size=None, seq_len=None, num_features=None, seed=0, hyperparameters=None, device=None, **_):
        self.num_steps, self.B, self.T, self.num_features, self.seed, self.hp, self.cursor = num_steps, batch_size, seq_len, num_features, seed, hyperparameters, 0

    def __len__(self):
        return self.num_steps

    def __iter__(self):
        for _ in range(self.num_steps):
            x, y = gp_batch(self.seed, self.cursor, self.B, self.T, self.num_features, self.hp)
            self.cursor += 1
            yield (x, y), y.clone()


def borders_for(F, bars_mod):
    ys = torch.cat([gp_batch(999, i, 50, 20, F, CFG['hyperparameters'])[1].flatten() for i in range(4)])
    return bars_mod.get_bucket_limits(CFG['num_bars'], ys=ys)


def run(train_fn, bars_mod, encoders, schedule_fn, F, seed, record_cls_base, **extra):
    log = []

    class Recording(record_cls_base):
        def forward(self, logits, y):
            losses = super().forward(logits, y)
            log.append(losses.detach().mean())
            return losses
    Recording.__name__ = 'RecordingFullSupportBarDistribution'
    rng = random.Random(seed)
    T = CFG['bptt']
    weights = [1 / (T - i) for i in range(T)]
    sep = lambda: rng.choices(range(T), weights)[0]                 # utils.get_weighted_single_eval_pos_sampler on a private stream
    torch.manual_seed(seed)                                          # model initialisation (each side initialises with its own modules: same distribution)
    t0 = time.time()
    total, _, model = train_fn(SeededLoader, Recording(borders_for(F, bars_mod)), encoders.Linear, emsize=CFG['emsize'], nhid=CFG['nhid'], nlayers=CFG['nlayers'],
                               nhead=CFG['nhead'], dropout=0.0, epochs=CFG['epochs'], steps_per_epoch=CFG['steps_per_epoch'], batch_size=CFG['batch_size'],
                               bptt=T, lr=CFG['lr'], warmup_epochs=CFG['warmup_epochs'], y_encoder_generator=encoders.Linear,
                               extra_prior_kwargs_dict=dict(num_features=F, seed=seed, hyperparameters=CFG['hyperparameters']), scheduler=schedule_fn,
                               single_eval_pos_gen=sep, verbose=False, **extra)
    spe = CFG['steps_per_epoch']
    losses = [float(v) for v in log]
    return dict(epoch_losses=[sum(losses[e * spe:(e + 1) * spe]) / spe for e in range(CFG['epochs'])], seconds=time.time() - t0, final=total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', action='store_true')
    ap.add_argument('--hip', action='store_true')
    a = ap.parse_args()
    rec = json.load(open(OUT)) if os.path.exists(OUT) else dict(config=CFG, runs={})
    if a.reference:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import make_golden
        ref = make_golden.import_reference()
        torch.set_num_threads(8)
        for F in (5, 18):
            r = run(ref['train'].train, ref['bar_distribution'], ref['encoders'], ref['utils'].get_cosine_schedule_with_warmup, F, seed=7,
                    record_cls_base=ref['bar_distribution'].FullSupportBarDistribution)
            rec['runs'].setdefault(f'{F}_features', {})['reference_cpu_f32'] = r
            print(F, 'features, reference:', [round(v, 3) for v in r['epoch_losses']], f"{r['seconds']:.0f} s")
    if a.hip:
        torch.set_num_threads(4)       # (the per-batch 200 x 200 f64 Cholesky on the host: a many-core box thrashes with its default thread count)
        sys.path.insert(0, ROOT)
        from transformerscandobayesianinference_amd import bar_distribution, encoders, train as train_mod, utils
        for F in (5, 18):
            for precision in ('bf16', 'f32'):
                r = run(train_mod.train, bar_distribution, encoders, utils.get_cosine_schedule_with_warmup, F, seed=7,
                        record_cls_base=bar_distribution.FullSupportBarDistribution, gpu_device='cuda:0', precision=precision, micro_streams=1)
                rec['runs'].setdefault(f'{F}_features', {})[f'hip_{precision}'] = r
                print(F, 'features, HIP', precision, [round(v, 3) for v in r['epoch_losses']], f"{r['seconds']:.0f} s")
    for F, runs in rec['runs'].items():
        if 'reference_cpu_f32' in runs:
            for k, v in runs.items():
                if k != 'reference_cpu_f32':
                    d = [abs(x - y) for x, y in zip(v['epoch_losses'], runs['reference_cpu_f32']['epoch_losses'])]
                    v['max_abs_epoch_loss_difference_vs_reference'] = max(d)
                    v['last_epoch_loss_difference_vs_reference'] = v['epoch_losses'][-1] - runs['reference_cpu_f32']['epoch_losses'][-1]
    rec['note'] = ('identical batches and eval positions on both sides (regenerated from seeds); initial weights are drawn by each side\'s own modules from the same '
                   'distributions (torch.manual_seed(7): the reference and this repo construct the same parameter tensors in the same order, so the draws coincide '
                   'where the construction order does); epoch_losses = mean training loss per epoch (bar NLL, nats); the prior level is the first epoch (lr 0)')
    json.dump(rec, open(OUT, 'w'), indent=1)
    if a.hip and os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        json.dump(rec, open(os.path.join(ROOT, 'gpurun_out', os.path.basename(OUT)), 'w'), indent=1)


if __name__ == '__main__':
    main()
