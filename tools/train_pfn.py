"""Train a PFN with the HIP stack and score the TRAINED checkpoint.

    python tools/train_pfn.py --stage config2 [--epochs 80 --steps-per-epoch 100 --batch 64 --lr 3e-4] --out gpurun_out/r03_trained_config2.json
    python tools/train_pfn.py --stage curves                    # the same short run in bf16 and in exact-f32 mode, same seeds
    python tools/train_pfn.py --stage config1 --ckpt gpurun_out/trained_config1.pt   # the 1.3 MB checkpoint behind tests/golden

Stages
  config2 : BASELINE.json configs[1] (priors.fast_gp {noise 1e-4, outputscale 1, lengthscale .6}, bptt 2000, 18 features,
            emsize 512, 6 layers, 1000 bars from 100000 x 20 prior ys -- the recipe of SetupForGPFittingExperiments.ipynb
            cell 5: cosine schedule with warm-up over a quarter of the epochs, weighted eval-position sampler) through
            `train.train`; per-epoch loss; then on fixed-seed draws the PFN's bar NLL / squared error of its mean per
            evaluation position next to the exact GP posterior's (priors.fast_gp.gp_posterior, reference
            priors/fast_gp.py:88-120) ON THE SAME DATA, the API-level sweeps (`evaluation.run_test`, `evaluation.gp_baseline`),
            and the parity of the trained weights in bf16 and exact-f32 mode against the f64 oracle.
  curves  : two short training runs from identical seeds (weights, prior draws, eval positions), precision bf16 vs f32.
  config1 : BASELINE.json configs[0] shape (bptt 100, 5 features, emsize 128, 2 layers, 100 bars) trained and saved.
"""
import argparse
import contextlib
import io
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HPS = dict(noise=1e-4, outputscale=1.0, lengthscale=0.6)
SHAPES = {
    'config2': dict(bench.CONFIGS[2]),
    # the recipe AS THE NOTEBOOK RUNS IT (SetupForGPFittingExperiments.ipynb cell 5): five features, otherwise configs[1]
    'notebook5': dict(bench.CONFIGS[2], num_features=5),
    'config1': dict(prior='fast_gp', bptt=100, num_features=5, emsize=128, nhead=4, nhid=256, nlayers=2, criterion='bar', num_bars=100,
                    hyperparameters=HPS, parity_batch=8, parity_sep=81, eval_pos='weighted'),
}


def seed_all(seed):
    from transformerscandobayesianinference_amd.priors import fast_gp
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    fast_gp._call_counter[0] = 0          # the sampler's Philox offset: identical draws for identical seeds


def make_borders(w, device, n=100000):
    from transformerscandobayesianinference_amd import bar_distribution
    from transformerscandobayesianinference_amd.priors import fast_gp
    ys = fast_gp.get_batch(n, 20, w['num_features'], device=device, hyperparameters=w['hyperparameters'])[1]
    with bench.quiet():
        return bar_distribution.get_bucket_limits(w['num_bars'], ys=ys.cpu())


def run_training(w, device, precision, epochs, steps_per_epoch, batch, lr, seed, log, aggregate=1, streams=2):
    """`train.train` on the configuration's prior; returns (model on the CPU, per-epoch records)."""
    from transformerscandobayesianinference_amd import bar_distribution, encoders, train as train_mod
    from transformerscandobayesianinference_amd.priors import fast_gp
    from transformerscandobayesianinference_amd.utils import get_weighted_single_eval_pos_sampler
    seed_all(seed)
    criterion = bar_distribution.FullSupportBarDistribution(make_borders(w, device))
    seed_all(seed)
    curve = []
    t0 = time.time()

    def on_epoch(model, epoch, loss, lr_now, seconds):
        curve.append(dict(epoch=epoch, loss=loss, lr=lr_now, seconds=seconds, datasets=epoch * steps_per_epoch * batch))
        if epoch % max(1, epochs // 20) == 0 or epoch == epochs:
            log(f'  epoch {epoch:4d}  loss {loss:8.4f}  lr {lr_now:.2e}  {seconds:5.1f}s  ({time.time() - t0:6.1f}s total)')

    with bench.quiet():
        _, _, model = train_mod.train(fast_gp.DataLoader, criterion, encoders.Linear, emsize=w['emsize'], nhid=w['nhid'], nlayers=w['nlayers'],
                                      nhead=w['nhead'], dropout=0.0, epochs=epochs, steps_per_epoch=steps_per_epoch, batch_size=batch,
                                      bptt=w['bptt'], lr=lr, warmup_epochs=epochs // 4, y_encoder_generator=encoders.Linear,
                                      extra_prior_kwargs_dict={'num_features': w['num_features'], 'fuse_x_y': False, 'hyperparameters': w['hyperparameters']},
                                      single_eval_pos_gen=get_weighted_single_eval_pos_sampler(w['bptt']), gpu_device=str(device),
                                      aggregate_k_gradients=aggregate, verbose=False, precision=precision, micro_streams=streams,
                                      epoch_callback=on_epoch)
    return model, curve, time.time() - t0


@torch.no_grad()
def paired_curves(model, w, device, positions, n, seed, sub=64):
    """PFN and exact GP on the SAME fixed-seed draws: for every evaluation position p the loss at point p given points
    0..p-1 (the quantity of reference priors/fast_gp.py:88-120 and of the notebook's run_test)."""
    from transformerscandobayesianinference_amd.priors import fast_gp
    S, nf = w['bptt'], w['num_features']
    model.eval()
    crit = model.criterion
    pfn_nll, pfn_se, gp_nll, gp_se, prior_nll = [], [], [], [], []
    seed_all(seed)
    for _ in range(n // sub):
        x, y, _ = fast_gp.get_batch(sub, S, nf, device=device, hyperparameters=w['hyperparameters'])
        xb, yb = x.transpose(0, 1).contiguous(), y.transpose(0, 1).contiguous()
        mean, var, nl, _ = fast_gp.gp_posterior(xb, yb, HPS['lengthscale'], HPS['outputscale'], HPS['noise'])
        idx = torch.as_tensor(positions, device=device)
        gp_nll.append(nl.index_select(1, idx))
        gp_se.append(((mean - yb) ** 2).index_select(1, idx))
        v0 = HPS['outputscale'] + HPS['noise']
        prior_nll.append((0.5 * math.log(2 * math.pi * v0) + yb ** 2 / (2 * v0)).index_select(1, idx))
        a, b = [], []
        for p in positions:
            logits = model((x[:p + 1].contiguous(), y[:p + 1].contiguous()), single_eval_pos=p)    # [1, sub, bars]
            a.append(crit(logits[0], y[p]))
            b.append((crit.mean(logits)[0] - y[p]) ** 2)
        pfn_nll.append(torch.stack(a, 1))
        pfn_se.append(torch.stack(b, 1))
    cat = lambda l: torch.cat(l).double().cpu()
    pfn_nll, pfn_se, gp_nll, gp_se, prior_nll = map(cat, (pfn_nll, pfn_se, gp_nll, gp_se, prior_nll))
    sem = lambda t: (t.std(0) / math.sqrt(t.shape[0])).tolist()
    return dict(positions=list(positions), datasets=n, seed=seed,
                pfn_bar_nll=pfn_nll.mean(0).tolist(), pfn_bar_nll_sem=sem(pfn_nll), pfn_mean_mse=pfn_se.mean(0).tolist(),
                exact_gp_nll=gp_nll.mean(0).tolist(), exact_gp_nll_sem=sem(gp_nll), exact_gp_mse=gp_se.mean(0).tolist(),
                prior_nll=prior_nll.mean(0).tolist(),
                pfn_minus_gp_nll=(pfn_nll - gp_nll).mean(0).tolist(), pfn_minus_gp_nll_sem=sem(pfn_nll - gp_nll),
                summary=dict(pfn_bar_nll=pfn_nll.mean().item(), exact_gp_nll=gp_nll.mean().item(), prior_nll=prior_nll.mean().item(),
                             pfn_mean_mse=pfn_se.mean().item(), exact_gp_mse=gp_se.mean().item()))


def parity_of(sd, w, device, seps=None):
    """The trained state dict through the product model (training precision bf16, inference precision f32) against the f64 oracle on
    the same inputs: `outputs` = model.eval() under no_grad, `training_forward` = the bf16 forward of the training path."""
    out = {}
    for prec in ('bf16', 'fp16'):      # round 6: both 16-bit training formats on the same trained weights
        model = bench.build_model(device, prec, w, criterion=_criterion_from(sd))
        model.load_state_dict(sd)
        model.to(device)
        for sep in (seps or [w['parity_sep']]):
            res, _ = bench.parity_check(model, dict(w, parity_sep=sep), device, prec)
            res.pop('against', None); res.pop('inputs', None)
            if prec == 'bf16':
                out[f'sep{sep}'] = res
            else:
                out[f'sep{sep}']['training_forward_fp16'] = res['training_forward']
        del model
    return out


def log_parity(parity, log):
    for k, v in parity.items():
        t = v['training_forward']
        log(f"  {k}: outputs ({v['precision']}) nll_rel {v['nll_rel']:.2e} mean_rel_l2 {v['mean_rel_l2']:.2e} logits_rel_l2 {v['logits_rel_l2']:.2e} | "
            f"training forward ({t['precision']}) nll_rel {t['nll_rel']:.2e} mean_rel_l2 {t['mean_rel_l2']:.2e} mean_max_over_range {t['mean_max_over_range']:.2e} "
            f"logits_rel_l2 {t['logits_rel_l2']:.2e}  (means rms {v['mean_ref_rms']:.3f}, targets rms {v['y_test_rms']:.3f})")
        if 'training_forward_fp16' in v:
            t = v['training_forward_fp16']
            log(f"  {k}: training forward (fp16) nll_rel {t['nll_rel']:.2e} mean_rel_l2 {t['mean_rel_l2']:.2e} mean_max_over_range {t['mean_max_over_range']:.2e} logits_rel_l2 {t['logits_rel_l2']:.2e}")


def _criterion_from(sd):
    from transformerscandobayesianinference_amd import bar_distribution
    return bar_distribution.FullSupportBarDistribution(sd['criterion.borders'].clone())


def stage_config2(args, device, log, shape='config2'):
    from transformerscandobayesianinference_amd import evaluation
    w = SHAPES[shape]
    log(f"{shape}: {args.epochs} epochs x {args.steps_per_epoch} steps x {args.batch} datasets, lr {args.lr}, precision {args.precision}")
    model, curve, seconds = run_training(w, device, args.precision, args.epochs, args.steps_per_epoch, args.batch, args.lr, args.seed, log,
                                         aggregate=args.aggregate)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.to(device)
    positions = list(range(1, w['bptt'], 100)) + [1755, 1999]
    positions = sorted(set(positions))
    log('paired evaluation (trained)')
    trained = paired_curves(model, w, device, positions, args.eval_datasets, seed=777)
    if args.light:
        log(f"  trained  : PFN bar NLL {trained['summary']['pfn_bar_nll']:.4f}  exact GP {trained['summary']['exact_gp_nll']:.4f}  prior {trained['summary']['prior_nll']:.4f}  skipped steps {getattr(model, 'optimizer_steps_skipped', None)}")
        return dict(recipe=dict(shape=shape, epochs=args.epochs, steps_per_epoch=args.steps_per_epoch, batch_size=args.batch, lr=args.lr, precision=args.precision, seed=args.seed, tune=args.tune),
                    training_seconds=seconds, optimizer_steps_skipped=getattr(model, 'optimizer_steps_skipped', None), loss_curve=curve, paired_eval_trained=trained)
    fresh = bench.build_model(device, args.precision, w, criterion=_criterion_from(sd))
    with torch.no_grad():   # the reference's fresh model: zero-initialised residual branches (transformer.py:49-53)
        for layer in fresh.transformer_encoder.layers:
            layer.linear2.weight.zero_(); layer.self_attn.out_proj.weight.zero_()
    fresh.mark_params_updated()
    untrained = paired_curves(fresh, w, device, positions, min(args.eval_datasets, 64), seed=777)
    del fresh
    log(f"  trained  : PFN bar NLL {trained['summary']['pfn_bar_nll']:.4f}  exact GP {trained['summary']['exact_gp_nll']:.4f}  prior {trained['summary']['prior_nll']:.4f}")
    log(f"  untrained: PFN bar NLL {untrained['summary']['pfn_bar_nll']:.4f}")
    log('API sweeps (evaluation.run_test / gp_baseline)')
    seed_all(778)
    pos, mse, mode_mse, nll, conf = evaluation.run_test(model, device=str(device), step_size=200, start_pos=1, batch_size=128, sub_batch_size=64,
                                                         seq_len=w['bptt'], num_features=w['num_features'], hyperparameters=dict(HPS))
    seed_all(778)
    gpos, gmse, gnll, gconf = evaluation.gp_baseline(device=str(device), step_size=200, start_pos=1, batch_size=128, sub_batch_size=64,
                                                     seq_len=w['bptt'], num_features=w['num_features'], hyperparameters=dict(HPS))
    api = dict(positions=pos, run_test_nll=nll.tolist(), run_test_nll_conf=conf.tolist(), run_test_mean_mse=mse.tolist(), run_test_mode_mse=mode_mse.tolist(),
               gp_baseline_nll=gnll.tolist(), gp_baseline_nll_conf=gconf.tolist(), gp_baseline_mse=gmse.tolist())
    log('parity of the trained weights vs the f64 oracle')
    parity = parity_of(sd, w, device, seps=[1755, 1000])
    log_parity(parity, log)
    val = bench.validation_loss(model, w, device)
    result = dict(what=('BASELINE.json configs[1]' if shape == 'config2' else 'the GP-fitting notebook recipe (5 features) at the configs[1] model and bptt') +
                       f' trained with the HIP stack (tools/train_pfn.py --stage {shape})',
                  recipe=dict(prior='priors.fast_gp', hyperparameters=HPS, bptt=w['bptt'], num_features=w['num_features'], emsize=w['emsize'], nhead=w['nhead'],
                              nhid=w['nhid'], nlayers=w['nlayers'], bars=w['num_bars'], borders=f"get_bucket_limits(1000, ys of get_batch(100000, 20, {w['num_features']}))",
                              epochs=args.epochs, steps_per_epoch=args.steps_per_epoch, batch_size=args.batch, aggregate_k_gradients=args.aggregate, lr=args.lr,
                              warmup_epochs=args.epochs // 4, schedule='get_cosine_schedule_with_warmup, stepped per epoch (lr 0 in epoch 1, reference quirk Q5)',
                              eval_pos='get_weighted_single_eval_pos_sampler(2000)', precision=args.precision, seed=args.seed,
                              datasets=args.epochs * args.steps_per_epoch * args.batch),
                  tune=args.tune, optimizer_steps_skipped=getattr(model, 'optimizer_steps_skipped', None),
                  training_seconds=seconds, datasets_per_second=args.epochs * args.steps_per_epoch * args.batch / seconds,
                  loss_curve=curve, paired_eval_trained=trained, paired_eval_untrained=untrained, api_sweeps=api, parity_trained=parity,
                  val_bar_nll=dict(value=val['value'], datasets=val['datasets'], eval_position=val['eval_position'], seed=val['seed']))
    if args.ckpt:
        torch.save(({k: (v.to(torch.bfloat16) if v.dtype == torch.float32 and not k.startswith('criterion.') and args.ckpt_bf16 else v)
                     for k, v in sd.items()}, None), args.ckpt)
    return result


def stage_curves(args, device, log):
    w = SHAPES['config2']
    out = {}
    for prec in ('bf16', 'fp16', 'f32'):
        log(f'curves: {prec}')
        _, curve, seconds = run_training(w, device, prec, args.curve_epochs, args.curve_steps, args.curve_batch, args.lr, args.seed, log, streams=1)
        out[prec] = dict(loss=[c['loss'] for c in curve], seconds=seconds)
    a, b = np.array(out['bf16']['loss']), np.array(out['f32']['loss'])
    h = np.array(out['fp16']['loss'])
    out['fp16_vs_f32'] = dict(max_abs_diff=float(np.abs(h - b).max()), mean_abs_diff=float(np.abs(h - b).mean()))
    out['max_abs_diff'] = float(np.abs(a - b).max())
    out['mean_abs_diff'] = float(np.abs(a - b).mean())
    out['last_quarter_mean'] = dict(bf16=float(a[-len(a) // 4:].mean()), f32=float(b[-len(b) // 4:].mean()))
    out['run'] = dict(epochs=args.curve_epochs, steps_per_epoch=args.curve_steps, batch=args.curve_batch, lr=args.lr, seed=args.seed, micro_streams=1,
                      note='identical seeds: initial weights, prior draws (Philox seed + call counter) and eval positions are the same in both runs')
    log(f"  bf16 vs f32 per-epoch loss: max |diff| {out['max_abs_diff']:.4f}, mean |diff| {out['mean_abs_diff']:.4f}; last quarter {out['last_quarter_mean']}")
    return out


def stage_config1(args, device, log):
    w = SHAPES['config1']
    epochs, steps, batch, lr = args.c1_epochs, 100, args.c1_batch, args.c1_lr
    log(f'config1: {epochs} epochs x {steps} steps x {batch} datasets, lr {lr}')
    model, curve, seconds = run_training(w, device, 'bf16', epochs, steps, batch, lr, args.seed, log, streams=1)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.to(device)
    positions = [1, 11, 21, 41, 61, 81, 99]
    trained = paired_curves(model, w, device, positions, 256, seed=777)
    log(f"  trained: PFN bar NLL {trained['summary']['pfn_bar_nll']:.4f}  exact GP {trained['summary']['exact_gp_nll']:.4f}  prior {trained['summary']['prior_nll']:.4f}")
    parity = parity_of(sd, w, device, seps=[81, 50])
    log_parity(parity, log)
    if args.ckpt1:
        torch.save((sd, None), args.ckpt1)
    return dict(what='configs[0] shape trained with the HIP stack; the checkpoint is tests/golden/trained_config1.pt',
                recipe=dict(w, epochs=epochs, steps_per_epoch=steps, batch_size=batch, lr=lr, seed=args.seed, datasets=epochs * steps * batch),
                training_seconds=seconds, loss_curve=[c['loss'] for c in curve], paired_eval_trained=trained, parity_trained=parity)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage', nargs='+', default=['config2'], choices=['config2', 'notebook5', 'curves', 'config1'])
    ap.add_argument('--epochs', type=int, default=80)
    ap.add_argument('--steps-per-epoch', type=int, default=100)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--aggregate', type=int, default=1)
    ap.add_argument('--lr', type=float, default=3e-4)
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--eval-datasets', type=int, default=256)
    ap.add_argument('--curve-epochs', type=int, default=24)
    ap.add_argument('--curve-steps', type=int, default=10)
    ap.add_argument('--curve-batch', type=int, default=16)
    ap.add_argument('--c1-epochs', type=int, default=100)
    ap.add_argument('--c1-batch', type=int, default=64)
    ap.add_argument('--c1-lr', type=float, default=1e-3)
    ap.add_argument('--ckpt', default=None, help='save the config-2 state dict here (57 MB in f32)')
    ap.add_argument('--ckpt-bf16', action='store_true')
    ap.add_argument('--ckpt1', default=None, help='save the config-1 state dict here (1.3 MB)')
    ap.add_argument('--out', default='gpurun_out/r03_trained.json')
    ap.add_argument('--tune', default='', help='comma-separated key=value pairs for pfn_set_tuning (include/pfn_hip.h), e.g. 15=6: the fp16 loss-scale target')
    ap.add_argument('--light', action='store_true', help='config2 / notebook5: loss curve + paired evaluation of the trained model only')
    args = ap.parse_args()
    device = torch.device('cuda:0')
    torch.cuda.set_device(device)
    log = lambda s: print(s, flush=True)
    from transformerscandobayesianinference_amd import _hip
    for kv in filter(None, args.tune.split(',')):
        _hip.check(_hip.lib().pfn_set_tuning(*[int(v) for v in kv.split('=')]), 'pfn_set_tuning')
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    result = json.load(open(args.out)) if os.path.exists(args.out) else {}
    for stage in args.stage:
        result[stage] = {'config2': stage_config2, 'notebook5': lambda *a: stage_config2(*a, shape='notebook5'), 'curves': stage_curves,
                         'config1': stage_config1}[stage](args, device, log)
        json.dump(result, open(args.out, 'w'), indent=1)
    log(f'wrote {args.out}')


if __name__ == '__main__':
    main()
