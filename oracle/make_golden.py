"""Generates tests/golden/*.pt by running the REFERENCE ITSELF (imported from /root/reference) on
fixed-seed inputs.  Test infrastructure; run in the build container only (the reference does not
travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/make_golden.py

The reference imports gpytorch in priors/__init__.py:1 (absent here), so a stub `priors` package
is registered first; only reference modules that import cleanly are used:
transformer.TransformerModel, bar_distribution.*, encoders, positional_encodings, utils, train, priors.mlp,
priors.gp (the reference's sklearn statement of the RBF GP: pins the oracle's Gram matrix and exact-GP evaluation).
Weights: the reference zero-initialises out_proj / linear2 (transformer.py:49-53), which would make
attention and the MLP invisible, so those tensors are re-drawn N(0, 0.05) before recording.
"""
import os
import random
import sys
import types

import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def import_reference():
    sys.path.insert(0, REF)
    pk = types.ModuleType('priors')
    pk.__path__ = [os.path.join(REF, 'priors')]
    sys.modules['priors'] = pk
    import bar_distribution, encoders, positional_encodings, train, transformer, utils  # noqa
    return dict(bar_distribution=bar_distribution, encoders=encoders, positional_encodings=positional_encodings,
                train=train, transformer=transformer, utils=utils)


def gp_draw(B, T, F, gen, ls=0.6, os_=1.0, noise=1e-4):
    """Plain-torch statement of priors/fast_gp.py:41-58 (gpytorch absent): f64 Cholesky draw."""
    x = torch.rand(B, T, F, generator=gen)
    z = torch.randn(B, T, generator=gen)
    xs = x.double() / ls
    d2 = (xs.unsqueeze(2) - xs.unsqueeze(1)).pow(2).sum(-1)
    K = os_ * torch.exp(-0.5 * d2) + noise * torch.eye(T, dtype=torch.float64)
    y = (torch.linalg.cholesky(K) @ z.double().unsqueeze(-1)).squeeze(-1).float()
    return x.transpose(0, 1).contiguous(), y.transpose(0, 1).contiguous(), z


def model_case(ref, name, T, B, F, E, H, nhid, L, nbars, seps, seed):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    x, y, _ = gp_draw(B, T, F, gen)
    ys_for_borders = gp_draw(200, 20, F, gen)[1]
    borders = ref['bar_distribution'].get_bucket_limits(nbars, ys=ys_for_borders)
    criterion = ref['bar_distribution'].FullSupportBarDistribution(borders)
    model = ref['transformer'].TransformerModel(ref['encoders'].Linear(F, E), nbars, E, H, nhid, L, 0.0,
                                                y_encoder=ref['encoders'].Linear(1, E),
                                                pos_encoder=ref['positional_encodings'].NoPositionalEncoding(E, T * 2))
    model.criterion = criterion
    with torch.no_grad():
        for layer in model.transformer_encoder.layers:
            for t in (layer.linear2.weight, layer.linear2.bias, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias):
                t.normal_(0, 0.05)
            for t in (layer.norm1.weight, layer.norm2.weight):
                t.add_(0.1 * torch.randn_like(t))
            for t in (layer.norm1.bias, layer.norm2.bias, layer.self_attn.in_proj_bias):
                t.normal_(0, 0.05)
    model.train()
    rec = dict(config=dict(T=T, B=B, F=F, E=E, H=H, nhid=nhid, L=L, nbars=nbars), x=x, y=y, target_y=y.clone(),
               state_dict={k: v.clone() for k, v in model.state_dict().items()}, per_sep={})
    for sep in seps:
        model.zero_grad()
        logits = model((x, y), single_eval_pos=sep)                       # reference forward, transformer.py:55-91
        targets = y[sep:]
        losses = criterion(logits.reshape(-1, nbars), targets.flatten()).view(*logits.shape[:2])   # train.py:88-89
        loss = losses.mean()
        loss.backward()
        rec['per_sep'][sep] = dict(logits=logits.detach().clone(), losses=losses.detach().clone(), loss=loss.detach().clone(),
                                   mean=criterion.mean(logits.detach()).clone())
        if sep in seps[:2]:  # gradients for the first two positions only (fixture size)
            rec['per_sep'][sep]['grads'] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    # two optimisation steps exactly as train.py:92-97 (Adam lr 1e-3, clip 1.0), at seps[0]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    traj = []
    for step in range(2):
        opt.zero_grad()
        logits = model((x, y), single_eval_pos=seps[0])
        losses = criterion(logits.reshape(-1, nbars), y[seps[0]:].flatten())
        loss = losses.mean()
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.)
        opt.step()
        traj.append(dict(loss=loss.detach().clone(), grad_norm=norm.detach().clone()))
    rec['train'] = dict(sep=seps[0], steps=traj, final_state_dict={k: v.clone() for k, v in model.state_dict().items()})
    torch.save(rec, os.path.join(OUT, f'{name}.pt'))
    print(name, 'losses', {s: float(r['loss']) for s, r in rec['per_sep'].items()}, 'train', [float(t['loss']) for t in traj])


def bar_case(ref):
    gen = torch.Generator().manual_seed(7)
    rec = {}
    for nb, full in [(10, True), (100, True), (1000, True), (10, False), (100, False)]:
        ys = torch.randn(nb * 50, generator=gen)
        borders = ref['bar_distribution'].get_bucket_limits(nb, ys=ys)
        cls = ref['bar_distribution'].FullSupportBarDistribution if full else ref['bar_distribution'].BarDistribution
        crit = cls(borders)
        R = 64
        logits = torch.randn(R, nb, generator=gen) * 2
        lo, hi = borders[0].item(), borders[-1].item()
        y = torch.rand(R, generator=gen) * (hi - lo) + lo
        if full:
            y[:8] = lo - torch.rand(8, generator=gen) * 2
            y[8:16] = hi + torch.rand(8, generator=gen) * 2
        y[16], y[17], y[18] = borders[0], borders[-1], borders[nb // 2]
        logits.requires_grad_(True)
        nll = crit(logits, y)
        w = torch.rand(R, generator=gen)
        (nll * w).sum().backward()
        rec[(nb, full)] = dict(borders=borders, logits=logits.detach().clone(), y=y, nll=nll.detach().clone(), w=w,
                               dlogits=logits.grad.clone(), mean=crit.mean(logits.detach()).clone(),
                               bucket=crit.map_to_bucket_idx(y).clamp(0, nb - 1))
        # the evaluation-side methods (bar_distribution.py:40-80), from the same logits (no further random draws)
        lg = logits.detach()
        best_f = borders[nb // 3].item()
        rec[(nb, full)].update(quantile=crit.quantile(lg).clone(), quantile90=crit.quantile(lg, center_prob=.9).clone(),
                               mode=crit.mode(lg).clone(), best_f=best_f, ei_max=crit.ei(lg, best_f, maximize=True).clone(),
                               ei_min=crit.ei(lg, best_f, maximize=False).clone())
    rec['bucket_limits_uniform'] = ref['bar_distribution'].get_bucket_limits(8, full_range=(-2., 6.))
    torch.save(rec, os.path.join(OUT, 'bar_distribution.pt'))
    print('bar cases', list(k for k in rec if isinstance(k, tuple)))


def utils_case(ref):
    u = ref['utils']
    p = torch.nn.Parameter(torch.zeros(1))
    rec = {}
    for name, fn, args in [('cosine', u.get_cosine_schedule_with_warmup, (5, 20)), ('linear', u.get_linear_schedule_with_warmup, (5, 20))]:
        opt = torch.optim.SGD([p], lr=1.0)
        sch = fn(opt, *args)
        vals = []
        for _ in range(22):
            vals.append(sch.get_last_lr()[0])
            opt.step()
            sch.step()
        rec[name] = vals
    random.seed(123)
    s = u.get_weighted_single_eval_pos_sampler(50)
    rec['weighted_draws'] = [s() for _ in range(200)]
    random.seed(123)
    s = u.get_uniform_single_eval_pos_sampler(50)
    rec['uniform_draws'] = [s() for _ in range(200)]
    m = torch.nn.Linear(10, 10)
    rec['openai_lr_110'] = u.get_openai_lr(m)
    rec['d_q_mask_6_2'] = ref['transformer'].TransformerModel.generate_D_q_matrix(6, 2)
    torch.save(rec, os.path.join(OUT, 'utils.pt'))
    print('utils ok')


def mlp_prior_case(ref):
    """Runs the reference's priors.mlp.get_batch (non-causal tabular configuration) and records,
    by intercepting the torch calls it makes, every tensor a restatement needs to reproduce its output: the initialised
    parameters of each model, the cause / noise draws of each dataset in call order, the order_by_y coin flips."""
    import numpy as np
    from priors import mlp as ref_mlp
    from priors import utils as ref_putils
    T, B, NF, PER = 48, 6, 7, 3
    hps = (lambda: 3, ref_putils.scaled_beta_sampler_f(2., 4., 20, 2), torch.nn.Tanh,
           ref_putils.gamma_sampler_f(3.6187797729244253, 0.06773738681062867), ref_putils.gamma_sampler_f(1.8663049257557085, 0.05275478076173361),
           lambda: 0.0, True, ref_putils.scaled_beta_sampler_f(1., 1.6, NF, 2), None, False, None, None, None, True, False, lambda n: ([], []), 0.0)
    torch.manual_seed(21); random.seed(21); np.random.seed(21)
    rec = dict(config=dict(T=T, B=B, NF=NF, PER=PER, activation='tanh'), params=[], normals=[], coins=[])
    orig_init, orig_normal, orig_randint = torch.nn.init.normal_, torch.normal, random.randint

    def init_normal(p, mean=0., std=1.):
        out = orig_init(p, mean=mean, std=std)
        rec['params'].append(p.detach().clone())
        return out

    def normal(*a, **k):
        out = orig_normal(*a, **k)
        rec['normals'].append(out.detach().clone())
        return out

    def randint(a, b):
        v = orig_randint(a, b)
        rec['coins'].append(v)
        return v

    torch.nn.init.normal_, torch.normal, random.randint = init_normal, normal, randint
    try:
        x, y, _ = ref_mlp.get_batch(B, T, NF, device='cpu', hyperparameters=hps, batch_size_per_gp_sample=PER)
    finally:
        torch.nn.init.normal_, torch.normal, random.randint = orig_init, orig_normal, orig_randint
    rec['x'], rec['y'] = x.clone(), y.clone()
    torch.save(rec, os.path.join(OUT, 'mlp_prior.pt'))
    print('mlp prior', x.shape, y.shape, 'params', len(rec['params']), 'normals', len(rec['normals']), 'coins', rec['coins'], 'y mean', float(y.mean()))


def mlp_prior_causal_case(ref):
    """The OTHER branches of the reference's priors.mlp forward, recorded the same way: the causal-graph variant with pre-sampled
    causes (:94-104, :139-140, :158-166), categorical features -- one ordinal, one not (:168-179) -- and per-unit pre-sampled
    noise scales (`pre_sample_weights`, :119-121).  Intercepted: parameter initialisations, every torch.normal (construction: the
    noise-scale vectors; forward: causes, then each layer's noise), every torch.randperm (node selection, then categorical
    columns), the order_by_y coins."""
    import numpy as np
    from priors import mlp as ref_mlp
    T, B, NF, PER, NFU, HID = 40, 4, 6, 2, 4, 7
    cats = ([np.array([0.2, 0.7, 0.4]), np.array([0.6, 0.1])], [True, False])
    cause_law = (np.array([0.5, -1.0, 0.2, 1.5, 0.0]), np.array([1.0, 0.3, 2.0, 0.7, 1.2]))
    hps = (lambda: 3, lambda: HID, torch.nn.Tanh, lambda: 0.8, lambda: 0.05, lambda: 0.0, True, lambda: NFU,
           lambda: cause_law, True, True, True, False, True, False, lambda n: cats, 0.0)
    torch.manual_seed(33); random.seed(33); np.random.seed(33)
    rec = dict(config=dict(T=T, B=B, NF=NF, PER=PER, NFU=NFU, hidden=max(HID, 2 * NFU + 1), activation='tanh', cats=[c.tolist() for c in cats[0]], ordinal=cats[1]),
               params=[], normals=[], perms=[], coins=[])
    orig_init, orig_normal, orig_randint, orig_randperm = torch.nn.init.normal_, torch.normal, random.randint, torch.randperm

    def init_normal(p, mean=0., std=1.):
        out = orig_init(p, mean=mean, std=std)
        rec['params'].append(p.detach().clone())
        return out

    def normal(*a, **k):
        out = orig_normal(*a, **k)
        rec['normals'].append(out.detach().clone())
        return out

    def randint(a, b):
        v = orig_randint(a, b)
        rec['coins'].append(v)
        return v

    def randperm(*a, **k):
        out = orig_randperm(*a, **k)
        rec['perms'].append(out.detach().clone())
        return out

    torch.nn.init.normal_, torch.normal, random.randint, torch.randperm = init_normal, normal, randint, randperm
    try:
        x, y, _ = ref_mlp.get_batch(B, T, NF, device='cpu', hyperparameters=hps, batch_size_per_gp_sample=PER)
    finally:
        torch.nn.init.normal_, torch.normal, random.randint, torch.randperm = orig_init, orig_normal, orig_randint, orig_randperm
    rec['x'], rec['y'] = x.clone(), y.clone()
    torch.save(rec, os.path.join(OUT, 'mlp_prior_causal.pt'))
    print('mlp prior (causal + categorical + pre-sampled noise scales)', x.shape, 'params', len(rec['params']), 'normals', len(rec['normals']),
          [tuple(n.shape) for n in rec['normals'][:7]], 'perms', [len(p) for p in rec['perms']], 'coins', rec['coins'])


def gp_case(ref):
    """Pins the GP part of the oracle to the reference's OWN sklearn statement of the same GP (priors/gp.py): the Gram
    matrix of `get_gp(length_scale).kernel` (:14-17, RBF with fixed length scale, unit output scale) and the per-position
    losses of `evaluate` (:41-62: a new regressor fitted on the first t points, `predict(return_std=True)` at point t,
    Gaussian NLL with `full=True` / squared error) on fixed-seed inputs, in f64.  sklearn's regressor adds alpha = 1e-10 to
    the diagonal of the training covariance and reports the LATENT predictive variance (no noise term at the test point).
    For priors.fast_gp_mix (:28-34) the Matern-5/2 ARD Gram of sklearn's `Matern(nu=2.5, length_scale=<vector>)` -- the
    same closed form gpytorch's MaternKernel(nu=2.5, ard_num_dims=F) evaluates -- is recorded as well."""
    from priors import gp as ref_gp
    from sklearn.gaussian_process.kernels import Matern
    gen = torch.Generator().manual_seed(31)
    rec = dict(alpha=1e-10, cases=[])
    for (B, T, F, ls) in [(3, 24, 5, 0.6), (2, 40, 18, 0.6), (2, 16, 2, 0.25)]:
        x = torch.rand(B, T, F, generator=gen, dtype=torch.float64)
        z = torch.randn(B, T, generator=gen, dtype=torch.float64)
        gram = torch.stack([torch.from_numpy(ref_gp.get_gp(ls).kernel(x[b].numpy())) for b in range(B)])       # priors/gp.py:14-17
        y = (torch.linalg.cholesky(gram + 1e-6 * torch.eye(T, dtype=torch.float64)) @ z.unsqueeze(-1)).squeeze(-1)
        xt, yt = x.transpose(0, 1).contiguous(), y.transpose(0, 1).contiguous()
        nll, _ = ref_gp.evaluate(xt, yt, yt, use_mse=False, length_scale=ls)                                    # :41-62
        mse, _ = ref_gp.evaluate(xt, yt, yt, use_mse=True, length_scale=ls)
        rec['cases'].append(dict(B=B, T=T, F=F, length_scale=ls, x=x, y=y, gram=gram, evaluate_nll=nll.double(), evaluate_mse=mse.double()))
    rec['matern'] = []
    for (T, F) in [(20, 5), (12, 18)]:
        x = torch.rand(T, F, generator=gen, dtype=torch.float64)
        ls = torch.rand(F, generator=gen, dtype=torch.float64) * 2 + 0.1
        rec['matern'].append(dict(x=x, lengthscale=ls, gram=torch.from_numpy(Matern(nu=2.5, length_scale=ls.numpy())(x.numpy())),
                                  gram_nu={nu: torch.from_numpy(Matern(nu=nu, length_scale=ls.numpy())(x.numpy())) for nu in (0.5, 1.5, 2.5)}))
    torch.save(rec, os.path.join(OUT, 'gp_sklearn.pt'))
    print('gp cases', [(c['B'], c['T'], c['F'], c['length_scale'], [round(float(v), 4) for v in c['evaluate_nll'][:4]]) for c in rec['cases']])


class ReplayLoader:
    """PriorDataLoader protocol (reference priors/prior.py:4-12, priors/utils.py:14-42) over RECORDED batches: what `train.train` iterates
    once per epoch.  The cursor runs on across epochs, so epoch e sees batches [e * num_steps, (e + 1) * num_steps).  tests/ holds the same
    class (tests/replay.py) so the HIP `train()` replays the identical stream."""
    num_outputs = 1
    fuse_x_y = False

    def __init__(self, num_steps, batch_size=None, seq_len=None, batches=None, **_):
        self.num_steps, self.batches, self.cursor = num_steps, batches, 0
        self.num_features = batches[0][0].shape[-1]

    def __len__(self):
        return self.num_steps

    def __iter__(self):
        for _ in range(self.num_steps):
            x, y, target = self.batches[self.cursor % len(self.batches)]
            self.cursor += 1
            yield (x, y), target


def train_loop_case(ref, name='train_loop_small', T=60, B=3, F=5, E=64, H=2, nhid=128, L=2, nbars=50, epochs=4, steps_per_epoch=8, aggregate_k=2,
                    warmup_epochs=1, lr=1e-3, seed=17):
    """Pins the TRAINING LOOP to the reference's own `train.train` (train.py:22-135): recorded batches and a recorded
    eval-position stream are replayed through it on the CPU -- `aggregate_k_gradients` micro-batches summed per optimizer step (:92-97),
    clip-to-1 + Adam, the per-EPOCH cosine schedule with warm-up whose first epoch runs at lr = 0 (utils.py:10-22: LambdaLR(step 0) = 0 / warmup)
    -- and every batch loss, the learning rate of every batch, the per-epoch mean losses and the final state dict are recorded.  Dropout 0
    (torch's dropout stream cannot be replayed)."""
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    nb = epochs * steps_per_epoch
    batches = []
    for _ in range(nb):
        x, y, _ = gp_draw(B, T, F, gen)
        batches.append((x, y, y.clone()))
    random.seed(seed)
    sampler = ref['utils'].get_weighted_single_eval_pos_sampler(T)
    seps = [sampler() for _ in range(nb)]
    borders = ref['bar_distribution'].get_bucket_limits(nbars, ys=gp_draw(200, 20, F, gen)[1])
    log = dict(loss=[], lr=[])
    state = {}

    class RecordingFullSupportBarDistribution(ref['bar_distribution'].FullSupportBarDistribution):      # ("BarDistribution" stays in the class name: train.py:37)
        def forward(self, logits, y):
            losses = super().forward(logits, y)
            log['loss'].append(float(losses.detach().mean()))
            log['lr'].append(state['opt'].param_groups[0]['lr'])
            return losses

    criterion = RecordingFullSupportBarDistribution(borders)
    # initial weights: built the way train() builds the model, with the residual branches un-zeroed (transformer.py:49-53 zeroes them)
    init = ref['transformer'].TransformerModel(ref['encoders'].Linear(F, E), nbars, E, H, nhid, L, 0.0, y_encoder=ref['encoders'].Linear(1, E),
                                               pos_encoder=ref['positional_encodings'].NoPositionalEncoding(E, T * 2))
    init.criterion = ref['bar_distribution'].FullSupportBarDistribution(borders)
    with torch.no_grad():
        for layer in init.transformer_encoder.layers:
            for t in (layer.linear2.weight, layer.self_attn.out_proj.weight):
                t.normal_(0, 0.05)
    init_sd = {k: v.clone() for k, v in init.state_dict().items()}
    it = iter(seps)

    def scheduler(optimizer, warmup, total):
        state['opt'] = optimizer
        return ref['utils'].get_cosine_schedule_with_warmup(optimizer, warmup, total)

    total_loss, positional, model = ref['train'].train(
        ReplayLoader, criterion, ref['encoders'].Linear, emsize=E, nhid=nhid, nlayers=L, nhead=H, dropout=0.0, epochs=epochs,
        steps_per_epoch=steps_per_epoch, batch_size=B, bptt=T, lr=lr, warmup_epochs=warmup_epochs, y_encoder_generator=ref['encoders'].Linear,
        extra_prior_kwargs_dict=dict(batches=batches), scheduler=scheduler, load_weights_from_this_state_dict=init_sd,
        single_eval_pos_gen=lambda: next(it), aggregate_k_gradients=aggregate_k, verbose=False)
    assert len(log['loss']) == nb
    epoch_losses = [sum(log['loss'][e * steps_per_epoch:(e + 1) * steps_per_epoch]) / steps_per_epoch for e in range(epochs)]
    assert abs(epoch_losses[-1] - total_loss) < 1e-6
    rec = dict(config=dict(T=T, B=B, F=F, E=E, H=H, nhid=nhid, L=L, nbars=nbars, epochs=epochs, steps_per_epoch=steps_per_epoch,
                           aggregate_k_gradients=aggregate_k, warmup_epochs=warmup_epochs, lr=lr),
               batches=[(x, y) for x, y, _ in batches], seps=seps, borders=borders, init_state_dict=init_sd,
               batch_losses=log['loss'], batch_lr=log['lr'], epoch_losses=epoch_losses, returned_total_loss=total_loss,
               returned_positional_losses=positional, final_state_dict={k: v.clone() for k, v in model.state_dict().items()})
    torch.save(rec, os.path.join(OUT, f'{name}.pt'))
    print(name, 'epoch losses', [round(v, 5) for v in epoch_losses], 'lr per epoch', log['lr'][::steps_per_epoch], 'seps', seps[:8])


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    torch.set_num_threads(4)
    model_case(ref, 'model_small_h32', T=60, B=3, F=5, E=64, H=2, nhid=128, L=2, nbars=50, seps=[41, 0, 59, 7], seed=11)
    model_case(ref, 'model_small_h64', T=100, B=4, F=5, E=128, H=2, nhid=72, L=1, nbars=100, seps=[70, 33], seed=12)
    bar_case(ref)
    utils_case(ref)
    mlp_prior_case(ref)
    mlp_prior_causal_case(ref)
    gp_case(ref)
    train_loop_case(ref)
