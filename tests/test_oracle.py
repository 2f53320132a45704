"""CPU tests (no GPU): the oracle (oracle/pfn_oracle.py) against the golden vectors recorded from the
reference itself (oracle/make_golden.py -> tests/golden), plus algebraic properties the reference
implies.  This is what pins the checker that the GPU parity tests rely on."""
import math
import os

import pytest
import torch

from oracle import pfn_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def relerr(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('case', ['model_small_h32', 'model_small_h64'])
def test_oracle_forward_loss_grads_match_reference(case):
    rec = torch.load(os.path.join(GOLD, case + '.pt'))
    cfg, sd = rec['config'], rec['state_dict']
    for sep, want in rec['per_sep'].items():
        loss, logits, grads = pfn_oracle.loss_and_grads(sd, rec['x'], rec['y'], rec['target_y'], sep, cfg['H'], sd['criterion.borders'])
        assert relerr(logits, want['logits']) < 2e-5          # reference ran in f32, oracle in f64
        assert abs(loss.item() - want['loss'].item()) < 2e-5 * abs(want['loss'].item())
        losses = pfn_oracle.bar_nll(logits.reshape(-1, cfg['nbars']), rec['target_y'][sep:].reshape(-1), sd['criterion.borders'])
        assert relerr(losses.view(want['losses'].shape), want['losses']) < 2e-5
        assert relerr(pfn_oracle.bar_mean(logits, sd['criterion.borders']), want['mean']) < 2e-5
        if 'grads' in want:
            for k, g in want['grads'].items():
                if g.norm() > 1e-7:
                    assert relerr(grads[k], g) < 5e-4, (sep, k, relerr(grads[k], g))


def test_oracle_training_steps_match_reference():
    rec = torch.load(os.path.join(GOLD, 'model_small_h32.pt'))
    cfg, tr = rec['config'], rec['train']
    params = {k: v.double().clone() for k, v in rec['state_dict'].items() if not k.startswith('criterion.')}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    borders = rec['state_dict']['criterion.borders']
    for step in range(2):
        loss, _, grads = pfn_oracle.loss_and_grads(params, rec['x'], rec['y'], rec['target_y'], tr['sep'], cfg['H'], borders)
        norm = pfn_oracle.clip_adam_step(params, grads, m, v2, step + 1, 1e-3)
        assert abs(loss.item() - tr['steps'][step]['loss'].item()) < 5e-5 * abs(loss.item())
        assert abs(norm.item() - tr['steps'][step]['grad_norm'].item()) < 1e-4 * norm.item()
    # Adam divides each element by its own gradient history: elements whose gradient sits at the f32
    # rounding-noise level of the reference move differently, so compare the update as a whole.
    num = den = 0.0
    for k, v in params.items():
        d_ref = tr['final_state_dict'][k].double() - rec['state_dict'][k].double()
        num += ((v - rec['state_dict'][k].double() - d_ref) ** 2).sum().item()
        den += (d_ref ** 2).sum().item()
        assert (v - tr['final_state_dict'][k].double()).abs().max().item() < 2e-4, k   # well inside one step (lr = 1e-3)
    assert math.sqrt(num / den) < 1e-2


def test_oracle_bar_distribution_matches_reference():
    rec = torch.load(os.path.join(GOLD, 'bar_distribution.pt'))
    for key, c in rec.items():
        if not isinstance(key, tuple):
            continue
        nb, full = key
        logits = c['logits'].double().requires_grad_(True)
        nll = pfn_oracle.bar_nll(logits, c['y'], c['borders'], full)
        assert (nll - c['nll'].double()).abs().max().item() < 2e-5, key
        (nll * c['w'].double()).sum().backward()
        assert relerr(logits.grad, c['dlogits']) < 1e-5
        assert relerr(pfn_oracle.bar_mean(logits.detach(), c['borders'], full), c['mean']) < 1e-5
        assert torch.equal(pfn_oracle.bar_bucket(c['borders'], c['y']), c['bucket'])


def test_bar_density_integrates_to_one():
    """Uniform logits give density 1/(num_bars * width_k) inside bucket k (bar_distribution.py:30-33)."""
    borders = torch.tensor([-1.0, -0.5, 0.25, 0.5, 2.0], dtype=torch.float64)
    logits = torch.zeros(1, 4, dtype=torch.float64)
    ys = torch.linspace(-0.999, 1.999, 30001, dtype=torch.float64)
    dens = torch.exp(-pfn_oracle.bar_nll(logits.expand(len(ys), 4), ys, borders, full_support=False))
    assert abs(torch.trapz(dens, ys).item() - 1.0) < 2e-3
    mid = torch.tensor([-0.75], dtype=torch.float64)
    assert abs(torch.exp(-pfn_oracle.bar_nll(logits, mid, borders, False)).item() - 1 / (4 * 0.5)) < 1e-12


def test_mask_properties_of_the_oracle():
    """Test-row outputs ignore other test rows and the order of the train rows (transformer.py:34-41, no positional encoding)."""
    rec = torch.load(os.path.join(GOLD, 'model_small_h32.pt'))
    cfg, sd, x, y = rec['config'], rec['state_dict'], rec['x'], rec['y']
    sep = 41
    base = pfn_oracle.forward(sd, x, y, sep, cfg['H'])
    perm = torch.randperm(sep, generator=torch.Generator().manual_seed(0))
    xp, yp = x.clone(), y.clone()
    xp[:sep], yp[:sep] = x[perm], y[perm]
    assert relerr(pfn_oracle.forward(sd, xp, yp, sep, cfg['H']), base) < 1e-12
    x2 = x.clone()
    x2[sep + 3] += 1.0
    other = pfn_oracle.forward(sd, x2, y, sep, cfg['H'])
    keep = torch.ones(cfg['T'] - sep, dtype=torch.bool)
    keep[3] = False
    assert relerr(other[keep], base[keep]) < 1e-12 and relerr(other[3], base[3]) > 1e-6
    y2 = y.clone()
    y2[sep:] += 5.0   # test rows never see their own y (transformer.py:73-74)
    assert relerr(pfn_oracle.forward(sd, x, y2, sep, cfg['H']), base) < 1e-12
    assert torch.equal(pfn_oracle.d_q_mask(6, 4, torch.float32, 'cpu'), torch.load(os.path.join(GOLD, 'utils.pt'))['d_q_mask_6_2'])


def test_gp_oracle_covariance():
    """Sample covariance of the restated GP draw equals outputscale*RBF + noise*I (fast_gp.py:13-32,53-56)."""
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 12, 2, generator=g).expand(20000, 12, 2)
    z = torch.randn(20000, 12, generator=g)
    y = pfn_oracle.gp_sample(x, z, 0.7, 1.3, 0.2)
    emp = y.t() @ y / 20000
    K = pfn_oracle.gp_gram(x[:1].double(), torch.tensor(0.7).reshape(1, 1, 1), torch.tensor(1.3).reshape(1, 1, 1), torch.tensor(0.2).reshape(1, 1, 1))[0]
    assert (emp - K).abs().max().item() < 0.05
    assert abs(K[0, 0].item() - 1.5) < 1e-6
    xs, ys, ts = pfn_oracle.get_batch_fast_gp(4, 10, 3, (0.1, 0.1, 0.1), g)
    assert xs.shape == (10, 4, 3) and ys.shape == (10, 4) and ts.shape == (10, 4)


def test_gp_oracle_matches_reference_sklearn_gp():
    """The pin of the GP oracle: tests/golden/gp_sklearn.pt holds what the REFERENCE's sklearn GP (priors/gp.py) produced on
    fixed-seed inputs (oracle/make_golden.py::gp_case) -- the RBF Gram of `get_gp(ls).kernel` (:14-17), the per-position
    losses of `evaluate` (:41-62; alpha = 1e-10 as noise, latent predictive variance), and sklearn's Matern-5/2 ARD Gram."""
    rec = torch.load(os.path.join(GOLD, 'gp_sklearn.pt'))
    for c in rec['cases']:
        one = torch.ones(1, 1, 1, dtype=torch.float64)
        gram = pfn_oracle.gp_gram(c['x'], one * c['length_scale'], one, 0 * one, 'rbf')
        assert (gram - c['gram']).abs().max().item() < 1e-9
        xt, yt = c['x'].transpose(0, 1), c['y'].transpose(0, 1)
        hps = dict(noise=rec['alpha'], outputscale=1.0, lengthscale=c['length_scale'])
        nll, _, var = pfn_oracle.gp_evaluate(xt, yt, hyperparameters=hps, min_noise=0.0, noisy_predictive=False)
        assert var.min().item() > 1e-6        # nn.GaussianNLLLoss clamps the variance at 1e-6 (priors/gp.py:54): not active in these cases
        want = c['evaluate_nll'][1:]          # the reference returns [0.] + the batch mean per position
        assert (nll.mean(1) - want).abs().max().item() < 1e-9        # measured 4e-14
        mse, _, _ = pfn_oracle.gp_evaluate(xt, yt, use_mse=True, hyperparameters=hps, min_noise=0.0, noisy_predictive=False)
        assert (mse.mean(1) - c['evaluate_mse'][1:]).abs().max().item() < 1e-9
    for c in rec['matern']:
        one = torch.ones(1, 1, 1, dtype=torch.float64)
        gram = pfn_oracle.gp_gram(c['x'].unsqueeze(0), c['lengthscale'].reshape(1, 1, -1), one, 0 * one, 'matern')[0]
        assert (gram - c['gram']).abs().max().item() < 1e-9
        for nu, name in ((0.5, 'matern12'), (1.5, 'matern32'), (2.5, 'matern52')):      # hyperparameters['nu'] of priors/fast_gp_mix.py:40
            gram = pfn_oracle.gp_gram(c['x'].unsqueeze(0), c['lengthscale'].reshape(1, 1, -1), one, 0 * one, name)[0]
            assert (gram - c['gram_nu'][nu]).abs().max().item() < 1e-9


def test_gp_evaluate_oracle_chain_rule_and_noise_limit():
    """The restated priors.fast_gp.evaluate (one exact GP per position): its per-position negative log densities are
    the chain-rule factors of the joint Gaussian, so with the prior term of position 0 they add up to
    -log N(y; 0, outputscale k + noise I); and with a huge lengthscale-free noise the prediction is the prior."""
    g = torch.Generator().manual_seed(5)
    T, B, F = 40, 3, 2
    hps = (0.05, 0.9, 0.4)
    x, y, _ = pfn_oracle.get_batch_fast_gp(B, T, F, hps, g, dtype=torch.float64)
    nll, mean, var = pfn_oracle.gp_evaluate(x, y, hyperparameters=hps)
    assert nll.shape == (T - 1, B) and (var > 0).all()
    t64 = lambda v: torch.tensor(v, dtype=torch.float64).reshape(1, 1, 1)
    C = pfn_oracle.gp_gram(x.transpose(0, 1).double(), t64(0.4), t64(0.9), t64(0.05))
    yb = y.transpose(0, 1).double()
    joint = -torch.distributions.MultivariateNormal(torch.zeros(B, T, dtype=torch.float64), covariance_matrix=C).log_prob(yb)
    first = 0.5 * (math.log(2 * math.pi) + torch.log(C[:, 0, 0]) + yb[:, 0] ** 2 / C[:, 0, 0])
    assert torch.allclose(nll.sum(0) + first, joint, rtol=1e-9, atol=1e-9)
    mse, _, _ = pfn_oracle.gp_evaluate(x, y, use_mse=True, hyperparameters=hps, step_size=7, start_pos=3)
    assert mse.shape == (len(range(3, T, 7)), B)
    assert torch.allclose(mse[0], (mean[2] - yb[:, 3]) ** 2)
    # observation noise >> signal: the posterior mean collapses to the prior mean 0, the variance to outputscale + noise
    _, m2, v2 = pfn_oracle.gp_evaluate(x, y, hyperparameters=(1e6, 0.9, 0.4))
    assert m2.abs().max().item() < 1e-4 and torch.allclose(v2, torch.full_like(v2, 1e6 + 0.9), rtol=1e-6)


def test_oracle_mlp_prior_matches_reference():
    """priors.mlp.get_batch of the reference (non-causal tabular BNN prior) re-built by the oracle from the tensors the
    reference itself drew (tests/golden/mlp_prior.pt, recorded by oracle/make_golden.py::mlp_prior_case)."""
    rec = torch.load(os.path.join(GOLD, 'mlp_prior.pt'))
    cfg = rec['config']
    T, B, NF, PER = cfg['T'], cfg['B'], cfg['NF'], cfg['PER']
    for i in range(B):
        m = i // PER
        W = [rec['params'][6 * m + 2 * l] for l in range(3)]
        b = [rec['params'][6 * m + 2 * l + 1] for l in range(3)]
        causes, n1, n2 = [t[:, 0, :] for t in rec['normals'][3 * i: 3 * i + 3]]
        y_raw = pfn_oracle.mlp_prior_forward(W, b, causes, [n1, n2], cfg['activation'])
        sign = 1.0 if rec['coins'][i] else -1.0
        x, y = pfn_oracle.mlp_prior_postprocess(causes, y_raw, NF, binary=True, order_sign=sign)
        assert torch.equal(y, rec['y'][:, i]), i
        assert torch.allclose(x, rec['x'][:, i, :], atol=1e-5, rtol=1e-5), i
        assert x[:, causes.shape[1]:].abs().max() == 0


def test_torch_modules_model_matches_the_port():
    """oracle/torch_modules.py (the nn.TransformerEncoder stack the reference instantiates: bench.py's cpu_baseline leg) against the
    explicit-math port and the reference-generated logits, on a golden state dict."""
    from oracle import torch_modules
    rec = torch.load(os.path.join(GOLD, 'model_small_h32.pt'))
    cfg, sd = rec['config'], rec['state_dict']
    m = torch_modules.from_state_dict(sd, cfg['H']).train()
    for sep, want in rec['per_sep'].items():
        got = m((rec['x'], rec['y']), sep)
        assert relerr(got.detach(), want['logits']) < 2e-5
        port = pfn_oracle.forward({k: v for k, v in sd.items() if not k.startswith('criterion.')}, rec['x'], rec['y'], sep, cfg['H'])
        assert relerr(got.detach(), port) < 2e-5
    assert set(m.state_dict()) == {k for k in sd if not k.startswith('criterion.')}


def test_top_layer_train_rows_feed_nothing_in_the_reference_stack():
    """What licenses the HIP stack to run its top encoder layer on the test rows only (pfn_api.hip top_layer_on_test_rows): in the module stack the
    reference instantiates, everything the TOP layer computes for its train rows behind the K / V projection -- attention output, out_proj, both
    LayerNorms, FFN -- reaches neither the returned logits (output[single_eval_pos:], transformer.py:91) nor any parameter gradient.  Checked on the
    torch modules themselves: the top layer's train-row outputs are replaced by garbage (so is their gradient path), the train-row queries of its
    attention see garbage too; logits and every gradient stay what they were.  The K / V projection of the train rows IS needed -- removing it
    changes the result."""
    from oracle import torch_modules
    rec = torch.load(os.path.join(GOLD, 'model_small_h32.pt'))
    cfg, sd = rec['config'], rec['state_dict']
    x, y = rec['x'], rec['y']
    sep = sorted(rec['per_sep'])[len(rec['per_sep']) // 2]
    assert 0 < sep < len(x)

    def run(mangle):
        m = torch_modules.from_state_dict(sd, cfg['H']).train()
        top = m.transformer_encoder.layers[-1]
        handles = []
        if mangle in ('outputs', 'queries'):
            # the top layer's train-row outputs replaced by garbage without a gradient path
            handles.append(top.register_forward_hook(lambda mod, inp, out: torch.cat([torch.full_like(out[:sep], 1e3).detach(), out[sep:]], 0)))
        if mangle == 'queries':
            # ... and its attention's train-row QUERIES too (keys and values untouched): the self-attention module is called as attn(x, x, x)
            def pre(mod, args, kwargs):
                q, k, v = args[:3]
                return (torch.cat([torch.full_like(q[:sep], -7.0), q[sep:]], 0), k, v) + tuple(args[3:]), kwargs
            handles.append(top.self_attn.register_forward_pre_hook(pre, with_kwargs=True))
        if mangle == 'keys':
            def pre(mod, args, kwargs):
                q, k, v = args[:3]
                return (q, torch.cat([torch.zeros_like(k[:sep]), k[sep:]], 0), v) + tuple(args[3:]), kwargs
            handles.append(top.self_attn.register_forward_pre_hook(pre, with_kwargs=True))
        out = m((x, y), sep)
        out.square().mean().backward()
        for h in handles:
            h.remove()
        return out.detach(), {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}

    base_out, base_grads = run(None)
    for mangle in ('outputs', 'queries'):
        out, grads = run(mangle)
        assert torch.equal(out, base_out), mangle
        for k, g in base_grads.items():
            assert torch.allclose(grads[k], g, rtol=1e-5, atol=1e-7 * float(g.abs().max() + 1e-30)), (mangle, k)
    out, _ = run('keys')
    assert relerr(out, base_out) > 1e-3


def test_oracle_mlp_prior_causal_matches_reference():
    """The remaining branches of the reference's priors.mlp forward -- causal graph with pre-sampled causes, categorical features,
    per-unit pre-sampled noise scales -- re-built by the oracle from the tensors the reference drew (tests/golden/mlp_prior_causal.pt,
    oracle/make_golden.py::mlp_prior_causal_case)."""
    rec = torch.load(os.path.join(GOLD, 'mlp_prior_causal.pt'))
    cfg = rec['config']
    T, B, NF, PER, NFU = cfg['T'], cfg['B'], cfg['NF'], cfg['PER'], cfg['NFU']
    M = B // PER
    for i in range(B):
        m = i // PER
        W = [rec['params'][6 * m + 2 * l] for l in range(3)]
        b = [rec['params'][6 * m + 2 * l + 1] for l in range(3)]
        causes, n1, n2 = [t[:, 0, :].float() for t in rec['normals'][2 * M + 3 * i: 2 * M + 3 * i + 3]]     # (2 M construction draws: the noise scales; the causes are drawn in f64 and cast, :140)
        outs = pfn_oracle.mlp_prior_layers(W, b, causes, [n1, n2], cfg['activation'])
        x_sel, y_raw = pfn_oracle.mlp_prior_causal_select(outs, rec['perms'][2 * i], NFU)
        x_cat = pfn_oracle.mlp_prior_categorical(x_sel, cfg['cats'], cfg['ordinal'], rec['perms'][2 * i + 1])
        sign = 1.0 if rec['coins'][i] else -1.0
        x, y = pfn_oracle.mlp_prior_postprocess(x_cat, y_raw, NF, binary=True, order_sign=sign)
        assert torch.equal(y, rec['y'][:, i]), i
        assert torch.allclose(x, rec['x'][:, i, :], atol=1e-5, rtol=1e-5), i
