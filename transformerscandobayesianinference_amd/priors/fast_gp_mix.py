"""Mixture-of-GPs prior: every dataset draws its own kernel hyper-parameters from Gamma hyper-priors, then
y_b ~ N(0, outputscale_b * Matern_{5/2}(x_b, x_b; lengthscale_b[ARD]) + noise_b * I).

Replaces reference priors/fast_gp_mix.py `get_batch` (:58-134), whose model is botorch's SingleTaskGP with
gpytorch priors sampled through pyro (`get_model(...).pyro_sample_from_prior()`, :24-55): with a batched
train-x of shape [n, T, F] every one of the n datasets gets an independent draw of
    lengthscale_d ~ Gamma(lengthscale_concentration 3.0, rate 6.0)   per feature (ARD), :33
    outputscale   ~ Gamma(outputscale_concentration 0.5, rate 0.15)  :37
    noise         ~ Gamma(noise_concentration 1.1, rate 0.05), floored at botorch's MIN_INFERRED_NOISE_LEVEL 1e-4  :26-35
and the sample is `likelihood(model(x)).sample()` in prior mode (:96-99).  Here the hyper-parameters are drawn
with torch on the device and the whole batch goes through ONE call of the HIP sampler
(`pfn_gp_prior_sample`, Matern-5/2 ARD Gram -> blocked Cholesky -> L z) with per-dataset hyper-parameters;
the reference's Python loop over groups of `batch_size_per_gp_sample` datasets (:87) only exists to bound
gpytorch's memory and has no effect on the distribution, so the argument is accepted and ignored except for the
divisibility check.  `y_minmax_norm`, `sigmoid` (:100-103) and the `fix_to_range` rejection step (:104-122) are
kept.  Always exact Cholesky; nu in {0.5, 1.5, 2.5} (gpytorch MaternKernel's closed forms; the reference default is 2.5, :40).

Model fitting / MCMC comparison code of the reference file (get_fitted_model, get_mcmc_model, evaluate_, :156-287)
needs gpytorch / botorch / pyro and is out of scope.
"""

import torch
from torch import nn

from transformerscandobayesianinference_amd import _hip
from transformerscandobayesianinference_amd.bar_distribution import BarDistribution
from transformerscandobayesianinference_amd.priors import fast_gp
from transformerscandobayesianinference_amd.priors.utils import get_batch_to_dataloader
from transformerscandobayesianinference_amd.utils import default_device

MIN_INFERRED_NOISE_LEVEL = 1e-4   # botorch.models.gp_regression (reference :10,31)


def sample_hyperparameters(n, num_features, hyperparameters, device, generator=None):
    """Per-dataset draws of (lengthscale[n,F], outputscale[n], noise[n]) from the Gamma hyper-priors (reference :26,33,37)."""
    hp = hyperparameters or {}

    def gamma(concentration, rate, shape):
        c = torch.full(shape, float(concentration), device=device)
        if generator is None:
            return torch._standard_gamma(c) / float(rate)
        return torch._standard_gamma(c.cpu(), generator=generator).to(device) / float(rate)

    ls = gamma(hp.get('lengthscale_concentration', 3.0), hp.get('lengthscale_rate', 6.0), (n, num_features))
    osc = gamma(hp.get('outputscale_concentration', .5), hp.get('outputscale_rate', 0.15), (n,))
    nz = gamma(hp.get('noise_concentration', 1.1), hp.get('noise_rate', 0.05), (n,)).clamp_min(MIN_INFERRED_NOISE_LEVEL)
    # a Gamma(0.5, .) outputscale has mass at 0+: keep the Gram matrix positive definite in f32
    return ls.clamp_min(1e-6), osc.clamp_min(1e-12), nz


@torch.no_grad()
def get_batch(batch_size, seq_len, num_features, device=default_device, hyperparameters=None,
              batch_size_per_gp_sample=None, num_outputs=1, fix_to_range=None, equidistant_x=False):
    """Same signature and return layout as the reference: (x[T,B,F], y[T,B], target_y[T,B])."""
    assert num_outputs == 1
    hyperparameters = hyperparameters or {}
    nu = float(hyperparameters.get('nu', 2.5))
    if nu not in fast_gp.MATERN_KERNEL_OF_NU:
        raise ValueError(f'priors.fast_gp_mix: Matern nu must be 0.5, 1.5 or 2.5 (gpytorch MaternKernel, reference :40), got {nu}')
    matern = fast_gp.MATERN_KERNEL_OF_NU[nu]
    if batch_size_per_gp_sample is not None:   # grouping has no effect here (one batched sampler call); keep the reference's check (:77)
        assert batch_size % batch_size_per_gp_sample == 0
    factor = 2 if fix_to_range is not None else 1
    out_x, out_y = [], []
    need = batch_size
    attempts = 0
    while need > 0:
        n = need * factor                                  # the reference draws 2x candidates under fix_to_range (:81-82)
        x = None
        if equidistant_x:
            assert num_features == 1
            x = torch.linspace(0, 1., seq_len).unsqueeze(0).repeat(n, 1).unsqueeze(-1)
        ls, osc, nz = sample_hyperparameters(n, num_features, hyperparameters, device)
        # anything below that copies y (normalisation, rejection) must see a verified draw: check at once there; a plain
        # draw inside a prefetching loader is verified before its batch is handed out (priors/utils.py)
        copies = bool(hyperparameters.get('y_minmax_norm') or hyperparameters.get('sigmoid') or fix_to_range is not None)
        x, y, _, _ = fast_gp.gp_sample(n, seq_len, num_features, device, ls, osc, nz, matern, x=x,
                                       check='sync' if copies else True)
        if hyperparameters.get('y_minmax_norm'):
            lo, hi = y.min(1, keepdim=True)[0], y.max(1, keepdim=True)[0]
            y = (y - lo) / (hi - lo)
        if hyperparameters.get('sigmoid'):
            y = y.sigmoid()
        if fix_to_range is not None:
            ok = ~((y < fix_to_range[0]) | (y >= fix_to_range[1])).any(1)
            x, y = x[ok][:need], y[ok][:need]
            attempts += 1
            if attempts > 100:
                raise RuntimeError('priors.fast_gp_mix: fix_to_range rejects (almost) every draw; change the hyper-parameters '
                                   '(e.g. decrease the outputscale) -- the reference prints this advice and loops forever')
        out_x.append(x)
        out_y.append(y)
        need -= x.shape[0]
    x = torch.cat(out_x, 0) if len(out_x) > 1 else out_x[0]
    y = torch.cat(out_y, 0) if len(out_y) > 1 else out_y[0]
    sample = y.transpose(0, 1)
    return x.transpose(0, 1), sample, sample


class DataLoader(get_batch_to_dataloader(get_batch)):
    num_outputs = 1
    prefetch = True
    prefetch_group = 10
    prefetch_group_datasets = 640
    prefetch_memory_share = 0.125          # as priors.fast_gp: a group's factorisation workspace stays inside this share of free memory
    prefetch_bytes_per_dataset = staticmethod(fast_gp.workspace_bytes_per_dataset)     # from the library (pfn_gp_workspace_bytes), not a constant factor

    @torch.no_grad()
    def validate(self, model, step_size=1, start_pos=0):
        """Mean-squared error of the posterior-predictive mean at every evaluation position on one fresh batch
        (reference :139-153); forward-only passes through the HIP stack."""
        if isinstance(model.criterion, BarDistribution):
            (x, y), target_y = self.gbm(**self.get_batch_kwargs, fuse_x_y=self.fuse_x_y)
            model.eval()
            losses = []
            for eval_pos in range(start_pos, len(x), step_size):
                logits = model((x, y), single_eval_pos=eval_pos)
                means = model.criterion.mean(logits)  # num_evals x batch_size
                losses.append(nn.functional.mse_loss(means[0], target_y[eval_pos]))
            model.train()
            return torch.stack(losses)
        return 123.


class _PriorDraw:
    """What `likelihood(model(x))` is in the reference's sampling path (:96-99): the prior-predictive distribution of y at x under ONE draw of
    the hyper-parameters per dataset.  `.sample()` is the HIP sampler (Gram -> blocked Cholesky -> L z, csrc/gp_prior.hip)."""

    def __init__(self, model, x, with_noise):
        self.model, self.x, self.with_noise = model, x, with_noise

    @torch.no_grad()
    def sample(self):
        m = self.model
        x = self.x if self.x.dim() == 3 else self.x.unsqueeze(0)
        noise = m.noise if self.with_noise else torch.full_like(m.noise, 1e-9)      # the latent f: no observation noise (a jitter keeps f32 positive definite)
        _, y, _, _ = fast_gp.gp_sample(x.shape[0], x.shape[1], x.shape[2], x.device, m.lengthscale, m.outputscale, noise, m.kernel, x=x.float(), check='sync')
        return y


class _SampledLikelihood:
    """GaussianLikelihood of a sampled model: adds the sampled observation noise to the prior at x."""

    def __init__(self, model):
        self.model = model
        self.noise = model.noise

    def __call__(self, prior):
        return _PriorDraw(prior.model, prior.x, with_noise=True)


class SampledGP:
    """`get_model(x, y, hyperparameters, sample=True)[0]` (reference :24-55: `SingleTaskGP(...).pyro_sample_from_prior()`): a GP whose kernel
    hyper-parameters were DRAWN from the Gamma hyper-priors, one independent draw per dataset of the batch.  Exposes the sampled
    `lengthscale [n, F]` (ARD), `outputscale [n]`, `noise [n]` (floored at botorch's MIN_INFERRED_NOISE_LEVEL) and `nu`; `model(x)` is the
    prior at x, `likelihood(model(x)).sample()` the draw `get_batch` takes ([n, T], :96-99)."""

    def __init__(self, x, hyperparameters):
        hp = hyperparameters or {}
        x = x if x.dim() == 3 else x.unsqueeze(0)
        self.nu = float(hp.get('nu', 2.5))
        if self.nu not in fast_gp.MATERN_KERNEL_OF_NU:
            raise ValueError(f'priors.fast_gp_mix: Matern nu must be 0.5, 1.5 or 2.5 (gpytorch MaternKernel, reference :40), got {self.nu}')
        self.kernel = fast_gp.MATERN_KERNEL_OF_NU[self.nu]
        self.lengthscale, self.outputscale, self.noise = sample_hyperparameters(x.shape[0], x.shape[-1], hp, x.device)
        self.likelihood = _SampledLikelihood(self)

    def __call__(self, x):
        return _PriorDraw(self, x, with_noise=False)

    def to(self, device):
        self.lengthscale, self.outputscale, self.noise = self.lengthscale.to(device), self.outputscale.to(device), self.noise.to(device)
        return self


def get_model(x, y, hyperparameters: dict, sample=True):
    """Reference :24-55.  sample=True (the only form `get_batch` uses, :95): (sampled model, its likelihood), hyper-parameters drawn per dataset of the
    batched x [n, T, F]; `y` is ignored as in the reference's prior mode (it passes an empty tensor).  sample=False builds the botorch model for
    FITTING (get_fitted_model, :156-171), which needs gpytorch / botorch and is outside this path."""
    if not sample:
        raise NotImplementedError('priors.fast_gp_mix.get_model(sample=False) is the botorch SingleTaskGP for hyper-parameter FITTING (reference :156-171); '
                                  'fitting / MCMC baselines need gpytorch / botorch and are outside the MI355X hot path (SURVEY.md 2)')
    model = SampledGP(x, hyperparameters)
    return model, model.likelihood
