"""CPU oracle (test infrastructure only; see oracle/__init__.py).

Explicit-math restatement, in plain PyTorch on CPU tensors (f64 by default), of
  * TransformerModel.forward                      reference transformer.py:55-91
    with nn.TransformerEncoderLayer (post-norm, GELU-erf, eps 1e-5)
                                                  torch nn/modules/transformer.py:952-982
    and multi_head_attention_forward              torch nn/functional.py:6435, 5964-5968, 6637
    under the mask of generate_D_q_matrix         reference transformer.py:34-41
  * (FullSupport)BarDistribution.forward / mean   reference bar_distribution.py:19-38, 83-117
  * one optimisation step of train()              reference train.py:88-97 (mean loss, backward,
                                                  clip_grad_norm_(1.0), Adam)
  * priors.fast_gp.get_batch                      reference priors/fast_gp.py:13-58 (gpytorch
    semantics restated: K = outputscale * exp(-0.5 |dx/l|^2) + noise I, y = chol(K) z)
  * priors.fast_gp_mix kernel                     reference priors/fast_gp_mix.py:28-47 (Matern-5/2 ARD)
  * priors.fast_gp.evaluate                       reference priors/fast_gp.py:88-120 (one exact GP per position;
    gpytorch's ExactGP prediction equations restated)
The GP parts are pinned to the reference's own sklearn statement of the same RBF GP (priors/gp.py:14-17, 41-62) and to
sklearn's Matern-5/2 ARD Gram by tests/golden/gp_sklearn.pt; what stays unpinned is gpytorch-specific behaviour only
(hyper-parameter tuple conventions, the jitter ladder of a failed factorisation) -- the library is not installed.
Nothing here calls nn.TransformerEncoder: the layer math is written out, and is pinned against the
real reference modules by tests/golden (oracle/make_golden.py).
"""
import math

import torch


def _linear(x, w, b):
    return x @ w.t() + b


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def d_q_mask(T, sep, dtype, device):
    """Additive 0/-inf mask: key j visible to query i iff j < sep or i == j (transformer.py:34-41)."""
    allowed = torch.zeros(T, T, dtype=torch.bool, device=device)
    allowed[:, :sep] = True
    allowed |= torch.eye(T, dtype=torch.bool, device=device)
    return torch.zeros(T, T, dtype=dtype, device=device).masked_fill(~allowed, float('-inf'))


# ---- dropout masks: the integers of csrc/pfn_kernels.h (mix32 / dropout_site_seed / dropout_pair_seed / dropout_keep), in numpy ----
_M32 = 0xFFFFFFFF


def _mix32(h):
    import numpy as np
    h = np.asarray(h, dtype=np.uint64) & _M32
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x7feb352d)) & _M32
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x846ca68b)) & _M32
    h ^= h >> np.uint64(16)
    return h


def dropout_site_seed(seed, layer, site):
    return int(_mix32(int(_mix32((seed & _M32) ^ ((0x9E3779B1 * (layer * 4 + site + 1)) & _M32))) ^ (seed >> 32)))


def dropout_keep_mask(site_seed, rows, cols, p):
    """keep[i, j] (bool tensor [len(rows), len(cols)]) for integer index vectors rows / cols; p as the kernels see it (float32)."""
    import numpy as np
    thr = min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF)
    i = (np.asarray(rows, dtype=np.uint64) * np.uint64(0x9E3779B1)) & _M32
    j = (np.asarray(cols, dtype=np.uint64) * np.uint64(0x85EBCA77)) & _M32
    h = _mix32(np.uint64(site_seed) ^ i[:, None] ^ j[None, :])
    return torch.from_numpy(h >= np.uint64(thr))


def forward(sd, x, y, sep, nhead, dtype=torch.float64, return_hidden=False, dropout=None):
    """logits[T-sep, B, n_out] from a reference-format state dict `sd` (keys as in SURVEY.md 8(b)).
    dropout = (p, seed): TransformerEncoderLayer's four dropout sites (attention probabilities; after out_proj; after the FFN
    activation; after linear2 -- torch nn/modules/transformer.py, functional.py multi_head_attention_forward) with the masks the HIP
    stack generates from `seed` (element-wise sites: row = b * T + t, column = feature; attention: (query, key) per (dataset, head))."""
    p = {k: v.detach().to(dtype) if not v.requires_grad else v.to(dtype) for k, v in sd.items()}
    x, y = x.to(dtype), y.to(dtype)
    T, B, _ = x.shape
    if sep < 0:
        sep += T
    x_emb = _linear(x, p['encoder.weight'], p['encoder.bias'])                    # transformer.py:68
    y_emb = _linear(y.unsqueeze(-1), p['y_encoder.weight'], p['y_encoder.bias'])  # :69
    h = torch.cat([x_emb[:sep] + y_emb[:sep], x_emb[sep:]], 0)                    # :73-74
    E = h.shape[-1]
    D = E // nhead
    mask = d_q_mask(T, sep, dtype, h.device)
    nlayers = 1 + max(int(k.split('.')[2]) for k in p if k.startswith('transformer_encoder.layers.'))
    for l in range(nlayers):
        pre = f'transformer_encoder.layers.{l}.'
        qkv = _linear(h, p[pre + 'self_attn.in_proj_weight'], p[pre + 'self_attn.in_proj_bias'])   # [T,B,3E]
        q, k, v = [t.reshape(T, B, nhead, D).permute(1, 2, 0, 3) for t in qkv.split(E, -1)]        # [B,H,T,D]
        scores = q @ k.transpose(-1, -2) / math.sqrt(D) + mask
        probs = torch.softmax(scores, -1)
        if dropout is not None:
            pd, seed = dropout
            import numpy as np
            keep_scale = 1.0 / (1.0 - float(np.float32(pd)))
            s0 = dropout_site_seed(seed, l, 0)
            pm = torch.stack([torch.stack([dropout_keep_mask(int(_mix32(s0 ^ ((0xC2B2AE35 * (bb * nhead + hh + 1)) & _M32))), range(T), range(T), pd)
                                           for hh in range(nhead)]) for bb in range(B)])                       # [B,H,T,T]
            probs = probs * pm.to(dtype) * keep_scale
            rows = [bb * T + t for t in range(T) for bb in range(B)]                                        # token row of element [t, b]
            elem = lambda site, width: dropout_keep_mask(dropout_site_seed(seed, l, site), rows, range(width), pd).reshape(T, B, width).to(dtype) * keep_scale
        ctx = (probs @ v).permute(2, 0, 1, 3).reshape(T, B, E)
        att = _linear(ctx, p[pre + 'self_attn.out_proj.weight'], p[pre + 'self_attn.out_proj.bias'])
        if dropout is not None:
            att = att * elem(1, E)
        h = _layer_norm(h + att, p[pre + 'norm1.weight'], p[pre + 'norm1.bias'])
        act = _gelu(_linear(h, p[pre + 'linear1.weight'], p[pre + 'linear1.bias']))
        if dropout is not None:
            act = act * elem(2, act.shape[-1])
        ff = _linear(act, p[pre + 'linear2.weight'], p[pre + 'linear2.bias'])
        if dropout is not None:
            ff = ff * elem(3, E)
        h = _layer_norm(h + ff, p[pre + 'norm2.weight'], p[pre + 'norm2.bias'])
    if return_hidden:
        return h
    out = _linear(_gelu(_linear(h, p['decoder.0.weight'], p['decoder.0.bias'])), p['decoder.2.weight'], p['decoder.2.bias'])  # :85
    return out[sep:]                                                                                                          # :91


# ---- bar distribution -----------------------------------------------------------------------------
_HN_ICDF_HALF = 0.6744897501960817  # HalfNormal(1).icdf(0.5) = Normal.icdf(0.75)


def bar_bucket(borders, y):
    """map_to_bucket_idx (bar_distribution.py:19-23) + the clamp of the full-support variant (:92)."""
    nb = len(borders) - 1
    t = torch.searchsorted(borders, y) - 1
    t[y == borders[0]] = 0
    t[y == borders[-1]] = nb - 1
    return t.clamp(0, nb - 1)


def bar_nll(logits, y, borders, full_support=True):
    """FullSupportBarDistribution.forward (bar_distribution.py:89-108) / BarDistribution.forward (:25-33)."""
    dtype = logits.dtype
    borders = borders.to(dtype)
    y = y.to(dtype)
    nb = len(borders) - 1
    widths = borders[1:] - borders[:-1]
    t = bar_bucket(borders, y)
    lp = (torch.log_softmax(logits, -1) - torch.log(widths)).gather(-1, t.unsqueeze(-1)).squeeze(-1)
    if full_support:
        def halfnormal_logprob(scale, v):  # torch.distributions.HalfNormal(scale).log_prob(v), v >= 0
            return math.log(2.0) - torch.log(scale) - 0.5 * math.log(2 * math.pi) - 0.5 * (v / scale) ** 2
        s0, s1 = widths[0] / _HN_ICDF_HALF, widths[-1] / _HN_ICDF_HALF
        first, last = (t == 0), (t == nb - 1)
        lp = lp + first * (halfnormal_logprob(s0, (borders[1] - y).clamp(min=1e-8)) + torch.log(widths[0]))
        lp = lp + last * (halfnormal_logprob(s1, (y - borders[-2]).clamp(min=0)) + torch.log(widths[-1]))
    return -lp


def bar_mean(logits, borders, full_support=True):
    """FullSupportBarDistribution.mean (bar_distribution.py:110-117) / BarDistribution.mean (:35-38)."""
    borders = borders.to(logits.dtype)
    widths = borders[1:] - borders[:-1]
    means = borders[:-1] + widths / 2
    if full_support:
        means = means.clone()
        means[0] = borders[1] - widths[0] / _HN_ICDF_HALF * math.sqrt(2 / math.pi)
        means[-1] = borders[-2] + widths[-1] / _HN_ICDF_HALF * math.sqrt(2 / math.pi)
    return torch.softmax(logits, -1) @ means


# ---- one training step ------------------------------------------------------------------------------
def loss_and_grads(sd, x, y, target_y, sep, nhead, borders, full_support=True, dtype=torch.float64, dropout=None):
    """mean bar-NLL over the test rows and d loss / d parameter for every state-dict tensor
    (train.py:70-93 with criterion = FullSupportBarDistribution)."""
    leaves = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items() if not k.startswith('criterion.')}
    logits = forward(leaves, x, y, sep, nhead, dtype, dropout=dropout)
    T = x.shape[0]
    s = sep if sep >= 0 else sep + T
    losses = bar_nll(logits.reshape(-1, logits.shape[-1]), target_y[s:].reshape(-1), borders, full_support).view(logits.shape[:2])
    loss = losses.mean()
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in leaves.items()}


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
    """clip_grad_norm_(params, max_norm) then torch.optim.Adam.step() (train.py:95-96), dict-of-tensors in place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
    coef = min(1.0, max_norm / (total.item() + 1e-6))
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    for k in params:
        g = grads[k] * coef
        exp_avg[k].mul_(betas[0]).add_(g, alpha=1 - betas[0])
        exp_avg_sq[k].mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = exp_avg_sq[k].sqrt() / math.sqrt(bc2) + eps
        params[k].addcdiv_(exp_avg[k], denom, value=-lr / bc1)
    return total


# ---- GP priors ----------------------------------------------------------------------------------------
def gp_gram(x, lengthscale, outputscale, noise, kernel='rbf'):
    """x [B,T,F]; lengthscale broadcastable to [B,1,F]; outputscale / noise broadcastable to [B,1,1].
    RBF: ScaleKernel(RBFKernel) of priors/fast_gp.py:17 (k = exp(-0.5 |dx/l|^2)); Matern-5/2 ARD:
    priors/fast_gp_mix.py:28-34; + GaussianLikelihood noise on the diagonal."""
    xs = x / lengthscale
    d2 = (xs.unsqueeze(2) - xs.unsqueeze(1)).pow(2).sum(-1).clamp_min(0)
    if kernel == 'rbf':
        k = torch.exp(-0.5 * d2)
    elif kernel in ('matern', 'matern52'):
        r = torch.sqrt(5.0 * d2)
        k = (1 + r + r * r / 3) * torch.exp(-r)
    elif kernel == 'matern32':
        r = torch.sqrt(3.0 * d2)
        k = (1 + r) * torch.exp(-r)
    elif kernel == 'matern12':
        k = torch.exp(-torch.sqrt(d2))
    else:
        raise ValueError(kernel)
    eye = torch.eye(x.shape[1], dtype=x.dtype, device=x.device)
    return outputscale * k + noise * eye


def gp_sample(x, z, lengthscale, outputscale, noise, kernel='rbf', dtype=torch.float64):
    """y = chol(K) z  -- the draw of `likelihood(model(x)).sample()` in prior mode (fast_gp.py:53-56)
    for given uniform features x [B,T,F] and base normals z [B,T]."""
    x, z = x.to(dtype), z.to(dtype)
    B, T, F = x.shape
    as_t = lambda v: torch.as_tensor(v, dtype=dtype)
    ls = as_t(lengthscale)
    ls = ls.reshape(B, 1, -1) if ls.dim() > 0 and ls.numel() > 1 else ls.reshape(1, 1, 1)
    os_ = as_t(outputscale).reshape(-1, 1, 1)
    nz = as_t(noise).reshape(-1, 1, 1)
    K = gp_gram(x, ls, os_, nz, kernel)
    L = torch.linalg.cholesky(K)
    return (L @ z.unsqueeze(-1)).squeeze(-1)


def get_batch_fast_gp(batch_size, seq_len, num_features, hyperparameters=None, generator=None, dtype=torch.float32):
    """CPU statement of priors.fast_gp.get_batch (fast_gp.py:35-58): returns (x[T,B,F], y[T,B], y[T,B])."""
    if isinstance(hyperparameters, (tuple, list)):
        hyperparameters = {"noise": hyperparameters[0], "outputscale": hyperparameters[1], "lengthscale": hyperparameters[2]}
    elif hyperparameters is None:
        hyperparameters = {"noise": .1, "outputscale": .1, "lengthscale": .1}
    x = torch.rand(batch_size, seq_len, num_features, generator=generator)
    z = torch.randn(batch_size, seq_len, generator=generator)
    y = gp_sample(x, z, hyperparameters["lengthscale"], hyperparameters["outputscale"], max(hyperparameters["noise"], 1e-9), 'rbf', dtype)
    y = y.float()
    return x.transpose(0, 1), y.transpose(0, 1), y.transpose(0, 1)


def gp_evaluate(x, y, use_mse=False, hyperparameters=None, step_size=1, start_pos=0, kernel='rbf', dtype=torch.float64,
                min_noise=1e-9, noisy_predictive=True):
    """CPU statement of priors.fast_gp.evaluate (fast_gp.py:88-120), literally: for every t a NEW exact GP is
    conditioned on (x[:t], y[:t]) (train covariance = outputscale k + noise I, zero mean, fast_gp.py:13-32) and asked
    for the noisy predictive at x[t] (`likelihood(model(x[t]))`, :101-104); loss = -log N(y[t]; mean, var) (:115) or
    (mean - y[t])^2 (:112).  x [T,B,F], y [T,B].  Returns (losses [n_t,B], means [n_t,B], variances [n_t,B]).
    gpytorch itself is not installed here: this restates its ExactGP prediction equations,
    mean = k_t^T C_t^-1 y_t, var = k(x_t,x_t) + noise - k_t^T C_t^-1 k_t, with one Cholesky PER t -- on purpose a
    different algorithm from the product's single factorisation.
    PINNED (tests/test_oracle.py::test_gp_oracle_matches_reference_sklearn_gp) against the reference's own sklearn
    statement of the same model, priors/gp.py:41-62, whose regressor uses noise alpha = 1e-10 and reports the latent
    variance: `min_noise=0, noisy_predictive=False` selects exactly that convention."""
    if isinstance(hyperparameters, (tuple, list)):
        hyperparameters = {"noise": hyperparameters[0], "outputscale": hyperparameters[1], "lengthscale": hyperparameters[2]}
    elif hyperparameters is None:
        hyperparameters = {"noise": .1, "outputscale": .1, "lengthscale": .1}
    xb, yb = x.transpose(0, 1).to(dtype), y.transpose(0, 1).to(dtype)
    B, T, F = xb.shape
    as_t = lambda v: torch.as_tensor(v, dtype=dtype)
    ls = as_t(hyperparameters["lengthscale"])
    ls = ls.reshape(B, 1, -1) if ls.dim() > 0 and ls.numel() > 1 else ls.reshape(1, 1, 1)
    os_ = as_t(hyperparameters["outputscale"]).reshape(-1, 1, 1)
    nz = as_t(hyperparameters["noise"]).clamp_min(min_noise).reshape(-1, 1, 1)   # GaussianLikelihood(noise_constraint=GreaterThan(1e-9)), fast_gp.py:25
    C = gp_gram(xb, ls, os_, nz, kernel)                      # [B,T,T], noise on the diagonal
    losses, means, varis = [], [], []
    for t in range(max(start_pos, 1), T, step_size):
        L = torch.linalg.cholesky(C[:, :t, :t])
        k = C[:, :t, t].unsqueeze(-1)                         # cross covariance (no noise: off-diagonal)
        alpha = torch.cholesky_solve(yb[:, :t].unsqueeze(-1), L)
        v = torch.linalg.solve_triangular(L, k, upper=False)
        mean = (k * alpha).sum((1, 2))
        var = C[:, t, t] - (v * v).sum((1, 2))                # C[t,t] = outputscale k(x_t,x_t) + noise
        if not noisy_predictive:
            var = var - nz.reshape(-1)
        means.append(mean)
        varis.append(var)
        if use_mse:
            losses.append((mean - yb[:, t]) ** 2)
        else:
            losses.append(0.5 * (math.log(2 * math.pi) + torch.log(var) + (yb[:, t] - mean) ** 2 / var))
    return torch.stack(losses), torch.stack(means), torch.stack(varis)


# ---- BNN prior (priors.mlp) ----------------------------------------------------------------------------
def _activation(name):
    return {'tanh': torch.tanh, 'relu': torch.relu, 'identity': (lambda v: v), 'sigmoid': torch.sigmoid}[name]


def mlp_prior_forward(weights, biases, causes, noises, activation='tanh'):
    """One dataset of the non-causal BNN prior (reference priors/mlp.py:116-157): `causes` [T, nc] feed
    Linear(nc, h); every further layer is activation -> Linear -> additive Gaussian noise (:117-124, GaussianNoise
    :32-38).  weights[l] is [out_l, in_l] (torch Linear layout), noises[l-1] the already drawn noise of layer l >= 1.
    Returns the last layer's output [T] (y before normalisation; x is `causes`)."""
    act = _activation(activation)
    h = causes @ weights[0].t() + biases[0]
    for l in range(1, len(weights)):
        h = act(h) @ weights[l].t() + biases[l] + noises[l - 1]
    return h.squeeze(-1)


def mlp_prior_layers(weights, biases, causes, noises, activation='tanh'):
    """As `mlp_prior_forward`, but returns the outputs of layers 1 .. L-1 (noise included) -- `outputs[2:]` of the reference's
    forward (priors/mlp.py:150-157), the node pool of the causal variant."""
    act = _activation(activation)
    h = causes @ weights[0].t() + biases[0]
    outs = []
    for l in range(1, len(weights)):
        h = act(h) @ weights[l].t() + biases[l] + noises[l - 1]
        outs.append(h)
    return outs


def mlp_prior_causal_select(outs, perm, nfu, y_is_effect=False):
    """Reference priors/mlp.py:158-166 for one dataset: concatenate the layer outputs, take the target at perm[0] (or the last node when
    `y_is_effect`) and the features at perm[1 : 1 + nfu]; perm is a permutation of all nodes but the last."""
    flat = torch.cat(outs, -1)
    y = flat[:, -1] if y_is_effect else flat[:, perm[0]]
    return flat[:, perm[1:1 + nfu]], y


def mlp_prior_categorical(x, features, ordinal, perm):
    """Reference priors/mlp.py:168-179 for one dataset x [T, nfu]: column perm[i] becomes the count of feature i's thresholds
    (values - 0.5) below its standardised value; class ids of non-ordinal features are scrambled by * (127 k + 1) % k."""
    x = x.clone()
    for i, (feat, is_ordinal) in enumerate(zip(features, ordinal)):
        idx = int(perm[i])
        v = x[:, idx]
        temp = (v - v.mean(0)) / (v.std(0) + .000001)
        thr = torch.as_tensor(feat, dtype=x.dtype) - 0.5
        cls = (temp[None, :] > thr[:, None]).sum(0).to(x.dtype)
        if not is_ordinal:
            cls = cls * (127 * len(feat) + 1) % len(feat)
        x[:, idx] = cls
    return x


def mlp_prior_postprocess(x, y, num_features, binary=True, order_sign=None, nfu_scale=None):
    """x [T, nfu], y [T] of one dataset -> (x [T, num_features], y [T]) as the reference's MLP.forward does after the
    network (:185-201): normalize_data over the sequence axis (unbiased std, + 1e-6; priors/utils.py:73-78), median
    binarisation (Binarize :85-91, torch.median = lower median), optional division by the used-feature share (:191-192),
    order_by_y (:94-100; order_sign = +1 sorts by y, -1 by -y) and zero padding of the unused features (:198)."""
    norm = lambda d: (d - d.mean(0)) / (d.std(0) + .000001)
    x, y = norm(x), norm(y)
    if binary:
        y = (y > torch.median(y)).to(x.dtype)
    if nfu_scale is not None:
        x = x / nfu_scale
    if binary and order_sign is not None:
        order = torch.argsort(y * order_sign, dim=0, stable=False)
        order = order.reshape(2, -1).transpose(0, 1).reshape(-1)
        x, y = x[order], y[order]
    pad = torch.zeros(x.shape[0], num_features - x.shape[1], dtype=x.dtype)
    return torch.cat([x, pad], -1), y
