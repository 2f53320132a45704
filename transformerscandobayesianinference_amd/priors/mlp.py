"""BNN ("MLP") prior: every synthetic dataset is the output of a small random network on Gaussian inputs.

Replaces reference priors/mlp.py `get_batch` (:62-203); the hot configuration is the NON-CAUSAL one the tabular experiments use
(`is_causal=False`, no categorical features, `pre_sample_weights=False`):
per group of `batch_size_per_gp_sample` datasets one network is drawn -- dropout share, noise std, init std, number
of used features, depth, width from the caller's samplers, in the reference's order (:87-104); every parameter
~ N(0, init_std / (1 - dropout)) times a Bernoulli(1 - dropout) mask (:129-134, the first weight matrix unmasked)
-- and each dataset is one forward on fresh N(0,1) causes with fresh additive layer noise (:137-157), followed by
normalisation over the sequence, median binarisation, order_by_y and zero padding to `num_features` (:185-201).

MI355X: the reference's Python loop of B module constructions + B forwards of ~10 tiny kernels each (:195-197)
becomes: two torch launches that draw ALL parameters of the batch into padded tensors, ONE HIP kernel for the B
forwards (`pfn_mlp_prior_forward`, csrc/mlp_prior.hip; causes and layer noise from the device generator), and a few
batched torch ops for the order statistics (median, argsort).  No CPU fallback.

Round 3 added the remaining branches of the reference's forward on top of the same kernel: the causal-graph variant (`is_causal`,
`pre_sample_causes`, `y_is_effect`, :94-104, :139-140, :158-166 -- the kernel hands out every layer's nodes, features and target are
gathered from them per dataset), categorical features (:168-179) and per-unit pre-sampled noise scales (`pre_sample_weights`,
:119-121).  `canonical_args` and `nan_prob` are ignored like in the reference (:72-81, :181-183 are commented out there).
"""
import random

import torch
from torch import nn

from transformerscandobayesianinference_amd import _hip
from transformerscandobayesianinference_amd.priors.utils import get_batch_to_dataloader
from transformerscandobayesianinference_amd.utils import default_device

_ACTIVATIONS = {nn.Identity: 0, nn.ReLU: 1, nn.Tanh: 2, nn.Sigmoid: 3}
ACT_NAMES = {0: 'identity', 1: 'relu', 2: 'tanh', 3: 'sigmoid'}
_call_counter = [0]


def activation_code(activation_module):
    for cls, code in _ACTIVATIONS.items():
        if activation_module is cls or (isinstance(activation_module, type) and issubclass(activation_module, cls)):
            return code
    raise _hip.HipExtensionError(f'priors.mlp: activation {activation_module} is not implemented by the HIP sampler '
                                 f'(Identity, ReLU, Tanh, Sigmoid); there is no PyTorch fallback')


def pack_networks(weights, biases, device):
    """weights[m][l]: [out, in] tensors (torch Linear layout), biases[m][l]: [out].  Returns the padded, transposed
    tensors the kernel consumes: (W [M, Lmax, HP, HP] as [in][out], b [M, Lmax, HP], dims [M, 3] int32, HP)."""
    M = len(weights)
    Lmax = max(len(w) for w in weights)
    width = max(max(max(t.shape) for t in w) for w in weights)
    HP = (width + 3) // 4 * 4
    W = torch.zeros(M, Lmax, HP, HP, dtype=torch.float32, device=device)
    b = torch.zeros(M, Lmax, HP, dtype=torch.float32, device=device)
    dims = torch.zeros(M, 3, dtype=torch.int32)
    for m in range(M):
        for l, (w, bb) in enumerate(zip(weights[m], biases[m])):
            W[m, l, :w.shape[1], :w.shape[0]] = w.t().to(device)
            b[m, l, :bb.shape[0]] = bb.to(device)
        dims[m, 0], dims[m, 1], dims[m, 2] = weights[m][0].shape[1], weights[m][0].shape[0], len(weights[m])
    return W, b, dims.to(device), HP


def forward_networks(W, b, dims, noise_std, model_of, seq_len, activation, causes=None, noise=None, seed=None, want_hidden=False):
    """The batched forward through the C ABI.  Returns (causes [B,T,HP], y_raw [B,T]) and, with `want_hidden`, the outputs of
    layers 1 .. L-1 [B, Lmax-1, T, HP] (noise included) -- the node pool of the causal variant."""
    dev = W.device
    if dev.type != 'cuda':
        raise _hip.HipExtensionError(f'the BNN prior sampler runs on the GPU only (got device {dev}); no CPU fallback')
    M, Lmax, HP, _ = W.shape
    B = model_of.shape[0]
    gen = causes is None
    causes = torch.empty(B, seq_len, HP, dtype=torch.float32, device=dev) if gen else causes.to(dev).float().contiguous()
    y = torch.empty(B, seq_len, dtype=torch.float32, device=dev)
    hidden = torch.zeros(B, max(Lmax - 1, 1), seq_len, HP, dtype=torch.float32, device=dev) if want_hidden else None
    if seed is None:
        seed = torch.initial_seed()
    _call_counter[0] += 1
    _hip.check(_hip.lib().pfn_mlp_prior_forward(W.data_ptr(), b.data_ptr(), model_of.data_ptr(), dims.data_ptr(), noise_std.data_ptr(),
                                                causes.data_ptr(), _hip.ptr(noise), y.data_ptr(), _hip.ptr(hidden), B, seq_len, HP, Lmax, activation,
                                                int(gen), seed & (2 ** 64 - 1), _call_counter[0], _hip.stream_ptr(dev)), 'pfn_mlp_prior_forward')
    return (causes, y, hidden) if want_hidden else (causes, y)


def causal_select(hidden, nl, hid, nfu, y_is_effect, perm=None):
    """The causal variant's node selection for ONE dataset group that shares a network (reference :158-166): the outputs of layers
    1 .. L-1 (`outputs[2:]`: every node behind the first layer, noise included) are concatenated; a random permutation of all
    nodes but the last picks the target (position 0, or the network output itself when `y_is_effect`) and the `nfu` features
    (positions 1 .. nfu).  hidden [g, Lmax-1, T, HP]; perm [g, n-1] optional (tests inject the reference's draws).
    Returns (x [g, T, nfu], y [g, T])."""
    flat = torch.cat([hidden[:, l, :, :hid] for l in range(nl - 2)] + [hidden[:, nl - 2, :, :1]], -1)       # [g, T, (L-2) hid + 1]
    g, T, n = flat.shape
    if perm is None:
        perm = torch.rand(g, n - 1, device=flat.device).argsort(1)          # one uniform permutation per dataset (reference: randperm per forward)
    pick = lambda cols: torch.gather(flat, 2, cols[:, None, :].expand(g, T, cols.shape[1]))
    y = flat[:, :, -1] if y_is_effect else pick(perm[:, 0:1])[:, :, 0]
    return pick(perm[:, 1:1 + nfu]), y


def categorical_columns(x, features, ordinal, perm=None):
    """Discretise randomly chosen columns of x [g, T, nfu] in place (reference :168-179): column perm[i] of a dataset becomes the
    number of the i-th categorical feature's thresholds (class boundaries - 0.5) its standardised value exceeds; non-ordinal
    features have their class ids scrambled by `* (127 k + 1) % k`."""
    g, T, nfu = x.shape
    if perm is None:
        perm = torch.rand(g, nfu, device=x.device).argsort(1)
    for i, (feat, is_ordinal) in enumerate(zip(features, ordinal)):
        col = perm[:, i][:, None, None].expand(g, T, 1)                      # the dataset's i-th randomly chosen column
        v = torch.gather(x, 2, col)[:, :, 0]                                 # [g, T]
        temp = (v - v.mean(1, keepdim=True)) / (v.std(1, keepdim=True) + .000001)
        thr = torch.as_tensor(feat, dtype=torch.float32, device=x.device) - 0.5
        cls = (temp[None] > thr[:, None, None]).sum(0).to(x.dtype)
        if not is_ordinal:
            cls = cls * (127 * len(feat) + 1) % len(feat)
        x.scatter_(2, col, cls[:, :, None])
    return x


def postprocess(causes, y_raw, nfu, num_features, is_binary, order_sign, nfu_scale=None):
    """Batched statement of the tail of the reference's MLP.forward (:185-201) over datasets [B, T, .]:
    normalize_data, Binarize (lower median), normalize_by_used_features, order_by_y, zero padding."""
    B, T, HP = causes.shape
    col = torch.arange(HP, device=causes.device)
    used = (col[None, :] < nfu[:, None]).to(causes.dtype)                 # [B, HP]
    x = (causes - causes.mean(1, keepdim=True)) / (causes.std(1, keepdim=True) + .000001) * used[:, None, :]
    y = (y_raw - y_raw.mean(1, keepdim=True)) / (y_raw.std(1, keepdim=True) + .000001)
    if is_binary:
        y = (y > y.median(dim=1, keepdim=True)[0]).to(x.dtype)
    if nfu_scale is not None:
        x = x / nfu_scale[:, None, None]
    if is_binary and order_sign is not None:
        order = torch.argsort(y * order_sign[:, None], dim=1)
        order = order.reshape(B, 2, T // 2).transpose(1, 2).reshape(B, T)
        x = torch.gather(x, 1, order[:, :, None].expand(B, T, HP))
        y = torch.gather(y, 1, order)
    if HP >= num_features:
        x = x[:, :, :num_features]
    else:
        x = torch.cat([x, torch.zeros(B, T, num_features - HP, dtype=x.dtype, device=x.device)], -1)
    return x, y


@torch.no_grad()
def get_batch(batch_size, seq_len, num_features, device=default_device, hyperparameters=None,
              batch_size_per_gp_sample=None, num_outputs=1, canonical_args=None, sampling='normal'):
    """Same signature and return layout as the reference: (x[T,B,F], y[T,B], y[T,B])."""
    assert num_outputs == 1
    if hyperparameters is None or len(hyperparameters) != 17:
        raise ValueError('priors.mlp.get_batch needs the 17-entry hyper-parameter tuple of the reference (priors/mlp.py:65); '
                         'its 7-entry default (:62) cannot be unpacked by the reference either')
    (num_layers_sampler, hidden_dim_sampler, activation_module, init_std_sampler, noise_std_sampler, dropout_prob_sampler,
     is_binary_classification, num_features_used_sampler, causes_sampler, is_causal, pre_sample_causes, pre_sample_weights,
     y_is_effect, order_y, normalize_by_used_features, categorical_features_sampler, nan_prob) = hyperparameters
    if sampling not in ('normal', 'uniform'):
        raise ValueError(f'Sampling is set to invalid setting: {sampling}.')
    dev = torch.device(device)
    act = activation_code(activation_module)
    sample_batch_size = batch_size
    batch_size_per_gp_sample = batch_size_per_gp_sample or sample_batch_size // 8
    assert sample_batch_size % batch_size_per_gp_sample == 0, 'Please choose a batch_size divisible by batch_size_per_gp_sample.'
    num_models = sample_batch_size // batch_size_per_gp_sample

    # per-network scalars, drawn in the reference's order (:87-104): dropout, noise std, init std, used features, categorical
    # features, [causes of the causal variant], depth, width
    cfg = []
    for _ in range(num_models):
        dropout_prob, noise_std, init_std = dropout_prob_sampler(), noise_std_sampler(), init_std_sampler()
        nfu = int(num_features_used_sampler())
        cats, ordinal = categorical_features_sampler(nfu) if categorical_features_sampler is not None else ([], [])   # None: SURVEY.md Q12
        cause_means = cause_std = None
        if is_causal:
            cause_means, cause_std = causes_sampler()                       # (:94-97) the network's inputs are these causes, not the features
        num_layers, hidden = int(num_layers_sampler()), int(hidden_dim_sampler())
        if is_causal:
            hidden = max(hidden, 2 * nfu + 1)                               # (:103-104) enough nodes to pick nfu features + 1 target from
        assert num_layers > 2
        cfg.append(dict(dropout=float(dropout_prob), noise_std=float(noise_std), init_std=float(init_std), nfu=nfu, cats=list(cats),
                        ordinal=list(ordinal), means=cause_means, std=cause_std, nl=num_layers, hid=hidden,
                        nin=len(cause_means) if is_causal else nfu))
    Lmax = max(c['nl'] for c in cfg)
    HP = (max(max(c['nin'], c['hid']) for c in cfg) + 3) // 4 * 4

    # all parameters of the batch in two launches: N(0,1) * std * Bernoulli mask, inside each network's own shape
    lay = torch.arange(Lmax, device=dev)[None, :, None, None]
    kin = torch.arange(HP, device=dev)[None, None, :, None]
    jout = torch.arange(HP, device=dev)[None, None, None, :]
    t = lambda k, dt=torch.float32: torch.tensor([c[k] for c in cfg], dtype=dt, device=dev)
    p_drop, nstd, istd = t('dropout'), t('noise_std'), t('init_std')
    nfu, nin, nl, hid = t('nfu', torch.int64), t('nin', torch.int64), t('nl', torch.int64), t('hid', torch.int64)
    in_dim = torch.where(lay == 0, nin[:, None, None, None], hid[:, None, None, None])
    out_dim = torch.where(lay == nl[:, None, None, None] - 1, torch.ones_like(hid)[:, None, None, None], hid[:, None, None, None])
    live = (lay < nl[:, None, None, None]) & (kin < in_dim) & (jout < out_dim)
    std = (istd / (1. - p_drop))[:, None, None, None]
    std_w = torch.where(lay == 0, istd[:, None, None, None], std)                                                  # ... nor rescaled (:130)
    keep = 1. - torch.where(lay == 0, torch.zeros_like(p_drop)[:, None, None, None], p_drop[:, None, None, None])   # first matrix is never masked (:131)
    W = torch.randn(num_models, Lmax, HP, HP, device=dev) * std_w * torch.bernoulli(keep.expand(num_models, Lmax, HP, HP)) * live
    keep_b = (1. - p_drop)[:, None, None]
    live_b = live[:, :, 0, :] | ((lay[:, :, 0, :] < nl[:, None, None]) & (jout[:, :, 0, :] < out_dim[:, :, 0, :]))
    b = torch.randn(num_models, Lmax, HP, device=dev) * std[:, :, 0, :] * torch.bernoulli(keep_b.expand(num_models, Lmax, HP)) * live_b
    dims = torch.stack([nin, hid, nl], 1).to(torch.int32)
    model_of = torch.arange(sample_batch_size, device=dev, dtype=torch.int32) // batch_size_per_gp_sample
    mo = model_of.long()

    # network inputs: generated in the kernel (N(0,1) in the first `nin` columns) unless they are uniform or pre-sampled causes
    causes = None
    in_cols = kin[0, 0, :, 0][None, None, :] < nin[mo][:, None, None]
    if sampling == 'uniform':
        causes = torch.rand(sample_batch_size, seq_len, HP, device=dev) * in_cols
    elif is_causal and pre_sample_causes:                                   # (:139-140) N(means, |std|) per cause, the same law for every row
        mean_t = torch.zeros(num_models, HP, device=dev)
        std_t = torch.zeros(num_models, HP, device=dev)
        for m, c in enumerate(cfg):
            mean_t[m, :c['nin']] = torch.as_tensor(c['means'], dtype=torch.float32, device=dev)
            std_t[m, :c['nin']] = torch.as_tensor(c['std'], dtype=torch.float32, device=dev).abs()
        causes = (mean_t[mo][:, None, :] + std_t[mo][:, None, :] * torch.randn(sample_batch_size, seq_len, HP, device=dev)) * in_cols
    # layer noise: a scalar std per network (generated in the kernel), or -- pre_sample_weights (:119-121) -- one |N(0, noise_std)| std per
    # unit, fixed per network: drawn here and injected already scaled
    noise = None
    nstd_k = nstd
    if pre_sample_weights:
        unit_std = (torch.randn(num_models, max(Lmax - 1, 1), HP, device=dev) * nstd[:, None, None]).abs()
        noise = torch.randn(sample_batch_size, max(Lmax - 1, 1), seq_len, HP, device=dev) * unit_std[mo][:, :, None, :]
        nstd_k = torch.ones_like(nstd)
    need_hidden = bool(is_causal)
    out = forward_networks(W, b, dims, nstd_k, model_of, seq_len, act, causes=causes, noise=noise, want_hidden=need_hidden)
    causes, y_raw = out[0], out[1]

    if is_causal:
        # features and target are nodes of the network (:158-163), chosen per dataset; groups share their network's shape
        HX = (max(c['nfu'] for c in cfg) + 3) // 4 * 4
        x_raw = torch.zeros(sample_batch_size, seq_len, max(HX, 4), device=dev)
        y_raw = y_raw.clone()
        for m, c in enumerate(cfg):
            rows = slice(m * batch_size_per_gp_sample, (m + 1) * batch_size_per_gp_sample)
            xs, ys = causal_select(out[2][rows], c['nl'], c['hid'], c['nfu'], y_is_effect)
            x_raw[rows, :, :c['nfu']] = xs
            y_raw[rows] = ys
    else:
        x_raw = causes
    if any(len(c['cats']) > 0 for c in cfg):                               # (:168-179)
        x_raw = x_raw.clone() if x_raw is causes else x_raw
        for m, c in enumerate(cfg):
            if len(c['cats']) > 0:
                rows = slice(m * batch_size_per_gp_sample, (m + 1) * batch_size_per_gp_sample)
                x_raw[rows, :, :c['nfu']] = categorical_columns(x_raw[rows, :, :c['nfu']].clone(), c['cats'], c['ordinal'])

    order_sign = None
    if is_binary_classification and order_y:
        order_sign = torch.tensor([1.0 if random.randint(0, 1) else -1.0 for _ in range(sample_batch_size)], device=dev)
    nfu_b = nfu[mo]
    scale = (nfu_b.float() / num_features) if normalize_by_used_features else None
    x, y = postprocess(x_raw, y_raw, nfu_b, num_features, is_binary_classification, order_sign, scale)
    x, y = x.transpose(0, 1), y.transpose(0, 1)
    return x, y, y


DataLoader = get_batch_to_dataloader(get_batch)
DataLoader.num_outputs = 1
DataLoader.prefetch = True
