"""BNN ("MLP") prior: every synthetic dataset is the output of a small random network on Gaussian inputs.

Replaces reference priors/mlp.py `get_batch` (:62-203) for the NON-CAUSAL configuration the tabular experiments use
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

Not implemented (raise): the causal graph variant (`is_causal`, :146-166), categorical features (:168-179),
`pre_sample_weights`, `sampling='uniform'` is supported, `canonical_args` is ignored like in the reference (:72-81).
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


def forward_networks(W, b, dims, noise_std, model_of, seq_len, activation, causes=None, noise=None, seed=None):
    """The batched forward through the C ABI.  Returns (causes [B,T,HP], y_raw [B,T])."""
    dev = W.device
    if dev.type != 'cuda':
        raise _hip.HipExtensionError(f'the BNN prior sampler runs on the GPU only (got device {dev}); no CPU fallback')
    M, Lmax, HP, _ = W.shape
    B = model_of.shape[0]
    gen = causes is None
    causes = torch.empty(B, seq_len, HP, dtype=torch.float32, device=dev) if gen else causes.to(dev).float().contiguous()
    y = torch.empty(B, seq_len, dtype=torch.float32, device=dev)
    if seed is None:
        seed = torch.initial_seed()
    _call_counter[0] += 1
    _hip.check(_hip.lib().pfn_mlp_prior_forward(W.data_ptr(), b.data_ptr(), model_of.data_ptr(), dims.data_ptr(), noise_std.data_ptr(),
                                                causes.data_ptr(), _hip.ptr(noise), y.data_ptr(), B, seq_len, HP, Lmax, activation, int(gen),
                                                seed & (2 ** 64 - 1), _call_counter[0], _hip.stream_ptr(dev)), 'pfn_mlp_prior_forward')
    return causes, y


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
    if is_causal or pre_sample_weights:
        raise NotImplementedError('priors.mlp: the causal-graph / pre-sampled-weight variants are outside the MI355X hot path')
    if sampling not in ('normal', 'uniform'):
        raise ValueError(f'Sampling is set to invalid setting: {sampling}.')
    dev = torch.device(device)
    act = activation_code(activation_module)
    sample_batch_size = batch_size
    batch_size_per_gp_sample = batch_size_per_gp_sample or sample_batch_size // 8
    assert sample_batch_size % batch_size_per_gp_sample == 0, 'Please choose a batch_size divisible by batch_size_per_gp_sample.'
    num_models = sample_batch_size // batch_size_per_gp_sample

    # per-network scalars, drawn in the reference's order (:87-104)
    cfg = []
    for _ in range(num_models):
        dropout_prob, noise_std, init_std = dropout_prob_sampler(), noise_std_sampler(), init_std_sampler()
        nfu = int(num_features_used_sampler())
        if categorical_features_sampler is not None:
            cats, _ = categorical_features_sampler(nfu)
            if len(cats) > 0:
                raise NotImplementedError('priors.mlp: categorical features are outside the MI355X hot path')
        num_layers, hidden = int(num_layers_sampler()), int(hidden_dim_sampler())
        assert num_layers > 2
        cfg.append((float(dropout_prob), float(noise_std), float(init_std), nfu, num_layers, hidden))
    Lmax = max(c[4] for c in cfg)
    HP = (max(max(c[3], c[5]) for c in cfg) + 3) // 4 * 4

    # all parameters of the batch in two launches: N(0,1) * std * Bernoulli mask, inside each network's own shape
    lay = torch.arange(Lmax, device=dev)[None, :, None, None]
    kin = torch.arange(HP, device=dev)[None, None, :, None]
    jout = torch.arange(HP, device=dev)[None, None, None, :]
    t = lambda i, dt=torch.float32: torch.tensor([c[i] for c in cfg], dtype=dt, device=dev)
    p_drop, nstd, istd, nfu, nl, hid = t(0), t(1), t(2), t(3, torch.int64), t(4, torch.int64), t(5, torch.int64)
    in_dim = torch.where(lay == 0, nfu[:, None, None, None], hid[:, None, None, None])
    out_dim = torch.where(lay == nl[:, None, None, None] - 1, torch.ones_like(hid)[:, None, None, None], hid[:, None, None, None])
    live = (lay < nl[:, None, None, None]) & (kin < in_dim) & (jout < out_dim)
    std = (istd / (1. - p_drop))[:, None, None, None]
    std_w = torch.where(lay == 0, istd[:, None, None, None], std)                                                  # ... nor rescaled (:130)
    keep = 1. - torch.where(lay == 0, torch.zeros_like(p_drop)[:, None, None, None], p_drop[:, None, None, None])   # first matrix is never masked (:131)
    W = torch.randn(num_models, Lmax, HP, HP, device=dev) * std_w * torch.bernoulli(keep.expand(num_models, Lmax, HP, HP)) * live
    keep_b = (1. - p_drop)[:, None, None]
    live_b = live[:, :, 0, :] | ((lay[:, :, 0, :] < nl[:, None, None]) & (jout[:, :, 0, :] < out_dim[:, :, 0, :]))
    b = torch.randn(num_models, Lmax, HP, device=dev) * std[:, :, 0, :] * torch.bernoulli(keep_b.expand(num_models, Lmax, HP)) * live_b
    dims = torch.stack([nfu, hid, nl], 1).to(torch.int32)
    model_of = torch.arange(sample_batch_size, device=dev, dtype=torch.int32) // batch_size_per_gp_sample

    causes = None
    if sampling == 'uniform':
        causes = torch.rand(sample_batch_size, seq_len, HP, device=dev) * (kin[0, 0, :, 0][None, None, :] < nfu[model_of.long()][:, None, None])
    causes, y_raw = forward_networks(W, b, dims, nstd, model_of, seq_len, act, causes=causes)

    order_sign = None
    if is_binary_classification and order_y:
        order_sign = torch.tensor([1.0 if random.randint(0, 1) else -1.0 for _ in range(sample_batch_size)], device=dev)
    nfu_b = nfu[model_of.long()]
    scale = (nfu_b.float() / num_features) if normalize_by_used_features else None
    x, y = postprocess(causes, y_raw, nfu_b, num_features, is_binary_classification, order_sign, scale)
    x, y = x.transpose(0, 1), y.transpose(0, 1)
    return x, y, y


DataLoader = get_batch_to_dataloader(get_batch)
DataLoader.num_outputs = 1
DataLoader.prefetch = True
