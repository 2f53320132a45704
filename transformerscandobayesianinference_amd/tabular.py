"""Tabular entry point: build (and optionally train) the PFN that `TabularEvalSimple.ipynb` evaluates, and the
checkpoint convention the notebooks use to hand a trained model around.

Mirror of the model-building part of the reference `tabular.py`:
    get_uniform_single_eval_pos_sampler   :39-44
    get_*_prior_hyperparameters           :47-106
    get_model                             :109-155
plus `save_checkpoint` / `load_checkpoint` for the `(state_dict, optimizer_state)` tuples of
`BayesianModels_And_Custom_Pyro_Modules.ipynb` cells 14 / 16 and `TabularEvalSimple.ipynb` cell 12.
The evaluation half of the reference file (OpenML datasets, XGBoost / CatBoost / sklearn / pyro baselines,
:160-677) needs network access and libraries this stack does not carry; it is out of scope.

`get_model` trains through this package's `train()` -- HIP encoder stack, fused optimizer, HIP prior samplers --
so a PFN trained here loads into the reference's `TransformerModel` and vice versa: the state-dict keys are the
reference's.
"""
import torch

from transformerscandobayesianinference_amd import encoders, priors
from transformerscandobayesianinference_amd.priors.utils import gamma_sampler_f, trunc_norm_sampler_f
from transformerscandobayesianinference_amd.train import Losses, train
from transformerscandobayesianinference_amd.utils import get_uniform_single_eval_pos_sampler  # noqa: F401  (reference tabular.py:39-44 defines it here)


def _only(named_sampler: dict):
    """The notebook configs wrap every sampler as {'description': callable}; the callable is what the prior wants."""
    return next(iter(named_sampler.values()))


def get_mlp_prior_hyperparameters(config):
    """The 17-tuple `priors.mlp.get_batch` unpacks (priors/mlp.py:103-112), from a notebook config dict
    (`TabularEvalSimple.ipynb` cell 7).  The causal-graph fields are only filled when `prior_is_causal`."""
    causal = config['prior_is_causal']
    when_causal = lambda value: value if causal else None
    return (
        _only(config['prior_nlayers_sampler']),
        _only(config['prior_emsize_sampler']),
        config['prior_activations'],
        gamma_sampler_f(config['prior_sigma_gamma_k'], config['prior_sigma_gamma_theta']),          # init / weight std
        gamma_sampler_f(config['prior_noise_std_gamma_k'], config['prior_noise_std_gamma_theta']),  # per-layer noise std
        _only(config['prior_dropout_sampler']),
        True,                                                                                        # binary labels
        _only(config['prior_num_features_used_sampler']),
        _only(config['prior_causes_sampler']) if causal else None,
        causal,
        when_causal(config.get('prior_pre_sample_causes')),
        when_causal(config.get('prior_pre_sample_weights')),
        when_causal(config.get('prior_y_is_effect')),
        config['prior_order_y'],
        config['prior_normalize_by_used_features'],
        _only(config['prior_categorical_feats']) if causal else None,
        0.0,                                                                                         # nan_prob
    )


def get_gp_mix_prior_hyperparameters(config):
    """Hyper-prior dict of `priors.fast_gp_mix`.  Reference quirk kept on purpose (tabular.py:72-79): the two keys
    'categorical_data' and 'y_minmax_norm' are fed from `prior_y_minmax_norm` and `prior_lengthscale_concentration`
    respectively, so `y_minmax_norm` ends up truthy whenever the lengthscale concentration is non-zero."""
    return {
        'lengthscale_concentration': config['prior_lengthscale_concentration'],
        'nu': config['prior_nu'],
        'outputscale_concentration': config['prior_outputscale_concentration'],
        'categorical_data': config['prior_y_minmax_norm'],
        'y_minmax_norm': config['prior_lengthscale_concentration'],
        'noise_concentration': config['prior_noise_concentration'],
        'noise_rate': config['prior_noise_rate'],
    }


def get_gp_prior_hyperparameters(config):
    """7-tuple of the reference's 'gp' prior type (tabular.py:82-91): constant outputscale / lengthscale wrapped as samplers."""
    outputscale, lengthscale = config['prior_outputscale'], config['prior_lengthscale']
    return (config['prior_noise'], lambda: outputscale, lambda: lengthscale, True,
            _only(config['prior_num_features_used_sampler']), config['prior_normalize_by_used_features'], config['prior_order_y'])


def get_meta_gp_prior_hyperparameters(config):
    """7-tuple of the 'custom_gp_mix' prior type (tabular.py:94-105): truncated-normal samplers around the configured means."""
    def around(mean_key, std_factor_key):
        return trunc_norm_sampler_f(config[mean_key], config[mean_key] * config[std_factor_key])
    return (config['prior_noise'], around('prior_outputscale_mean', 'prior_outputscale_std_f'),
            around('prior_lengthscale_mean', 'prior_lengthscale_std_f'), True,
            _only(config['prior_num_features_used_sampler']), config['prior_normalize_by_used_features'], config['prior_order_y'])


def _prior_for(config):
    kind = config['prior_type']
    if kind == 'mlp':
        return priors.mlp.DataLoader, get_mlp_prior_hyperparameters(config), {'batch_size_per_gp_sample': 8}
    if kind == 'gp_mix':
        return priors.fast_gp_mix.DataLoader, get_gp_mix_prior_hyperparameters(config), {}
    if kind in ('gp', 'custom_gp_mix'):
        # the reference hands these 7-tuples (with callables for outputscale / lengthscale) to priors.fast_gp.get_batch,
        # which reads entries 0..2 as plain GP hyper-parameters -- that path cannot run there either (the notebook
        # raises "Not Implemented" for it, TabularEvalSimple.ipynb cell 12).  Draw the constants once instead.
        hps = get_gp_prior_hyperparameters(config) if kind == 'gp' else get_meta_gp_prior_hyperparameters(config)
        return priors.fast_gp.DataLoader, (hps[0], float(hps[1]()), float(hps[2]())), {}
    raise ValueError(f"unknown prior_type {kind!r} (mlp, gp, custom_gp_mix, gp_mix)")


def get_model(config, device, eval_positions, should_train=True, verbose=False, **train_kwargs):
    """Reference tabular.py:109-155: a binary-classification PFN (`Losses.bce`, one output) for the prior named by
    `config['prior_type']`; `should_train=False` only builds it (0 epochs) so a checkpoint can be loaded into it.
    Returns train()'s 3-tuple; the notebooks take element [2], the model.  `train_kwargs` reach `train()`
    (e.g. steps_per_epoch, precision) -- the reference fixes steps_per_epoch at 100."""
    prior_class, hyperparameters, extra = _prior_for(config)
    epochs = config['epochs'] if should_train else 0
    kwargs = dict(steps_per_epoch=100)
    kwargs.update(train_kwargs)
    return train(prior_class, Losses.bce, encoders.Linear,
                 emsize=config['emsize'], nhead=config['nhead'], nhid=config['emsize'] * config['nhid_factor'],
                 nlayers=config['nlayers'], dropout=config['dropout'], y_encoder_generator=encoders.Linear,
                 pos_encoder_generator=None, batch_size=config['batch_size'], bptt=config['bptt'], lr=config['lr'],
                 epochs=epochs, warmup_epochs=epochs // 4, gpu_device=device, verbose=verbose,
                 single_eval_pos_gen=get_uniform_single_eval_pos_sampler(max(eval_positions) + 1),
                 extra_prior_kwargs_dict={'num_features': config['num_features'], 'fuse_x_y': False,
                                          'hyperparameters': hyperparameters, 'device': device, **extra},
                 **kwargs)


def save_checkpoint(model, path, optimizer_state=None):
    """`torch.save((state_dict, optimizer_state), path)` -- the tuple the notebooks write and read."""
    state = {k: v.detach().to('cpu') for k, v in model.state_dict().items()}
    torch.save((state, optimizer_state), path)


def load_checkpoint(model, path, strict=True):
    """Inverse of `save_checkpoint`; also reads checkpoints written by the reference.  Returns the optimizer state
    stored next to the weights (None in every checkpoint the reference ships)."""
    model_state, optimizer_state = torch.load(path, map_location='cpu')
    model.load_state_dict(model_state, strict=strict)
    return optimizer_state
