"""Make the reference's top-level module names resolve to this package, so notebooks written
against the reference checkout (`from train import train; import priors, encoders, ...`) run unchanged:

    import transformerscandobayesianinference_amd.compat as compat; compat.install()
"""
import importlib
import sys

_NAMES = ['utils', 'encoders', 'decoders', 'positional_encodings', 'bar_distribution', 'transformer', 'train', 'priors', 'tabular']


def install():
    pkg = 'transformerscandobayesianinference_amd'
    for name in _NAMES:
        sys.modules[name] = importlib.import_module(f'{pkg}.{name}')
    for sub in ['prior', 'utils', 'fast_gp', 'fast_gp_mix', 'mlp', 'ridge', 'binarized_regression']:
        sys.modules[f'priors.{sub}'] = importlib.import_module(f'{pkg}.priors.{sub}')
