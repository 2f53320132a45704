"""MI355X-native PFN training hot path (drop-in for automl/TransformersCanDoBayesianInference).

Module names mirror the reference checkout (train, transformer, bar_distribution, encoders,
decoders, positional_encodings, utils, priors.*), so reference notebooks keep working after
`transformerscandobayesianinference_amd.compat.install()`.
"""
__version__ = '0.1.0'
