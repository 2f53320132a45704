"""Priors on the hot path: fast_gp (GP, RBF), with prior / utils plumbing.  Unlike the reference
package (priors/__init__.py:1) nothing here depends on gpytorch / botorch / pyro."""
from transformerscandobayesianinference_amd.priors import prior, utils, fast_gp  # noqa: F401
