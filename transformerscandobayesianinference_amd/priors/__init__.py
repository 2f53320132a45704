"""Priors on the hot path: fast_gp (GP, RBF), fast_gp_mix (Matern-5/2 ARD with Gamma hyper-priors), mlp (BNN prior), with prior / utils plumbing;
plus the two cheap priors of SURVEY.md 8(f) row 4 (ridge, binarized_regression).  Unlike the reference
package (priors/__init__.py:1) nothing here depends on gpytorch / botorch / pyro."""
from transformerscandobayesianinference_amd.priors import prior, utils, fast_gp, fast_gp_mix, mlp, ridge, binarized_regression  # noqa: F401
