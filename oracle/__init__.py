"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference algorithm for the PFN hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the timed CPU baseline.  The product package
(transformerscandobayesianinference_amd) never imports it and has no CPU fallback.

Parity pinning: the reference has no tests or golden vectors of its own, so
the restatement is pinned against outputs of the REFERENCE ITSELF, imported from /root/reference
in the build container by oracle/make_golden.py; the resulting fixtures live in tests/golden/ and
tests/test_oracle.py checks the oracle against them (model forward / loss / gradients / optimizer steps, bar
distribution, schedules and samplers, and the BNN prior priors.mlp -- rebuilt from every tensor the reference drew).
The GP priors: gpytorch 1.5.0 / botorch 0.6.0 / pyro-ppl 1.7.0 (requirements.txt:2-3,14) are not installed and not
vendored, so `priors/fast_gp.py` itself cannot run here -- but the reference holds a second, sklearn statement of the same
RBF GP (priors/gp.py:14-17 kernel, :41-62 sequential exact-GP evaluation) which does: `gp_gram` and `gp_evaluate` are
pinned to it (tests/golden/gp_sklearn.pt, agreement 1e-14 in f64), and the Matern-5/2 ARD Gram of priors.fast_gp_mix to
sklearn's `Matern(nu=2.5)`.  Unpinned remainder: gpytorch-only conventions (the (noise, outputscale, lengthscale) tuple,
the noise floor 1e-9, psd_safe_cholesky's jitter ladder) and the Gamma hyper-prior draws of fast_gp_mix (checked by moments).
"""
