"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference algorithm for the PFN hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the timed CPU baseline.  The product package
(transformerscandobayesianinference_amd) never imports it and has no CPU fallback.

Parity pinning: the reference has no tests or golden vectors of its own, so
the restatement is pinned against outputs of the REFERENCE ITSELF, imported from /root/reference
in the build container by oracle/make_golden.py; the resulting fixtures live in tests/golden/ and
tests/test_oracle.py checks the oracle against them (model forward / loss / gradients / optimizer steps, bar
distribution, schedules and samplers, and the BNN prior priors.mlp -- rebuilt from every tensor the reference drew).
The GP priors are the exception: its numerics
live in gpytorch 1.5.0 / botorch 0.6.0 / pyro-ppl 1.7.0 (requirements.txt:2-3,14), which are not
installed and not vendored -- that part of the oracle restates the published algorithm
(y = chol(outputscale * k(x,x) + noise I) z) and its parity is "unpinned" against the reference.
"""
