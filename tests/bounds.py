"""Asserted tolerances WITH their measured values on record: `within(label, value, bound)` asserts `value < bound` and remembers the
largest value seen per (test, label).  With PFN_RECORD_BOUNDS=<path> the session writes them as JSON (tests/conftest.py) -- the
committed copy is profiles/r03_parity_measured.json, and the bf16 bounds in the tests are set to <= 2x those measurements."""
import json
import os

_measured = {}


def within(label, value, bound):
    value = float(value)
    test = os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0].split('::', 1)[-1]
    rec = _measured.setdefault(f'{test} :: {label}', dict(bound=bound, max=value, n=0))
    rec['max'] = max(rec['max'], value)
    rec['bound'] = bound
    rec['n'] += 1
    if not os.environ.get('PFN_BOUNDS_MEASURE_ONLY'):      # (measuring a re-based bound: record every value of the test, assert nothing)
        assert value < bound, f'{label}: {value:.3e} is not below the bound {bound:.1e}'
    return value


def measure_only():
    return bool(os.environ.get('PFN_BOUNDS_MEASURE_ONLY'))


def dump():
    path = os.environ.get('PFN_RECORD_BOUNDS')
    if path and _measured:
        os.makedirs(os.path.dirname(os.path.abspath(path)) or '.', exist_ok=True)
        json.dump({k: dict(v, headroom=(v['bound'] / v['max'] if v['max'] > 0 else None)) for k, v in sorted(_measured.items())}, open(path, 'w'), indent=1)
