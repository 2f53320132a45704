import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """PFN_TEST_TUNE="key=value,..." applies pfn_set_tuning keys (include/pfn_hip.h) to the whole session: the same parity suite then runs over a
    non-default kernel schedule (e.g. 4=3: the ping-pong barrier placement in the key-block pass too)."""
    spec = os.environ.get('PFN_TEST_TUNE', '')
    if spec:
        import torch  # noqa: F401  (libamdhip64 resident before the library loads)
        from transformerscandobayesianinference_amd import _hip
        for kv in spec.split(','):
            k, v = kv.split('=')
            _hip.check(_hip.lib().pfn_set_tuning(int(k), int(v)), 'pfn_set_tuning')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bounds
    bounds.dump()
    if bounds.measure_only():
        # PFN_BOUNDS_MEASURE_ONLY turns every `within` bound into a recording: such a session must never read as a pass -- it ends with a failure
        # status and says why, whatever the tests did
        sys.stderr.write('\n' + '!' * 100 + '\nPFN_BOUNDS_MEASURE_ONLY is set: the tolerance bounds of this session were RECORDED, NOT ASSERTED -- this run is not a test result '
                         '(exit status forced to 1)\n' + '!' * 100 + '\n')
        session.exitstatus = 1
