import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'guard_fallback_expected: the test provokes the f16x3 range guard on purpose')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """`gpu` tests skip (instead of erroring in their fixtures) on a host without a GPU."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a real MI355X (no GPU visible)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _no_silent_matrix_mode_fallback(request):
    """A fallback is a failure (round-4 verdict, weak #2): a GPU test in which the f16x3 range guard moved an engine to
    bf16x6 (`engine.GUARD_LOG` grew), or which left the process-wide matrix mode changed, FAILS unless it is marked
    `guard_fallback_expected` -- otherwise a test of an f16x3 kernel can pass on the exact-split mode's result, which is how
    the 1 x 97 x 163 stem bug stayed hidden in round 4."""
    if 'gpu' not in request.keywords:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from eosvos_amd import engine
    n, mode = len(engine.GUARD_LOG), engine.get_matrix_mode()
    yield
    grown, now = list(engine.GUARD_LOG[n:]), engine.get_matrix_mode()
    if request.node.get_closest_marker('guard_fallback_expected') is not None:
        del engine.GUARD_LOG[n:]
        engine.set_matrix_mode(mode)
        return
    assert not grown, f'the f16x3 range guard fell back during this test: {grown}'
    assert now == mode, f'the test left the process-wide matrix mode at {now} (was {mode})'
