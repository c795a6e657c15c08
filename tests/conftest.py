import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def cm():
    from cmdg_loader import cm as _cm
    return _cm


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def torch():
    import torch as t
    assert t.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    return t


def pytest_sessionfinish(session, exitstatus):
    """Observed maxima of the comparisons whose tolerance is looser than 1e-12 (helpers.observe)."""
    try:
        import json
        from helpers import OBSERVED
        if OBSERVED:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "observed_maxima.json"), "w") as f:
                json.dump(dict(sorted(OBSERVED.items())), f, indent=1)
    except Exception:
        pass
