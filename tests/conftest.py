import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights():
    from vnect_amd.weights import synthetic_weights
    return synthetic_weights()


@pytest.fixture(scope="session")
def oracle_net(weights):
    import oracle
    return oracle.Oracle(weights, keep=True)


@pytest.fixture(scope="session")
def h3(weights):
    """One finalized 3-scale fp32 handle (BASELINE scales) shared by the GPU test files."""
    from tests.gpu_common import BASELINE_SCALES, _handle
    h = _handle(BASELINE_SCALES, weights)
    yield h
    h.close()


@pytest.fixture(scope="session")
def ref3(weights, oracle_net):
    import oracle
    from tests.gpu_common import BASELINE_SCALES
    return oracle.OracleEstimator(scales=BASELINE_SCALES, net=oracle_net)
