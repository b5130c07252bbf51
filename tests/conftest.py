import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# A tracer built without `use_graphs` replays HIP graphs from its second training trace on (the product default since round 5).  The tests' reference
# tracers mean the EAGER path (one launch per kernel, the thing the graph path is compared with); tests of the graph path say use_graphs=True /
# "static" themselves, and tests/test_abi_and_host.py::test_tracer_defaults_to_graphs checks the default without this variable.
os.environ.setdefault("PAG_GRAPHS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def table_from_seed(seed, shape, kind):
    """Same deterministic tables tests/golden/make_golden.py used."""
    rs = np.random.RandomState(seed)
    if kind == "uniform1e-4":
        return rs.uniform(-1e-4, 1e-4, size=shape).astype(np.float32)
    return rs.standard_normal(size=shape).astype(np.float32)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
