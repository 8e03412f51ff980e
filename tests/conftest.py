import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from temporal_latticenet_amd import _lib
    _lib.lib()  # raises if the extension was not built: GPU tests never run on a fallback
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _seeded():
    """every test starts from the same RNG state: the modules' default initialisation (kaiming-uniform through
    torch's global generator) must not differ from run to run"""
    import numpy as np
    import torch
    torch.manual_seed(20240607)
    np.random.seed(20240607)
    yield
    from temporal_latticenet_amd import options
    options.reset()      # kernel-selection options a failed test left pushed on this host thread



def pytest_sessionfinish(session, exitstatus):
    """every full-size comparison with the oracle left (what, max_abs, max|logit|) in tests.helpers.PARITY: written to
    gpurun_out/parity_errors.json (copied to profiles/ and tabulated in DESIGN.md section 2)"""
    from tests.helpers import PARITY
    if not PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(PARITY, f, indent=1)
    except OSError:
        pass
