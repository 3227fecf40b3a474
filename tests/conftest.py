"""pytest config: `gpu` marker for tests that need a real MI355X (run by the driver with -m gpu)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def load_golden(name):
    """Load tests/golden/<name>.npz; keys 'sd.*' are returned as a separate state_dict."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd.")}
    rest = {k: z[k] for k in z.files if not k.startswith("sd.")}
    return sd, rest


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def _get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return _get
