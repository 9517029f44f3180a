import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def record_measure(key, value):
    """Parity tests leave what they measured in gpurun_out/parity_measured.json (worst value per key over the run); the committed
    copy under profiles/ is what bench.py quotes in its `parity` object."""
    import json
    path = os.path.join(ROOT, "gpurun_out", "parity_measured.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        d = json.load(open(path)) if os.path.exists(path) else {}
        kind = "min" if key.endswith("_cos") else "max"
        old = d.get(key)
        d[key] = float(value) if old is None else (min(old, float(value)) if kind == "min" else max(old, float(value)))
        json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
