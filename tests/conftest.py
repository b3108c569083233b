import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (str(ROOT), str(GOLDEN)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_npz(name):
    return np.load(GOLDEN / name)


def load_json(name):
    return json.loads((GOLDEN / name).read_text())


@pytest.fixture(scope="session")
def golden():
    return {"npz": load_npz, "json": load_json}
