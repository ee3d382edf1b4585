import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pins():
    return json.loads((GOLDEN / "reference_pins.json").read_text())


@pytest.fixture(scope="session")
def data_cv():
    """The reference's bundled CV set (processed_dataset/data_cv.npy), float32 100 x 18."""
    return np.load(GOLDEN / "data_cv.npy")


@pytest.fixture(scope="session")
def data_labels():
    return np.load(GOLDEN / "data_labels.npy")


@pytest.fixture(scope="session")
def data8(data_cv, pins):
    return data_cv[:, pins["useful_ft"]]
