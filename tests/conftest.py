import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Tests load the library and, in other files of the same process, torch: the library must share torch's bundled HIP
# runtime (two runtimes in one process find no GPU), which it only does when asked.
os.environ.setdefault("SCANSPACE_SHARE_TORCH_HIP", "1")

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """tests/golden/editing_tools.{npz,json}: outputs of the reference's own
    editingTools.py (see oracle/make_golden.py) keyed by case name."""
    with open(os.path.join(GOLDEN_DIR, "editing_tools.json")) as f:
        meta = json.load(f)["cases"]
    data = np.load(os.path.join(GOLDEN_DIR, "editing_tools.npz"))
    return meta, data


def golden_input(meta_case):
    from oracle.inputs import GENERATORS

    return GENERATORS[meta_case["gen"]](meta_case["seed"], meta_case["h"], meta_case["w"])
