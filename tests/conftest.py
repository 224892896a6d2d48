import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nlp_golden():
    with open(os.path.join(GOLDEN, "nlp_eval.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def harness_golden():
    with open(os.path.join(GOLDEN, "harness.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _native_library_built():
    """libobca_mpc.so is a build product (git-ignored): make sure it exists before any test loads it.  A no-op when
    the in-tree library is newer than its sources; never a fallback -- without hipcc this raises."""
    import __graft_entry__ as ge
    ge.build()
