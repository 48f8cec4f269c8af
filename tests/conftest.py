import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# A bounded in-kernel hand-off wait that gives up is an ERROR in every test: `Zonos.generate` must not repeat the generation behind
# the test's back (zonos_amd/model.py: _generate_on).  The tests of the retry itself remove the variable (monkeypatch.delenv).
os.environ["ZONOS_HIP_NO_TIMEOUT_RETRY"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
