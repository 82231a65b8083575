import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


# fixtures with their own layout and their own tests (not single-env episode fixtures)
OTHER_FIXTURES = {"weather_resets", "harl_ny_n4", "harl_ny_n2_concat", "rbc_ny_m7", "tou_prices"}


def golden_names():
    """The single-env episode fixtures (tests/golden/gen_golden.py SPECS)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f[:-4] not in OTHER_FIXTURES)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
