import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The plain-C restatement (oracle/liboracle.so) — the checker, never the product."""
    from oracle_py import Oracle, ORACLE_SO
    if not os.path.exists(ORACLE_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return Oracle()


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
