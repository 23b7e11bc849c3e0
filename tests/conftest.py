import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """Build the CPU checkers (oracle restatement; reference build when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    yield
