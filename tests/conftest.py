import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (C oracle) and the product library once per session."""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    from ngp_pl_amd import build as b
    b.build()
    yield
