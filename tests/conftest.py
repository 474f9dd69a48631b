import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ASSETS = os.path.join(ROOT, "assets")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the host library and the oracle once per session (CPU only; the HIP library is built by __graft_entry__.build())."""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "vk_gltf_renderer_amd", "csrc"), "host"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return True


@pytest.fixture(scope="session")
def assets():
    return ASSETS
