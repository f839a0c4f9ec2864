import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Built artefacts are git-ignored.  Every session runs the (idempotent, make-driven) build exactly as
    # __graft_entry__.build() does, so neither a fresh checkout nor an edited source ever runs a stale library or
    # host program.  Building is not a fallback -- without hipcc the tests that need the library fail loudly.
    if os.path.exists("/opt/rocm/bin/hipcc") and os.environ.get("NUSLAM_SKIP_BUILD") != "1":
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def hip():
    """The product library through its C ABI. GPU tests fail (not skip) when it is missing or no device is visible."""
    import nuslam_hip
    nuslam_hip.lib()
    assert nuslam_hip.device_count() > 0, "no HIP device visible: GPU tests must run on the GPU box"
    return nuslam_hip
