import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """Build artefacts are git-ignored: on a fresh checkout compile them once (hipcc cross-compiles without a GPU)."""
    csrc = ROOT / "spherical_bundle_adjuster_amd" / "csrc"
    if not (ROOT / "spherical_bundle_adjuster_amd" / "libsba_hip.so").exists() or not (csrc / "build" / "sba_main").exists():
        subprocess.run(["make", "-C", str(csrc), "-j8"], check=True, capture_output=True)
    if not (ROOT / "oracle" / "libsba_oracle.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built on demand with g++."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py
