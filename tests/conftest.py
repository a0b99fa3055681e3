import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")
GOLDEN = ROOT / "tests" / "golden"

# The serving pool test starts GPU worker processes.  Their fork server is started here, while this process has not
# touched a GPU yet, so that the workers descend from a clean process rather than from one with HIP state.
if sys.platform == "linux":
    from multiprocessing import forkserver as _forkserver

    try:
        _forkserver.set_forkserver_preload([])
        _forkserver.ensure_running()
    except Exception:  # no fork server here: the pool starts one on demand instead
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
