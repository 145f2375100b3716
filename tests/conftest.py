import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:          # tests/chains: the kernel-by-kernel launch chains the C entry points are checked against
    sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """The shared library is git-ignored (built in-tree, it travels with the working copy): a fresh checkout builds it once
    here, exactly as `__graft_entry__.build()` does (hipcc cross-compiles gfx950 without a GPU)."""
    lib = ROOT / "stamp_amd" / "lib" / "libamdstamp.so"
    if not lib.is_file():
        import subprocess
        subprocess.run(["make", "-j", str(os.cpu_count() or 4)], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from stamp_amd import _lib

    _lib.lib()  # the HIP extension must be present on a GPU box: fail loudly, never fall back
    return torch.device("cuda:0")
