import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built libraries (they are git-ignored): build them once, as __graft_entry__.build() does
    pkg = os.path.join(ROOT, "tredparse_amd")
    if not all(os.path.exists(os.path.join(pkg, n)) for n in ("libtredgpu.so", "libtredbam.so")):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(pkg, "csrc")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def ctx():
    """One libtredgpu context on cuda:0 for the whole GPU session."""
    # PyTorch-ROCm ships its own HIP runtime: a process that uses both must load torch first,
    # whatever order the test files run in
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from tredparse_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def loci():
    from tredparse_amd import synth
    return synth.load_loci()
