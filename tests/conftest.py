import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """torch's bundled HIP runtime must come up BEFORE libcomposer_hip.so (linked against /opt/rocm) first touches the GPU: in
    the other order torch.cuda later reports "No HIP GPUs are available" (two HIP runtimes in one process).  Test files
    run in any order / selection, so the session initialises torch's side first when a GPU is present."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
