import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# torch ships its own HIP runtime; libfmx.so links the system one.  Whichever is loaded first serves both, and
# torch only finds the GPU through its own — so tests that use torch for device memory / streams need it first.
try:
    import torch  # noqa: F401
except ImportError:  # the C-ABI tests do not need it
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
