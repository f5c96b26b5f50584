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
    config.addinivalue_line("markers", "plan_policy: the test runs under the library's own plan policy (plan_sa_min = 786,432, plan_min_per_string = 16, walk_order_min = boundary_order_min = 32,768) "
                                       "instead of the GPU suite's default of planning every batch of sort_min patterns or more")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _plan_every_large_batch(request):
    """The GPU parity tests were written around the planned path (suffix order, code words, translated plans of segment sets):
    they keep planning every batch of >= sort_min patterns.  Since round 4 the library skips the plan stage where it does not
    pay (fmx_count_batch_is_planned); tests marked `plan_policy` — the full-size configs, the policy's own test — run that way."""
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("plan_policy") is not None:
        yield
        return
    import index4j_amd as ia

    ia.lib.fmx_set_option(b"plan_min_per_string", 0)
    ia.lib.fmx_set_option(b"plan_sa_min", 0)
    ia.lib.fmx_set_option(b"walk_order_min", 1)  # ... and locate walks every batch by the first row of the ranges
    ia.lib.fmx_set_option(b"boundary_order_min", 1)  # ... extractUntilBoundary takes every batch by text position
    try:
        yield
    finally:
        ia.lib.fmx_set_option(b"plan_min_per_string", 16)
        ia.lib.fmx_set_option(b"plan_sa_min", 786432)
        ia.lib.fmx_set_option(b"walk_order_min", 32768)
        ia.lib.fmx_set_option(b"boundary_order_min", 32768)
