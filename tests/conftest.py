import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout, when it is installed)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that does not come back (a kernel that spins) must fail, not hold the box until somebody's limit kills it:
    ten minutes per test, enforced from a watchdog thread (the main thread may sit in a blocking HIP call).  The limit is
    pytest-timeout's: without the plugin the marker would do nothing, so GPU tests refuse to run without it."""
    if not config.pluginmanager.hasplugin("timeout") and any(i.get_closest_marker("gpu") is not None for i in items):
        markexpr = getattr(config.option, "markexpr", "") or ""
        if "not gpu" not in markexpr:
            raise pytest.UsageError("the GPU tests need pytest-timeout (their watchdog): it is not installed")
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(int(os.environ.get("PCC_TEST_TIMEOUT", "600")), method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def work_lists_for_small_batches():
    """The library steps batches below 8192 envs without work lists (index order).  The parity tests run small batches and
    are there for the work-list machinery -- classes, team items, batched wave-path items, the 8- and 16-lane retire
    workgroups, restart items -- so they switch the lists on for every batch size; the tests of the small-batch path
    itself set BatchedNetworkEnv.DEFAULT_LIST_MIN_ENVS back to None."""
    import pcc_rl_amd
    B = pcc_rl_amd.BatchedNetworkEnv
    old = (B.DEFAULT_LIST_MIN_ENVS, B.DEFAULT_FUSED, B.DEFAULT_FUSED_ACQUIRE, B.DEFAULT_NOISE_SORTED)
    B.DEFAULT_LIST_MIN_ENVS = 0
    # the step as two launches (the library's default) or as one (PCC_TEST_FUSED=1), and the fused step's acquire mode:
    # tests/test_variants.py runs slices of the parity suite through each
    if os.environ.get("PCC_TEST_FUSED") is not None:
        B.DEFAULT_FUSED = int(os.environ["PCC_TEST_FUSED"])
    if os.environ.get("PCC_TEST_FUSED_ACQUIRE") is not None:
        B.DEFAULT_FUSED_ACQUIRE = int(os.environ["PCC_TEST_FUSED_ACQUIRE"])
    # latency noise: intervals by sorting (default), by the event loop (0), or the two crossed (2: the small instance + the event loop)
    if os.environ.get("PCC_TEST_NOISE_SORTED") is not None:
        B.DEFAULT_NOISE_SORTED = int(os.environ["PCC_TEST_NOISE_SORTED"])
    yield
    B.DEFAULT_LIST_MIN_ENVS, B.DEFAULT_FUSED, B.DEFAULT_FUSED_ACQUIRE, B.DEFAULT_NOISE_SORTED = old
