import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (this container only)")


def pytest_collection_modifyitems(config, items):
    """Plain ``pytest`` on a box without a GPU: every ``gpu``-marked test is SKIPPED (not an error), whether or not it goes
    through the ``hip_ops`` fixture.  On a GPU box nothing is skipped, and the HIP library itself is mandatory there:
    ``HipOps()`` raises if ``libbasq_hip.so`` is missing (no fallback), so a broken build cannot pass silently."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_ops():
    import torch

    if not torch.cuda.is_available():
        # plain `pytest` on a CPU box: the GPU tests are skipped, not errors.  On a GPU box the HIP library itself is
        # mandatory: HipOps() raises if libbasq_hip.so is missing (no fallback), so a broken build cannot pass silently.
        pytest.skip("no HIP device visible (run with -m gpu on an MI355X)")
    from basq_amd._ops import HipOps

    return HipOps(torch.device("cuda", 0))
