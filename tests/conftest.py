import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long variant of a lock-step run whose shorter form is in the default suite (MEMGYM_SLOW=1 runs them)")


def pytest_collection_modifyitems(config, items):
    """The default GPU suite stays under the driver's step limit: every family keeps long lock-step runs in it, the longer
    duplicates are marked `slow` and run with MEMGYM_SLOW=1."""
    import pytest

    if os.environ.get("MEMGYM_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow variant: set MEMGYM_SLOW=1")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)
