import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive checks (minutes on 8 cores)")


def pytest_collection_modifyitems(config, items):
    # A test stuck in a device call must end as a failure, not as a hung box: pytest-timeout's "thread" method dumps the
    # stacks and exits the process even when the main thread sits inside a C call.  900 s is far above any test here
    # (the whole GPU suite takes ~30 s, the CPU suite ~15 s).
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900, method="thread"))
