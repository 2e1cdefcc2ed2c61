import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # children of the pyramid-form tests: another tile size / split level for the large-rig pyramid plan (a debug hook of the library,
    # include/orb_debug.h: orbx_debug_pyramid_plan -- set before any handle exists)
    plan = os.environ.get("MORB_TEST_PYRAMID_PLAN")
    if plan:
        from multi_orb_slam_amd import _lib
        w, h, split = (int(x) for x in plan.split(","))
        _lib.lib().orbx_debug_pyramid_plan(w, h, split)


def pytest_collection_modifyitems(config, items):
    """No test may hang the suite: with pytest-timeout present (it is in this image) every test gets a limit -- generous next to the
    slowest one (~10 s; minutes for the oracle-heavy CPU tests) -- unless it set its own."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    import pytest
    for item in items:
        if item.get_closest_marker("timeout") is None:
            if item.get_closest_marker("gpu") is None:
                item.add_marker(pytest.mark.timeout(600))
            else:   # (a thread stuck inside a HIP call never returns to the interpreter: only the watchdog-thread method can end it)
                item.add_marker(pytest.mark.timeout(240, method="thread"))
