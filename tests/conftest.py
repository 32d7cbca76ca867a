import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Build libwxhip.so if stale (hipcc cross-compiles here without a GPU)."""
    from wxfactory_amd.build import build

    return build()


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """A failing GPU test leaves its exception text ON DISK (write + fsync) before any teardown runs: round 4 lost the
    message of a test whose process was aborted from another thread while the main thread was still in its `except` /
    `finally` (profiles/r05_process_group_abort.md) - the next such failure names itself in gpurun_out/last_gpu_failure.txt,
    which gpurun copies back even when the process did not survive."""
    outcome = yield
    rep = outcome.get_result()
    if rep.failed and item.get_closest_marker("gpu") is not None:
        try:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            fd = os.open(os.path.join(out, "last_gpu_failure.txt"), os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
            try:
                os.write(fd, f"{item.nodeid} [{rep.when}]\n{rep.longreprtext}\n\n".encode("utf-8", "replace"))
                os.fsync(fd)
            finally:
                os.close(fd)
        except OSError:
            pass
