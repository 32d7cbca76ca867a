"""The one test of the suite that creates a torch.distributed process group (tests/test_euler3d_gpu.py::
test_rccl_exchange_path_on_one_gpu: the N > 1 code path on one GPU, graphs captured under the group), run in an interpreter of
its own, as a rank of a several-GPU run is - and LAST (the file name): about once in ten runs of the whole suite that process
ended with SIGABRT and nothing on the captured stderr (profiles/r04_capture_crash.md, second part; ten runs of the test alone
under rocgdb: none).  In a child the rest of the suite is out of its reach, and its whole output - torch's own message
included - is in the assertion below if it happens again."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_exchange_path_on_one_gpu_in_its_own_process(built_lib):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-m", "pytest", "-q", "-x", "-s", "-m", "gpu",
                        "tests/test_euler3d_gpu.py", "-k", "test_rccl_exchange_path_on_one_gpu"], cwd=root,
                       env=dict(os.environ, WX_RCCL_TEST_CHILD="1", NCCL_DEBUG="WARN", TORCH_SHOW_CPP_STACKTRACES="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "1 passed" in r.stdout, f"rc={r.returncode}\n{r.stdout[-6000:]}\n{r.stderr[-6000:]}"
