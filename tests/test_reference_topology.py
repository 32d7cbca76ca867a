"""The generator of the exchange fixtures is itself pinned: the reference's own ProcessTopologyTest - 13 cases of scalar and
vector exchanges, 1e-15 (tests/unit/common/test_process_topology.py:78-545 of the reference) - passes on six ranks of the
threaded mpi4py stand-in under which oracle/refharness/gen_golden.py ran the reference to produce tests/golden/*.npz
(the halos `q_itf_{s,n,w,e}` that tests/test_exchange_gloo.py and the GPU tests compare the product's exchange with).
Build container only: the reference tree does not travel to the GPU box (there this test is skipped; nothing else reads it)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WX_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "wx_factory")), reason="the reference tree exists in the build container only")
def test_reference_process_topology_cases_pass_under_the_stand_in():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "refharness", "check_topology.py")], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "ALL 13 PASSED" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    assert r.stdout.count("ok on 6 ranks") == 13
