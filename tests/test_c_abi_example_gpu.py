"""examples/c_abi_rhs.c: a plain-C caller of libwxhip.so (gcc -std=c99, the HIP runtime for device memory; no Python, no
torch in the process) evaluates R(Q) of one tile of a golden fixture - plan, both launches, the nine-stamp timing row - and
compares it with the reference's R itself.  The boundary of this repo is a C ABI: this is a consumer of nothing else."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.util import golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["euler3d_c31p_n3_h4_v2", "euler3d_c21_n4_h3_v4"])
def test_plain_c_caller(built_lib, name, tmp_path):
    from tests.gpu_util import device_metric

    gcc, rocm = shutil.which("gcc"), os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no C toolchain / HIP headers on this box")
    g = golden(name)
    p = g.metric_panels()[0]
    d = tmp_path / "case"
    d.mkdir()
    (d / "meta.txt").write_text(f"{g.n} {g.H} {g.V} {g.case} {p}\n")
    for k in ("extrap_neg", "extrap_pos", "diff_solpt", "correction", "highfilter"):
        np.ascontiguousarray(g.ops[k], dtype=np.float64).tofile(d / f"ops_{k}.bin")
    for k, t in device_metric(g, p, "cpu").items():
        t.numpy().tofile(d / f"metric_{k}.bin")
    np.ascontiguousarray(g.q(p, False)).tofile(d / "q.bin")
    np.ascontiguousarray(g.r(p, False)).tofile(d / "r.bin")
    for e, h in enumerate(g.halo(p, False)):
        np.ascontiguousarray(h).tofile(d / f"halo_{e}.bin")
    exe = tmp_path / "c_abi_rhs"
    libdir = os.path.dirname(built_lib)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include",
                    os.path.join(ROOT, "examples", "c_abi_rhs.c"), f"-L{libdir}", "-lwxhip", f"-L{rocm}/lib", "-lamdhip64",
                    "-lm", "-o", str(exe)], check=True, capture_output=True, text=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([libdir, f"{rocm}/lib", os.environ.get("LD_LIBRARY_PATH", "")]))
    r = subprocess.run([str(exe), str(d)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PASS" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "timing row" in r.stdout and "expected refusal" in r.stdout
    assert "bit-identical to the single launch" in r.stdout   # the RCCL loopback leg (csrc/exchange.hip) from plain C
