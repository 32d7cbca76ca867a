#!/usr/bin/env python3
"""Which kernels the callers spend their time in at the benchmark's resolution (development tool; run under
`rocprofv3 --kernel-trace --stats`): one Ros2 + FGMRES step, two EPI2 + KIOPS steps and a filtered SSP-RK3 step at
n = 8, 60 x 60 x V elements per panel.  Rows that are not this library's kernels are passes the host code adds."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from wxfactory_amd.filters import ExpFilter3D, NanFlag, make_filter  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.integrators import Epi, Ros2, StepLoop, Tvdrk3  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402
from wxfactory_amd.synthetic import dfr_ops  # noqa: E402

n, H, V = 8, 60, int(os.environ.get("V", "2"))
dev = torch.device("cuda", 0)
plans, Q, sg = {}, [], []
gen = torch.Generator(device=dev).manual_seed(3)
for p in range(6):
    tile = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
    m = metric3d_torch(tile, dev)
    sg.append(m["sqrtG"])
    plans[p] = Euler3DPlan(n, H, V, 31, p, dfr_ops(n), m)
    q = torch.from_numpy(initial_state(tile)).to(dev)
    Q.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5)))
Q0 = torch.stack(Q)
rhs = RhsEuler3D(plans)
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def clock(label, fn, reps=1):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / reps * 1e3:9.1f} ms", out if isinstance(out, (dict, str)) else "", flush=True)


if what in ("all", "ros2"):
    ros = Ros2(rhs, tol=1e-4, gmres_restart=20)

    def step():
        ros.step(Q0, 0.5)
        return {k: ros.solver_info[k] for k in ("iterations", "flag")}

    clock("Ros2 + FGMRES step (tol 1e-4, restart 20)", step)
if what in ("all", "epi"):
    epi = Epi(2, rhs, tol=1e-7)

    def step():
        epi.step(Q0, 0.5)
        return {k: epi.solver_info[k] for k in ("iterations", "substeps", "rejected")}

    clock("EPI2 + KIOPS step (tol 1e-7)", step, reps=2)
if what in ("all", "rk3"):
    F = make_filter(1e-3, 4, 0.5, np.polynomial.legendre.leggauss(n)[0])
    loop = StepLoop(Tvdrk3(rhs), ExpFilter3D(F, sg), NanFlag(dev), check_every=1000)
    state = {"q": Q0}

    def step():
        state["q"] = loop.step(state["q"], 1e-3)

    clock("filtered SSP-RK3 step", step, reps=5)
