#!/usr/bin/env python3
"""Development probe: the Ros2+FGMRES and KIOPS callers on the reference fixture, printing the numbers the
GPU tests bound (iteration counts, errors against the reference's results)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.gpu_util import device_metric  # noqa: E402
from tests.util import Golden  # noqa: E402
from wxfactory_amd.integrators import Epi, Ros2  # noqa: E402
from wxfactory_amd.matvec import matvec_fun  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402
from wxfactory_amd.solvers import kiops  # noqa: E402

DEV = "cuda:0"
g = Golden("callers_euler3d_n3_h3_v2")
plans = {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV)) for p in range(6)}
rhs = RhsEuler3D(plans)
stack = lambda key: torch.from_numpy(np.stack([g[f"p{p}/{key}"] for p in range(6)])).to(DEV)  # noqa: E731
ax = (0, 2, 3, 4, 5)
for ortho in ("igs", "cgs"):
    ros = Ros2(rhs, tol=1e-9, gmres_restart=30, ortho=ortho)
    Qn = ros.step(stack("Q"), float(g["meta/dt_jvp"]))
    ref, q0 = stack("ros2").cpu().numpy(), stack("Q").cpu().numpy()
    upd = np.abs(ref - q0).max(axis=ax)
    err = np.abs(Qn.cpu().numpy() - ref).max(axis=ax)
    print(ortho, "ros2:", ros.solver_info, "err/upd", err / upd, "ref iterations", [k for k in g.z.files if "ros2" in k])
Q, R = stack("Q"), stack("R")
dt = float(g["meta/dt_jvp"])
vec = torch.zeros((2, R.numel()), dtype=torch.float64, device=DEV)
vec[1] = R.flatten()
phiv, stats = kiops([1], lambda v: matvec_fun(v, dt, Q, R, rhs, "complex"), vec, tol=1e-7, m_init=1, mmin=16, mmax=64)
print("kiops stats", stats, "ref", g["p0/kiops_stats"])
ref = stack("kiops_phiv").cpu().numpy()
print("kiops err", np.abs(phiv.cpu().numpy().reshape(ref.shape) - ref).max(axis=ax) / np.abs(ref).max(axis=ax))
