#!/usr/bin/env python3
"""Diagnostic: phase timeline of the fused RHS kernel from in-kernel wall-clock stamps
(needs a library built with -DWX_K2_DIAG=1; shares, not absolute speed)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WXHIP_LIB", os.path.join(ROOT, "wxfactory_amd", "lib", "v_stamps.so"))
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan  # noqa: E402

dev = torch.device("cuda", 0)
n, H, V = 8, 60, 8
DUAL = "--dual" in sys.argv  # phase timeline of the dual-number instantiation (JVP kernels)
dt = torch.complex128 if DUAL else torch.float64
COLUMN = "--column" in sys.argv  # the column form of the plan on a level-invariant metric
m = synthetic.euler3d_metric(n, H, V, 0, dev)
m["christoffel"].view(3, 9, -1)[:, :3] = 0.0   # non-rotating planet, as the benchmark
if COLUMN:
    synthetic.make_level_invariant(m, n, H, V)
plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), m, dtype=dt, dual=DUAL, column_metric=COLUMN)
q = synthetic.euler3d_state(n, H, V, 0, dev)
if DUAL:
    q = q + 1e-8j * q
send = torch.zeros((4, plan.edge_count), dtype=dt, device=dev)
sp = [send[e].data_ptr() for e in range(4)]
out = torch.empty_like(q)
nb = V * H * H
st = torch.zeros((nb, 8), dtype=torch.int64, device=dev)
lib = _lib.load()
fn = lib._handle if False else ctypes.CDLL(_lib.LIB_PATH).wx_euler3d_debug_set_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
fn(plan._h, st.data_ptr())
for _ in range(3):
    plan.extrap_pack(q, sp)
    plan.rhs(q, sp, out, 0)
torch.cuda.synchronize()
s = st.cpu().numpy().astype("float64")
names = ["const+early loads+face", "point loads/pointwise/forcing", "dir 0", "dir 1", "dir 2", "assemble+store"]
d = (s[:, 1:7] - s[:, 0:6]) * 10.0  # 100 MHz -> ns
tot = (s[:, 6] - s[:, 0]) * 10.0
print(f"workgroups {nb}; kernel span {(s[:,6].max()-s[:,0].min())*10/1e6:.3f} ms; mean WG lifetime {tot.mean()/1e3:.2f} us")
for i, nm in enumerate(names):
    print(f"  {nm:32s} {d[:, i].mean()/1e3:7.2f} us  ({100*d[:, i].mean()/tot.mean():5.1f} %)")
