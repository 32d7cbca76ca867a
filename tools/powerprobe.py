#!/usr/bin/env python3
"""One kernel of the sweep in a tight loop for a fixed time (development tool), so that clocks and power can be sampled beside it
(rocm-smi --showclocks --showpower from the shell): which of them drives the chip into its power cap?
    python tools/powerprobe.py k2|k1|copy|sweep [seconds]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan  # noqa: E402

mode = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
dev = torch.device("cuda", 0)
n, H, V = 8, 60, 8
lib = _lib.load()
if mode == "copy":
    nbytes = 2 << 30
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream

    def one():
        _lib.check(lib.wx_stream_copy(b.data_ptr(), a.data_ptr(), ctypes.c_size_t(nbytes), st), "wx_stream_copy")
    unit = f"copy of {nbytes >> 30} GiB"
else:
    m = synthetic.euler3d_metric(n, H, V, 0, dev)
    m["christoffel"].view(3, 9, -1)[:, :3] = 0.0
    plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), m)
    q = synthetic.euler3d_state(n, H, V, 0, dev)
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=dev)
    sp = [send[e].data_ptr() for e in range(4)]
    out = torch.empty_like(q)
    if mode == "k2":
        plan.extrap_pack(q, sp)

        def one():
            plan.rhs(q, sp, out, _lib.WX_REGION_ALL)
    elif mode == "k1":
        def one():
            plan.extrap_pack(q, sp)
    else:
        def one():
            plan.extrap_pack(q, sp)
            plan.rhs(q, sp, out, _lib.WX_REGION_ALL)
    unit = {"k2": "fused kernel, one E7 panel", "k1": "extrapolation kernel, one E7 panel", "sweep": "K1 + K2, one E7 panel"}[mode]
for _ in range(20):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
count = 0
marks = []
while time.perf_counter() - t0 < secs:
    ta = time.perf_counter()
    for _ in range(50):
        one()
    torch.cuda.synchronize()
    marks.append((time.perf_counter() - ta) / 50)
    count += 50
third = max(1, len(marks) // 3)
print(f"{mode}: {unit}: {count} launches in {time.perf_counter() - t0:.1f} s; ms per launch: first third {sum(marks[:third]) / third * 1e3:.4f}, "
      f"last third {sum(marks[-third:]) / third * 1e3:.4f}", flush=True)
