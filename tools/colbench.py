#!/usr/bin/env python3
"""The column form of the metric on one E7 panel with the true DCMIP 3-1 metric (development tool): K1 + K2, general
kernel against the column kernel, HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan  # noqa: E402

dev = torch.device("cuda", 0)
cases = [(8, 60, 8)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for n, H, V in cases:
    tile = CubedSphere3DTile(n, H, V, 0, 10000.0, 31)
    m = metric3d_torch(tile, dev)
    q = torch.from_numpy(initial_state(tile)).to(dev)
    q = q * (1.0 + 0.01 * (torch.rand_like(q) - 0.5))
    res = {}
    for label, col in (("general", False), ("column", True)):
        plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), m, column_metric=col)
        send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=dev)
        sp = [send[e].data_ptr() for e in range(4)]
        out = torch.zeros_like(q)
        t1, t2 = [], []
        for it in range(23):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            plan.extrap_pack(q, sp)
            e1.record()
            plan.rhs(q, sp, out, _lib.WX_REGION_ALL)
            e2.record()
            torch.cuda.synchronize()
            if it >= 3:
                t1.append(e0.elapsed_time(e1))
                t2.append(e1.elapsed_time(e2))
        res[label] = (sum(t1) / len(t1), sum(t2) / len(t2), out)
        del plan
    # the complex-step JVP kernel (unprepared form: dual state formed on load), same comparison
    v = torch.randn_like(q) * 1e-3 * q.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
    jres = {}
    for label, col in (("general", False), ("column", True)):
        plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), m, dtype=torch.complex128, dual=True, column_metric=col)
        send = torch.zeros((4, plan.edge_count), dtype=torch.complex128, device=dev)
        sp = [send[e].data_ptr() for e in range(4)]
        out = torch.zeros_like(q)
        tj = []
        for it in range(13):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            plan.jvp_extrap_pack(q, v, 1e-8, sp)
            e1.record()
            plan.jvp(q, v, 1e-8, sp, out, 1.0, _lib.WX_REGION_ALL)
            e2.record()
            torch.cuda.synchronize()
            if it >= 3:
                tj.append(e1.elapsed_time(e2))
        jres[label] = (sum(tj) / len(tj), out)
        del plan
    js = jres["general"][1].abs().amax(dim=(1, 2, 3, 4), keepdim=True)
    print(f"   JVP kernel: general {jres['general'][0]:.4f} ms, column {jres['column'][0]:.4f} ms ({jres['general'][0] / jres['column'][0]:.2f} x); "
          f"max |difference| / max per variable {float(((jres['general'][1] - jres['column'][1]).abs() / js).max()):.1e}", flush=True)
    a, b = res["general"][2], res["column"][2]
    scale = a.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
    print(f"n={n} {H}x{H}x{V}: K1 {res['general'][0]:.4f} ms; K2 general {res['general'][1]:.4f} ms, column {res['column'][1]:.4f} ms "
          f"({res['general'][1] / res['column'][1]:.2f} x); max |difference| / max |R| per variable {float(((a - b).abs() / scale).max()):.1e}",
          flush=True)
