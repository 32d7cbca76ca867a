#!/usr/bin/env python3
"""A/B of the DIRECT face stage (development tool): one panel of the reference's benchmark matrix per order, K2 on the
INTERIOR region with the interface buffer (K1 first) against the direct form (WXHIP_DIRECT=1: both face states
extrapolated from the nodal state in memory), HIP events, same process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan  # noqa: E402

dev = torch.device("cuda", 0)
cases = [(2, 30, 30), (3, 20, 20), (4, 15, 15), (5, 12, 12), (6, 10, 10)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for n, H, V in cases:
    m = synthetic.euler3d_metric(n, H, V, 0, dev)
    m["christoffel"].view(3, 9, -1)[:, :3] = 0.0
    plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), m)
    q = synthetic.euler3d_state(n, H, V, 0, dev)
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=dev)
    sp = [send[e].data_ptr() for e in range(4)]
    res = {}
    for mode in ("0", "1"):
        os.environ["WXHIP_DIRECT"] = mode
        out = torch.zeros_like(q)
        t1, t2 = [], []
        for it in range(33):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            plan.extrap_pack(q, sp)
            e1.record()
            plan.rhs(q, None, out, _lib.WX_REGION_INTERIOR)
            e2.record()
            torch.cuda.synchronize()
            if it >= 3:
                t1.append(e0.elapsed_time(e1))
                t2.append(e1.elapsed_time(e2))
        res[mode] = (sum(t1) / len(t1), sum(t2) / len(t2), out)
    os.environ["WXHIP_DIRECT"] = "0"
    a, b = res["0"][2], res["1"][2]
    scale = a.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
    diff = float(((a - b).abs() / scale).max())
    pts_int = V * (H - 2) ** 2 * n**3
    print(f"n={n} {H}x{H}x{V}: K1 {res['0'][0]*1e3:6.1f} us; K2 interior: buffer {res['0'][1]*1e3:6.1f} us, direct {res['1'][1]*1e3:6.1f} us "
          f"({plan.bytes_per_point*pts_int/res['1'][1]/1e6/80:.1f}% of 8 TB/s on compulsory bytes); max rel diff {diff:.1e}", flush=True)
