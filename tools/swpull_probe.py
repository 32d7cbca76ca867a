#!/usr/bin/env python3
"""Shallow water S7, direct form (development tool): the tile-edge lines pulled by the RHS launch itself from the neighbour
tiles' nodal values (wx_sw_batch_direct_pulls) against the ring pack in front of it (WXHIP_SW_PULL=0), alternating rounds in one
process; HIP events around 400 evaluations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.geometry import CubedSphereTile2D, metric2d_torch  # noqa: E402
from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan  # noqa: E402

dev = torch.device("cuda", 0)
n, H = 8, 60
ops = synthetic.dfr_ops(n)
plans, qs = {}, []
for p in range(6):
    plans[p] = SwPlan(n, H, p, ops, metric2d_torch(CubedSphereTile2D(n, H, p, phi0=0.7853981633974483), dev))
    qs.append(synthetic.sw_state(n, H, p, dev))
Q = torch.stack(qs)
forms = {}
for name, env in (("pulled", "1"), ("packed", "0")):
    os.environ["WXHIP_SW_PULL"] = env
    r = RhsShallowWater(plans)
    R = r(Q)
    forms[name] = (r, r._batches[torch.float64], R)
    assert forms[name][1].pulls == (env == "1")
assert torch.equal(forms["pulled"][2], forms["packed"][2])
out = torch.empty_like(Q)


def t(fn, reps=400):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for rnd in range(4):
    for name, (r, b, _) in forms.items():
        print(f"{name}: R(Q) {t(lambda: r(Q)):5.1f} us   RHS launch alone {t(lambda: b.rhs_direct(Q, out, _lib.WX_REGION_ALL)):5.1f} us"
              + (f"   ring pack alone {t(lambda: b.extrap_pack_ring(Q)):4.1f} us" if name == "packed" else ""))
