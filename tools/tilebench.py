#!/usr/bin/env python3
"""24 tiles (k = 2) of the E7 sphere on one GPU, stacked state: per-tile launches vs one launch per phase for all tiles
(wx_euler3d_batch_*), without and with the multi-GPU region split (RCCL loopback) - development tool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.exchange import PanelExchange, RcclComm  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch  # noqa: E402
from wxfactory_amd.panels import CubeTopology  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
comm = RcclComm(0, 1, device=dev)   # the library's own communicator: no torch.distributed process group
n, H, V, k = 8, 60, 8, int(os.environ.get("K", "2"))
topo, Ht = CubeTopology(k), H // k
ops = synthetic.dfr_ops(n)
plans, qs = {}, []
for t in range(topo.ntiles):
    p, r, c = topo.locate(t)
    plans[t] = Euler3DPlan(n, Ht, V, 31, p, ops, metric3d_torch(CubedSphere3DTile(n, Ht, V, p, 10000.0, 31, row=r, col=c, k=k), dev),
                           on_panel_edge=topo.on_panel_edge(t))
    qs.append(synthetic.euler3d_state(n, Ht, V, t, dev))
Q = torch.stack(qs)
for loop in (False, True):
    ex = PanelExchange(5 * V * Ht * n * n, dev, rank=0, world_size=1, tiles_per_side=k, loopback=loop,
                       backend="rccl" if loop else "torch", comm=comm if loop else None)
    rhs = RhsEuler3D(plans, ex)
    outs = {}
    for label, maxpts in (("per-tile launches", 0), ("one launch per phase", 10**9)):
        rhs.batch_max_points = maxpts
        for _ in range(3):
            out = rhs(Q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = rhs(Q)
        torch.cuda.synchronize()
        outs[label] = out
        print(f"{topo.ntiles} tiles, split={loop!s:5} {label:22s}: {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per whole-sphere R(Q)", flush=True)
    print("   identical:", bool(torch.equal(*outs.values())))
torch.cuda.synchronize()
comm.close()
