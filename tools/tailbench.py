#!/usr/bin/env python3
"""Development benchmark: the whole-sphere E7 evaluation of ONE rank that owns all six panels (bench.py at N = 1), with
the twelve launches of an evaluation (six K1, six K2) issued in different stream arrangements.  On one stream every
launch waits for the last workgroups of its predecessor (the tail: 28 800 workgroups over 512 slots = 56.25 rounds, and
a barrier packet between dependent launches); K2 launches of different panels are independent of each other once every
K1 has run, so striping them over two streams lets one launch's tail fill with the next launch's head.

    python tools/tailbench.py [--reps 30]

Prints ms per evaluation for: one stream; K2 striped over 2 / 3 streams; K1 and K2 striped."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.exchange import PanelExchange  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--H", type=int, default=60)
ap.add_argument("--V", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, H, V = a.n, a.H, a.V
ops = synthetic.dfr_ops(n)
plans, qs, outs = [], [], []
for p in range(6):
    m = synthetic.euler3d_metric(n, H, V, p, dev)
    m["christoffel"].view(3, 9, -1)[:, :3] = 0.0   # non-rotating planet, as the benchmark's DCMIP 3-1
    plans.append(Euler3DPlan(n, H, V, 31, p, ops, m))
    qs.append(synthetic.euler3d_state(n, H, V, p, dev))
    outs.append(torch.empty_like(qs[-1]))
ex = PanelExchange(plans[0].edge_count, dev, rank=0, world_size=1)
send = [ex.send_ptrs(p) for p in range(6)]
halo = [ex.halo_ptrs(p) for p in range(6)]
main = torch.cuda.current_stream(dev)
sides = [torch.cuda.Stream(dev) for _ in range(2)]


def evaluation(k1_streams, k2_streams):
    """k*_streams: how many streams the launches of that phase are striped over (1 = the current stream only)."""
    def striped(nstreams, fn):
        if nstreams == 1:
            for p in range(6):
                fn(p)
            return
        used = sides[:nstreams - 1]
        for s in used:
            s.wait_stream(main)
        for p in range(6):
            k = p % nstreams
            if k == 0:
                fn(p)
            else:
                with torch.cuda.stream(used[k - 1]):
                    fn(p)
        for s in used:
            main.wait_stream(s)

    striped(k1_streams, lambda p: plans[p].extrap_pack(qs[p], send[p]))
    striped(k2_streams, lambda p: plans[p].rhs(qs[p], halo[p], outs[p], _lib.WX_REGION_ALL))


def split_evaluation(pipelined):
    """INTERIOR launches on a second stream, BOUNDARY launches (ring elements: 6.5 % of a panel) on the main one, beside
    them; pipelined: the INTERIOR launch of a panel follows that panel's K1 at once (it needs no other panel's faces)."""
    side = sides[0]
    if not pipelined:
        for p in range(6):
            plans[p].extrap_pack(qs[p], send[p])
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for p in range(6):
                plans[p].rhs(qs[p], None, outs[p], _lib.WX_REGION_INTERIOR)
    else:
        for p in range(6):
            plans[p].extrap_pack(qs[p], send[p])
            side.wait_stream(main)
            with torch.cuda.stream(side):
                plans[p].rhs(qs[p], None, outs[p], _lib.WX_REGION_INTERIOR)
    for p in range(6):
        plans[p].rhs(qs[p], halo[p], outs[p], _lib.WX_REGION_BOUNDARY)
    main.wait_stream(side)


def clock(k1s, k2s):
    if k1s == "split":
        fn = lambda: split_evaluation(k2s)  # noqa: E731
    else:
        fn = lambda: evaluation(k1s, k2s)  # noqa: E731
    return clock_fn(fn)


def clock_fn(evaluation):
    for _ in range(3):
        evaluation()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        evaluation()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


evaluation(1, 1)
torch.cuda.synchronize()
ref = [o.clone() for o in outs]

# the product's host classes on the same plans: RhsEuler3D over the aliasing exchange (Python launches, one stream, ALL) and
# over the library's own exchange in loopback mode (one call of wx_euler3d_rhs_overlapped: INTERIOR on the second stream,
# grouped sends / receives and BOUNDARY on the first)
from wxfactory_amd.exchange import RcclComm  # noqa: E402
from wxfactory_amd.rhs_euler3d import RhsEuler3D  # noqa: E402

state = torch.stack(qs)
pd = {p: plans[p] for p in range(6)}
rhs_alias = RhsEuler3D(pd, PanelExchange(plans[0].edge_count, dev, rank=0, world_size=1))
comm = RcclComm(0, 1, device=dev)
rhs_loop = RhsEuler3D(pd, PanelExchange(plans[0].edge_count, dev, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm))
for name, r in (("RhsEuler3D, aliasing exchange", rhs_alias), ("RhsEuler3D, library exchange in loopback (one C call)", rhs_loop)):
    got = r(state)
    torch.cuda.synchronize()
    same = all(torch.equal(got[p], ref[p]) for p in range(6))
    for rnd in range(2):
        ms = clock_fn(lambda: r(state))
        print(f"{name}: {ms:7.4f} ms per evaluation; bit-identical to one stream: {same}", flush=True)
pts = 6 * V * H * H * n**3
bpp = plans[0].bytes_per_point
for rnd in range(2):
    for k1s, k2s in ((1, 1), (1, 2), (2, 2), ("split", False), ("split", True), (1, 1)):
        ms = clock(k1s, k2s)
        same = all(torch.equal(o, r) for o, r in zip(outs, ref))
        label = (f"K1 on {k1s} stream(s), K2 on {k2s}" if k1s != "split" else
                 "INTERIOR on a second stream beside K1 (pipelined per panel) and BOUNDARY" if k2s else
                 "INTERIOR on a second stream beside BOUNDARY")
        print(f"{label}: {ms:7.4f} ms per evaluation = {bpp * pts / ms / 1e6 / 80:5.2f} % of 8 TB/s "
              f"on {bpp:.0f} B/point; bit-identical to one stream: {same}", flush=True)
