#!/usr/bin/env python3
"""Probe (development tool): which capture of a torch process dies in hipStreamEndCapture when the library's RCCL exchange is
forked to its communication stream inside the capture?  Step-by-step prints.  PROBE_STEPS picks what runs:
  a  one capture of R(Q), nothing before it          b  three eager evaluations before the capture (GraphedFunction's warm-up)
  c  a second capture on the same exchange            d  torch.cuda.graph on the CURRENT stream's pool without GraphedFunction"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests.gpu_util import make_plan, to_dev  # noqa: E402
from tests.util import golden  # noqa: E402
from wxfactory_amd.exchange import PanelExchange, RcclComm  # noqa: E402
from wxfactory_amd.rhs_euler3d import RhsEuler3D  # noqa: E402

say = lambda *a: print(*a, flush=True)  # noqa: E731
DEV = "cuda:0"
steps = os.environ.get("PROBE_STEPS", "a")
comm = RcclComm(0, 1, device=DEV)
g = golden("euler3d_c31p_n3_h4_v2")
plans = {p: make_plan(g, p) for p in range(6)}
Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm)
plain = RhsEuler3D(plans)
want = plain(Q)
rhs = RhsEuler3D(plans, ex, overlap=True)
rhs.batched = False
if os.environ.get("PROBE_RAW_COMM_STREAM") == "1":
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1) == 0   # hipStreamNonBlocking
    ex.comm_stream = torch.cuda.ExternalStream(raw.value, device=DEV)
say("steps", steps, "comm stream", ex.comm_stream)


def capture(tag, warm):
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            rhs(Q)
        side.synchronize()
        say(tag, "begin capture on", side)
        with torch.cuda.graph(graph, stream=side):
            out = rhs(Q)
        say(tag, "capture ended")
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    say(tag, "replay == eager:", bool(torch.equal(out, want)))
    return graph


graphs = []
if "a" in steps:
    graphs.append(capture("a", 0))
if "b" in steps:
    graphs.append(capture("b", 3))
if "c" in steps:
    graphs.append(capture("c", 0))
say("done")
del graphs
torch.cuda.synchronize()
