#!/usr/bin/env python3
"""Probe (development tool), second stage: which ingredient of a torch capture makes hipStreamEndCapture of the bundled HIP
runtime recurse without bound when the library's RCCL exchange is forked inside it?  PROBE_VARIANT:
  raw     torch imported and initialised, but streams / capture / graph through the HIP runtime directly (ctypes)
  bare    torch.cuda.graph on a torch side stream, ONLY wx_exchange_start / _wait inside
  empty   bare + a torch.empty inside the capture
  kernel  bare + one of the library's kernel launches (extrap_pack) in front of the start
  interior  bare + an INTERIOR launch between start and wait
  mirror  the arrangement the product uses: the exchange on the ORIGIN stream, the INTERIOR launch forked to the second one"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests.gpu_util import make_plan, to_dev  # noqa: E402
from tests.util import golden  # noqa: E402
from wxfactory_amd import _lib  # noqa: E402
from wxfactory_amd.exchange import PanelExchange, RcclComm  # noqa: E402

say = lambda *a: print(*a, flush=True)  # noqa: E731
DEV = "cuda:0"
variant = os.environ.get("PROBE_VARIANT", "bare")
torch.cuda.init()
x = torch.zeros(8, device=DEV)
comm = RcclComm(0, 1, device=DEV)
g = golden("euler3d_c31p_n3_h4_v2")
plan = make_plan(g, 0)
q = to_dev(g.q(0))
ex = PanelExchange(plan.edge_count, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm)
lib = _lib.load()
say("variant", variant)

if variant == "raw":
    hip = ctypes.CDLL("libamdhip64.so")
    s1, s2, graph, gexec = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s1), 1) == 0 and hip.hipStreamCreateWithFlags(ctypes.byref(s2), 1) == 0
    torch.cuda.synchronize()
    say("begin capture (raw)")
    assert hip.hipStreamBeginCapture(s1, 0) == 0
    _lib.check(lib.wx_exchange_start(ex._native, s1, s2), "start")
    _lib.check(lib.wx_exchange_wait(ex._native, s1), "wait")
    say("end capture (raw)")
    assert hip.hipStreamEndCapture(s1, ctypes.byref(graph)) == 0
    say("ended")
else:
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    out = torch.empty_like(q)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        say("begin capture on", side)
        with torch.cuda.graph(graph, stream=side):
            if variant == "empty":
                t = torch.empty(1000, device=DEV)
            if variant == "kernel":
                plan.extrap_pack(q, ex.send_views(0))
            if variant == "mirror":
                plan.extrap_pack(q, ex.send_views(0))
                ex.fork()
                with torch.cuda.stream(ex.comm_stream):
                    plan.rhs(q, None, out, _lib.WX_REGION_INTERIOR)
                ex.start(on_compute=True)
                plan.rhs(q, ex.halo_views(0), out, _lib.WX_REGION_BOUNDARY)
                ex.join()
            else:
                ex.start()
                if variant == "interior":
                    plan.rhs(q, None, out, _lib.WX_REGION_INTERIOR)
                ex.wait()
        say("capture ended")
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
say("done")
