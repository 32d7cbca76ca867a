#!/usr/bin/env python3
"""Probe (development tool): whole evaluations INCLUDING the RCCL halo exchange captured into HIP graphs - R(Q) and the
prepared complex-step matvec - on one GPU through a 1-rank RCCL group (exchange in loopback mode), step by step with
prints, so that a crash in capture / replay / teardown is attributable.  PROBE_ASYNC=1 keeps the overlapping
(work-handle) form of the collective inside the capture: crashes hipStreamEndCapture on ROCm 7.0 / RCCL 2.26."""
import faulthandler
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.gpu_util import make_plan, to_dev  # noqa: E402
from tests.util import golden  # noqa: E402
from wxfactory_amd.exchange import PanelExchange  # noqa: E402
from wxfactory_amd.graph import GraphedFunction  # noqa: E402
from wxfactory_amd.matvec import ComplexStepOperator, matvec_fun  # noqa: E402
from wxfactory_amd.rhs_euler3d import RhsEuler3D  # noqa: E402

faulthandler.enable()
DEV = "cuda:0"
say = lambda *a: print(*a, flush=True)  # noqa: E731
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
g = golden("euler3d_c31p_n3_h4_v2")
plans = {p: make_plan(g, p) for p in range(6)}
Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
v = torch.stack([to_dev(g[f"p{p}/V"]) for p in range(6)])
plain = RhsEuler3D(plans)
R = plain(Q)
graphs = []
for batched in (True, False):
    gr = RhsEuler3D(plans, PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=True), overlap=True)
    gr.batched = batched
    say("capture R(Q), batched =", batched)
    g_rhs = GraphedFunction(gr, Q, rhs=None if os.environ.get("PROBE_ASYNC") == "1" else gr)
    graphs.append(g_rhs)
    for scale in (1.0, 1.01):
        say("  replay == eager:", bool(torch.equal(g_rhs(Q * scale), plain(Q * scale))))
    say("prepare")
    op = ComplexStepOperator(1.0, Q, R, gr)
    say("capture matvec, prepared =", gr._jvp_is_prepared(Q))
    g_mv = GraphedFunction(op, v.flatten(), rhs=gr)
    graphs.append(g_mv)
    for scale in (1.0, -0.37):
        say("  replay == eager:", bool(torch.equal(g_mv((scale * v).flatten()), matvec_fun((scale * v).flatten(), 1.0, Q, R, plain, "complex"))))
    gr.jvp_release()
say("deleting graphs")
del graphs[:]
g_rhs = g_mv = None
torch.cuda.synchronize()
say("destroying group")
dist.destroy_process_group()
say("done")
