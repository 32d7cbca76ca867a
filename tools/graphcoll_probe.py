#!/usr/bin/env python3
"""Probe (development tool): can a whole evaluation INCLUDING the RCCL halo exchange be captured into one HIP graph?
One GPU, 1-rank RCCL group, exchange in loopback mode (every edge message through all_to_all_single)."""
import os
import socket
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.gpu_util import make_plan, to_dev  # noqa: E402
from tests.util import golden  # noqa: E402
from wxfactory_amd.exchange import PanelExchange  # noqa: E402
from wxfactory_amd.rhs_euler3d import RhsEuler3D  # noqa: E402

DEV = "cuda:0"
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
g = golden("euler3d_c31p_n3_h4_v2")
plans = {p: make_plan(g, p) for p in range(6)}
ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=True)
rhs = RhsEuler3D(plans, ex, overlap=True)
Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
eager = rhs(Q)
torch.cuda.synchronize()
import faulthandler
faulthandler.enable()
SYNC = os.environ.get("PROBE_SYNC") == "1"
if SYNC:   # collective enqueued with async_op=False: the current stream waits for it at once (no Work kept)
    def start_sync():
        ec = ex.edge_count
        dist.all_to_all_single(ex.recv_buf[: ex.n_remote_in * ec], ex.send_buf[: ex.n_remote_out * ec],
                               output_split_sizes=ex.recv_splits, input_split_sizes=ex.send_splits, group=ex.group)
    ex.start = start_sync
    ex.wait = lambda: None
for mode in (os.environ.get("PROBE_MODE", "global"),):
    try:
        static_q = Q.clone()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                rhs(static_q)
            side.synchronize()
            with torch.cuda.graph(graph, stream=side, capture_error_mode=mode):
                out = rhs(static_q)
        torch.cuda.current_stream().wait_stream(side)
        for scale in (1.0, 1.01):
            static_q.copy_(Q * scale)
            graph.replay()
            torch.cuda.synchronize()
            ref = rhs(Q * scale)
            torch.cuda.synchronize()
            print(mode, scale, "replay == eager:", bool(torch.equal(out, ref)), flush=True)
        break
    except Exception:
        print(mode, "FAILED"); traceback.print_exc()
        torch.cuda.synchronize()
print('deleting graph', flush=True)
del graph, out
torch.cuda.synchronize()
print('destroying group', flush=True)
dist.destroy_process_group()
print('done', flush=True)
