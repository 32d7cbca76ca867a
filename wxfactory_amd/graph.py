"""HIP-graph replay of launch-bound evaluations (one GPU or one rank of several).

The kernels of this package are plain launches on torch's current stream, so a whole R(Q) - or a
Krylov matvec, or a Runge-Kutta step - can be captured once into a HIP graph and replayed with one
host call (the reference pays Python + launch overhead per array expression; SURVEY.md section 7 hard
part 9 / BASELINE config 5 "hipGraph-captured matvec").  Capture goes through torch.cuda.graph so
that tensors allocated inside the captured function come from a graph-private pool.

Over several GPUs the halo exchange is part of the evaluation (the reference's Krylov vector = matvec + exchange +
reductions, solvers/kiops.py:170-207, process_topology.py:318): pass the RHS object as `rhs=` and its exchanges switch to
their stream-ordered form (PanelRhs.set_inline_exchange: fixed buffers, no work handle, one launch for the whole tile
instead of INTERIOR / BOUNDARY), which records into the graph - "tangent extrapolation -> RCCL exchange -> JVP kernels"
replays from one host call on every rank.  Every rank must capture and replay the same sequence; delete the graph before
the process group is destroyed.  Anything that reads a device value on the host (the collective decision of
jvp_prepare, a norm) has to happen outside the captured function.
"""
from typing import Callable

import torch


class GraphedFunction:
    """fn(*static_inputs) captured once; __call__ copies new inputs into the static buffers, replays,
    and returns the static output (valid until the next call)."""

    def __init__(self, fn: Callable, *example_inputs: torch.Tensor, warmup: int = 3, rhs=None):
        if rhs is not None and hasattr(rhs, "set_inline_exchange"):
            rhs.set_inline_exchange(True)   # stays on: replays and later eager calls share the buffers' protocol
        self.inputs = [x.clone() for x in example_inputs]
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # lazy state (complex twins, batches, exchange buffers) is built here
                fn(*self.inputs)
            side.synchronize()
            # thread-local capture mode: with a process group alive, torch's communication watchdog thread polls events
            # while this thread captures; in the default (global) mode such a call from ANOTHER thread is an error there, and
            # the watchdog answers an error by aborting the process (seen twice in 12 runs of the GPU suite, round 4)
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                self.output = fn(*self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        # a graph that holds RCCL nodes (R(Q) or a matvec over the library's exchange) must go before its communicator:
        # ncclCommDestroy under such a graph does not return (profiles/r04_capture_crash.md).  RcclComm.close() resets
        # what is registered with it.
        holders = []
        try:
            holders = [getattr(rhs, "reduce_group", None), getattr(getattr(rhs, "ex", None), "comm", None)]
        except Exception:   # noqa: BLE001 - an RHS object without plans yet
            pass
        for holder in holders:
            if holder is not None and hasattr(holder, "register_graph"):
                holder.register_graph(self.graph)

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.output
