"""HIP-graph replay of launch-bound evaluations (single rank).

The kernels of this package are plain launches on torch's current stream, so a whole R(Q) - or a
Krylov matvec, or a Runge-Kutta step - can be captured once into a HIP graph and replayed with one
host call (the reference pays Python + launch overhead per array expression; SURVEY.md section 7 hard
part 9 / BASELINE config 5 "hipGraph-captured matvec").  Capture goes through torch.cuda.graph so
that tensors allocated inside the captured function come from a graph-private pool.  Collectives are not
captured: use this on the single-GPU path (world size 1) only.
"""
from typing import Callable

import torch


class GraphedFunction:
    """fn(*static_inputs) captured once; __call__ copies new inputs into the static buffers, replays,
    and returns the static output (valid until the next call)."""

    def __init__(self, fn: Callable, *example_inputs: torch.Tensor, warmup: int = 3):
        self.inputs = [x.clone() for x in example_inputs]
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # lazy state (complex twins, batches, exchange buffers) is built here
                fn(*self.inputs)
            side.synchronize()
            with torch.cuda.graph(self.graph, stream=side):
                self.output = fn(*self.inputs)
        torch.cuda.current_stream().wait_stream(side)

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.output
