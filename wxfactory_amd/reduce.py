"""Reductions over the ranks for the callers of the RHS (Krylov solvers, NaN flag, collective decisions).

Replaces the MPI allreduce calls of reference wx_factory/solvers/global_operations.py:14-36, solvers/kiops.py:165-200,
solvers/pmex.py:150-173, solvers/fgmres.py:41 and simulation.py:399-408.  `group` names who takes part:

* an RcclComm (wxfactory_amd.exchange) - the library's own communicator, the one the halo exchange runs on: the reduction
  is an ncclAllReduce behind the C ABI (wx_comm_allreduce) on torch's current stream, records into a HIP-graph capture,
  and needs no torch.distributed process group.  This is the several-GPU data path;
* a torch.distributed group, or None for the default group when one is initialised (gloo in the CPU tests);
* None without an initialised process group: one rank, nothing to reduce.
"""
import torch
import torch.distributed as dist

_TORCH_OPS = {"sum": "SUM", "max": "MAX", "min": "MIN"}


def is_comm(group) -> bool:
    """A communicator of the library (duck-typed: allreduce(tensor, op) and world)."""
    return group is not None and hasattr(group, "allreduce") and hasattr(group, "world")


def world_size(group=None) -> int:
    if is_comm(group):
        return int(group.world)
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def allreduce(t: torch.Tensor, group=None, op: str = "sum") -> torch.Tensor:
    """In place; returns t.  One rank: nothing happens."""
    if is_comm(group):
        if group.world > 1 or getattr(group, "always", False):
            group.allreduce(t, op)
        return t
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        red = getattr(dist.ReduceOp, _TORCH_OPS[op])
        if t.is_cuda and dist.get_backend(group) == "gloo":
            # a host-only process group and a device tensor: through a host copy (tests and checks; the data path of several
            # GPUs is the library's communicator above)
            h = t.cpu()
            dist.all_reduce(h, op=red, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=red, group=group)
    return t


def capturable(group=None) -> bool:
    """True when a reduction over `group` can be recorded into a HIP graph: the library's communicator (a graph node on the
    capture's origin stream), or a single rank (no reduction at all).  torch.distributed collectives are kept out of
    captures (the process group's own threads and streams; profiles/r05_process_group_abort.md)."""
    return is_comm(group) or world_size(group) == 1
