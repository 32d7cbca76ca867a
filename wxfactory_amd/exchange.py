"""Panel-edge halo exchange for the panels a rank owns.

Replaces reference wx_factory/process_topology.py:269-386 + 564-606 (start_exchange_scalars /
start_exchange_vectors / ExchangeRequest.wait): there, three Ineighbor_alltoall messages per
RHS carry (rho), (rho u1, rho u2, rho w), (rho theta); here ONE message per panel edge carries
all five variables, already rotated and flipped by the pack kernel (wx_euler3d_extrap_pack).

* neighbours on the same rank: zero copy - the receiver's halo pointer aliases the sender's send slot;
* backend "rccl" - THE data path over several GPUs: the library's own exchange behind the C ABI (csrc/exchange.hip,
  wx_exchange_* of include/wxhip.h: grouped ncclSend / ncclRecv on a communicator made by wx_comm_init_rank, RcclComm
  below) with no torch.distributed call.  Overlap = the exchange on the compute stream and the INTERIOR launches forked
  to a second stream (PanelRhs._phases); that shape also records into a HIP graph (RCCL on the capture's origin stream).
  The callers' small reductions run on the same communicator (RcclComm.allreduce, reduce.py);
* backend "torch" - host logic and checks: torch.distributed.all_to_all_single on a gloo group, for CPU buffers (the
  world_size > 1 tests that pin the slot layout against the halos the reference delivered) or, staged through host
  copies, for device buffers (the independent route bench.py checks the library's exchange against).  Device buffers on a
  torch NCCL group are refused: no NCCL process group exists anywhere in this package's processes
  (profiles/r05_process_group_abort.md, profiles/r04_capture_crash.md).
"""
import ctypes
import os
import weakref
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

from .panels import CubeTopology, owner_of_tiles


_REDUCE_OPS = {"sum": 0, "max": 1, "min": 2}   # wx_reduce_op


def share_comm_id(ident: bytes, rank: int, world_size: int, group=None, store=None, key: str = "wx_comm_id") -> bytes:
    """Rank 0's communicator id on every rank - the only thing the library's communicator needs from the outside
    (the reference gets its communicator from MPI, process_topology.py:259-261).  Through `store` (a torch.distributed
    Store, e.g. TCPStore: no process group at all) or through `group` (any torch.distributed group - gloo; None = the
    default group).  One rank: returned as is."""
    if world_size <= 1:
        return ident
    if store is not None:
        if rank == 0:
            store.set(key, ident)
        return bytes(store.get(key))   # (get blocks until the key is there)
    box = [ident]
    dist.broadcast_object_list(box, src=0, group=group)
    return bytes(box[0])


class RcclComm:
    """An RCCL communicator of this library's own (wx_comm_unique_id / wx_comm_init_rank, include/wxhip.h): the halo
    exchange of backend "rccl" AND the small reductions of the Krylov callers (allreduce below) run on it through the C
    ABI, with no torch.distributed call on the data path - the one communicator the reference builds from MPI
    (process_topology.py:259-261, solvers/global_operations.py:14-36).

    Bootstrap: only the 128-byte unique id has to travel from rank 0 to the others, once.  Give either `group` - any
    torch.distributed group, normally gloo: NO NCCL process group is needed or wanted (its watchdog thread issues HIP event
    calls beside this process's graph captures, profiles/r05_process_group_abort.md) - or `store` - a torch.distributed
    Store such as TCPStore, for processes that have no process group at all; one rank needs neither.
    Collective; the calling process must have its GPU current (torch.cuda.set_device)."""

    def __init__(self, rank: int = 0, world_size: int = 1, group=None, device=None, store=None, key: str = "wx_comm_id"):
        from . import _lib

        self.lib = _lib.load()
        self.rank, self.world = rank, world_size
        refuse_nccl_process_group("RcclComm")
        ident = (ctypes.c_ubyte * _lib.WX_COMM_ID_BYTES)()
        if rank == 0:
            _lib.check(self.lib.wx_comm_unique_id(ident), "wx_comm_unique_id")
        ident = (ctypes.c_ubyte * _lib.WX_COMM_ID_BYTES)(*share_comm_id(bytes(ident), rank, world_size, group, store, key))
        self._h = ctypes.c_void_p()
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            _lib.check(self.lib.wx_comm_init_rank(ctypes.byref(self._h), world_size, ident, rank), "wx_comm_init_rank")
        self.device = dev
        # what lives on this communicator and has to go before it: exchanges (they keep the wx_comm*), and HIP graphs that
        # hold RCCL nodes (ncclCommDestroy under such a graph does not return, profiles/r04_capture_crash.md)
        self.always = False   # tests: issue the reductions on one rank too (reduce.allreduce skips them otherwise)
        self._exchanges = weakref.WeakSet()
        self._graphs = weakref.WeakSet()

    @property
    def version(self) -> int:
        return int(self.lib.wx_comm_rccl_version())

    @property
    def hip_runtime_version(self) -> int:
        """The HIP runtime this process bound (inside torch: the wheel's, not the one libwxhip.so was compiled with)."""
        return int(self.lib.wx_hip_runtime_version())

    def allreduce(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        """In-place all-reduce of a device tensor on torch's current stream (ncclAllReduce behind wx_comm_allreduce): a node
        of the graph when that stream is being captured.  float64 tensors are reduced where they are; anything else
        (an int32 NaN flag, an int64 length) goes through a float64 copy - exact for integers below 2^53."""
        from . import _lib

        if not t.is_cuda:
            raise ValueError("RcclComm.allreduce reduces device tensors")
        if t.numel() == 0:
            return t
        st = torch.cuda.current_stream(t.device).cuda_stream
        if t.dtype == torch.float64 and t.is_contiguous():
            _lib.check(self.lib.wx_comm_allreduce(self._h, t.data_ptr(), t.numel(), _REDUCE_OPS[op], st), "wx_comm_allreduce")
            return t
        if t.dtype.is_complex:
            if op != "sum":
                raise ValueError("complex tensors can only be summed")
            buf = torch.view_as_real(t.contiguous()).contiguous()
        else:
            buf = t.to(torch.float64).contiguous()
        _lib.check(self.lib.wx_comm_allreduce(self._h, buf.data_ptr(), buf.numel(), _REDUCE_OPS[op], st), "wx_comm_allreduce")
        t.copy_(torch.view_as_complex(buf).reshape(t.shape) if t.dtype.is_complex else buf.reshape(t.shape).to(t.dtype))
        return t

    def register_graph(self, graph):
        """A HIP graph that holds RCCL nodes of this communicator: close() resets it before the communicator goes."""
        self._graphs.add(graph)

    def close(self):
        """Graphs with RCCL nodes first, then the exchanges, then the communicator (wx_comm_destroy refuses otherwise)."""
        if not self._h:
            return
        for g in list(self._graphs):
            try:
                g.reset()
            except Exception:   # noqa: BLE001 - a graph that is gone already
                pass
        for ex in list(self._exchanges):
            ex.close()
        from . import _lib

        h, self._h = self._h, ctypes.c_void_p()
        _lib.check(self.lib.wx_comm_destroy(h), "wx_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def refuse_nccl_process_group(who: str):
    """A torch.distributed NCCL process group in this process brings a watchdog thread that polls HIP events beside this
    package's graph captures; the HIP runtime inside the torch wheel answers such a call by invalidating the capture and failing
    the query, and the watchdog answers an error by aborting the process (profiles/r05_process_group_abort.md).  The library's
    communicator and a NCCL group do not share a process: refused here, before any capture starts, not at the first exchange."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    backends = set()
    try:
        backends.add(str(dist.get_backend()))
        pg_map = getattr(dist.distributed_c10d, "_world", None)
        for pg in list(getattr(pg_map, "pg_map", {}) or {}):
            backends.add(str(dist.get_backend(pg)))
    except Exception:   # noqa: BLE001 - a group this rank is not part of
        pass
    if any("nccl" in b.lower() for b in backends):
        raise RuntimeError(f"{who}: a torch.distributed NCCL process group is initialised in this process; its watchdog thread's "
                           "HIP event queries abort the process under this package's graph captures - use a gloo group (or a "
                           "TCPStore) for the host side, the library's communicator carries the device data")


class PanelExchange:
    def __init__(self, edge_doubles: int, device, rank: int = 0, world_size: int = 1, group=None,
                 loopback: bool = False, tiles_per_side: int = 1, mode: dict = None, backend: str = "torch",
                 comm: "RcclComm" = None):
        """edge_doubles: float64 words per edge message (5*V*H*n^2, doubled for complex128);
        buffers are float64 (complex payloads travel as interleaved re/im, which RCCL accepts).
        mode: a dict shared by all exchanges of one RHS object; mode["inline"] switches them together.
        backend: "torch" - torch.distributed.all_to_all_single on `group` (RCCL on GPUs, gloo on CPU) - or "rccl" - the
        library's own exchange behind the C ABI (wx_exchange_*: grouped ncclSend / ncclRecv on a communication stream
        forked from the compute stream by an event; `comm` = an RcclComm, not needed when nothing travels)."""
        dtype = torch.float64
        if backend not in ("torch", "rccl"):
            raise ValueError("backend must be 'torch' or 'rccl'")
        self.backend = backend
        self.inline = False          # stream-ordered collective, no work handle (graph capture)
        self.mode = mode
        self.edge_count = int(edge_doubles)
        # loopback (tests): route same-rank messages through the collective too, so that a single
        # process exercises the RCCL path (split sizes, slot order, async wait) end to end
        self.loopback = loopback
        self.rank, self.world = rank, world_size
        self.group = group
        # tiles: 6 k^2 of them (k = 1: one tile per panel); "panel" below means tile id
        self.topo = topo = CubeTopology(tiles_per_side)
        owner = owner_of_tiles(world_size, topo.ntiles)
        self.owner = owner
        self.local = [p for p in range(topo.ntiles) if owner[p] == rank]

        # messages this rank sends: (src panel, src edge) -> (dst rank, dst panel, dst edge)
        local_msgs, remote_out, remote_in = [], {}, {}
        for p in self.local:
            for e in range(4):
                q, e2 = topo.neighbor(p, e), topo.landing(p, e)
                if owner[q] == rank and not loopback:
                    local_msgs.append((p, e, q, e2))
                else:
                    remote_out.setdefault(owner[q], []).append((q, e2, p, e))
        for q in self.local:
            for e2 in range(4):
                p = topo.neighbor(q, e2)
                if owner[p] != rank or loopback:
                    remote_in.setdefault(owner[p], []).append((q, e2))
        # canonical order inside each rank pair: by (destination panel, destination edge)
        n_remote_out = sum(len(v) for v in remote_out.values())
        n_remote_in = sum(len(v) for v in remote_in.values())
        ec = self.edge_count
        self.send_buf = torch.zeros((n_remote_out + len(local_msgs)) * ec, dtype=dtype, device=device)
        self.recv_buf = torch.zeros(max(n_remote_in, 1) * ec, dtype=dtype, device=device)
        self._send_slot: Dict[Tuple[int, int], int] = {}
        self._halo_src: Dict[Tuple[int, int], Tuple[str, int]] = {}
        self.send_splits, self.recv_splits = [0] * world_size, [0] * world_size
        slot = 0
        for r in range(world_size):
            msgs = sorted(remote_out.get(r, []))
            self.send_splits[r] = len(msgs) * ec
            for (q, e2, p, e) in msgs:
                self._send_slot[(p, e)] = slot
                slot += 1
        self.n_remote_out = slot
        for (p, e, q, e2) in local_msgs:
            self._send_slot[(p, e)] = slot
            self._halo_src[(q, e2)] = ("send", slot)
            slot += 1
        rslot = 0
        for r in range(world_size):
            msgs = sorted(remote_in.get(r, []))
            self.recv_splits[r] = len(msgs) * ec
            for (q, e2) in msgs:
                self._halo_src[(q, e2)] = ("recv", rslot)
                rslot += 1
        self.n_remote_in = rslot
        self._work = None
        self._native = None
        if backend == "rccl":
            self._bind_native(comm, device, tiles_per_side)

    def _bind_native(self, comm, device, k):
        """The C-ABI exchange over the SAME two buffers: the library derives the tile graph, the ownership and the slot
        order itself (csrc/wx_panels.h) - every send / halo address it names must be the one this class computed."""
        from . import _lib

        self.lib = _lib.load()
        self.comm = comm
        if self.needs_comm and comm is None:
            raise ValueError("backend='rccl': messages travel, an RcclComm is needed")
        if torch.device(device).type != "cuda":
            raise ValueError("backend='rccl' moves device buffers: the exchange must live on a GPU")
        h = ctypes.c_void_p()
        _lib.check(self.lib.wx_exchange_create(ctypes.byref(h), comm._h if comm is not None else None, self.rank, self.world,
                                               k, self.edge_count, int(self.loopback)), "wx_exchange_create")
        self._native = h
        if comm is not None:
            comm._exchanges.add(self)   # (RcclComm.close() closes us first: the library keeps the wx_comm* in the exchange)
        if self.send_buf.numel() != self.lib.wx_exchange_send_doubles(h) or self.recv_buf.numel() != self.lib.wx_exchange_recv_doubles(h):
            raise RuntimeError("the library's edge-buffer sizes differ from the host mirror's")
        _lib.check(self.lib.wx_exchange_bind(h, self.send_buf.data_ptr() if self.send_buf.numel() else None,
                                             self.recv_buf.data_ptr()), "wx_exchange_bind")
        for p in self.local:
            for e in range(4):
                if (self.lib.wx_exchange_send_ptr(h, p, e) != self.send_view(p, e).data_ptr()
                        or self.lib.wx_exchange_halo_ptr(h, p, e) != self.halo_view(p, e).data_ptr()):
                    raise RuntimeError(f"the library's slot of tile {p}, edge {e} differs from the host mirror's")
        # the second stream of the overlapped evaluation (one per exchange object; the join is an event wait): it carries the
        # INTERIOR launches, which fill every CU.  Its priority was MEASURED (profiles/r06_overlap_priority_ab.txt): at normal
        # priority the grouped sends / receives enqueued behind INTERIOR complete 50 us after INTERIOR's start - the dispatcher
        # serves the queues in turn, a workgroup of the fused kernel lives a few microseconds -, while on a lowest-priority stream
        # (include/wxhip.h: wx_stream_create) INTERIOR itself runs 23 % slower with nothing beside it.  Default: normal;
        # WXHIP_SIDE_PRIORITY=low selects the low-priority stream.
        self.comm_stream, self._side = None, None
        if self.needs_comm:
            if os.environ.get("WXHIP_SIDE_PRIORITY", "normal") == "low":
                side = ctypes.c_void_p()
                with torch.cuda.device(device):
                    _lib.check(self.lib.wx_stream_create(ctypes.byref(side), 1), "wx_stream_create")
                self._side = side
                self.comm_stream = torch.cuda.ExternalStream(side.value, device=device)
            else:
                self.comm_stream = torch.cuda.Stream(device=device)
        self.side_priority = "lowest" if self._side is not None else "normal"

    def close(self):
        if self._native:
            self.lib.wx_exchange_destroy(self._native)
            self._native = None
        if getattr(self, "_side", None) is not None:
            self.comm_stream = None
            self.lib.wx_stream_destroy(self._side)
            self._side = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- views (tests / host logic) and raw pointers (kernels)
    def send_view(self, panel: int, edge: int) -> torch.Tensor:
        s = self._send_slot[(panel, edge)]
        return self.send_buf[s * self.edge_count:(s + 1) * self.edge_count]

    def halo_view(self, panel: int, edge: int) -> torch.Tensor:
        kind, s = self._halo_src[(panel, edge)]
        buf = self.send_buf if kind == "send" else self.recv_buf
        return buf[s * self.edge_count:(s + 1) * self.edge_count]

    def send_views(self, panel: int) -> List[torch.Tensor]:
        return [self.send_view(panel, e) for e in range(4)]

    def halo_views(self, panel: int) -> List[torch.Tensor]:
        return [self.halo_view(panel, e) for e in range(4)]

    def send_ptrs(self, panel: int) -> List[int]:
        return [self.send_view(panel, e).data_ptr() for e in range(4)]

    def halo_ptrs(self, panel: int) -> List[int]:
        return [self.halo_view(panel, e).data_ptr() for e in range(4)]

    @property
    def needs_comm(self) -> bool:
        return self.world > 1 or self.loopback

    @property
    def is_inline(self) -> bool:
        return self.inline or bool(self.mode and self.mode.get("inline"))

    @property
    def native(self) -> bool:
        """The library's own exchange behind the C ABI (backend "rccl") carries the messages."""
        return self._native is not None

    def fork(self):
        """backend "rccl": fork the exchange's second stream off the current one (it will carry the INTERIOR launches)."""
        from . import _lib

        cs = torch.cuda.current_stream(self.send_buf.device).cuda_stream
        _lib.check(self.lib.wx_exchange_fork(self._native, cs, self.comm_stream.cuda_stream), "wx_exchange_fork")

    def join(self):
        """... and make the current stream wait for what was enqueued on the second stream since."""
        from . import _lib

        cs = torch.cuda.current_stream(self.send_buf.device).cuda_stream
        _lib.check(self.lib.wx_exchange_join(self._native, cs, self.comm_stream.cuda_stream), "wx_exchange_join")

    def start(self, on_compute: bool = False):
        """Post the exchange of everything the pack kernels wrote (stream-ordered after them).  Inline mode: the
        collective is complete, in stream order, when this returns - nothing to wait for, nothing kept."""
        if not self.needs_comm:
            return
        if self._native is not None:
            # on the current stream (inline, or the arrangement whose INTERIOR launches were forked instead), or forked
            # to the communication stream by an event (csrc/exchange.hip)
            from . import _lib

            cs = torch.cuda.current_stream(self.send_buf.device).cuda_stream
            # (the second stream is a LOW-priority one, made for the INTERIOR launches: the exchange itself never goes there - a
            # caller that asks for it, the timed evaluation's stamps, gets the exchange in stream order on the compute stream)
            ms = cs if (self.is_inline or on_compute or self._side is not None) else self.comm_stream.cuda_stream
            _lib.check(self.lib.wx_exchange_start(self._native, cs, ms), "wx_exchange_start")
            return
        ec = self.edge_count
        send = self.send_buf[: self.n_remote_out * ec]
        recv = self.recv_buf[: self.n_remote_in * ec]
        if send.is_cuda and dist.get_backend(self.group) != "gloo":
            # (a torch NCCL process group brings a watchdog thread that issues HIP event calls beside this process's graph
            # captures - an intermittent SIGABRT in round 4, profiles/r05_process_group_abort.md - and an internal stream
            # that the bundled HIP runtime cannot capture; the library's own communicator has neither)
            raise RuntimeError("device buffers travel on the library's own communicator: PanelExchange(backend='rccl', "
                               "comm=RcclComm(...)); a torch.distributed NCCL group is not used on the data path")
        if send.is_cuda:
            # device buffers and a host-only process group: staged through host copies, synchronously.  NOT a data path -
            # it is the independent second route bench.py checks the library's exchange against before it times anything
            # (the gloo route is the one tests/test_exchange_gloo.py pins against the halos the reference delivered), with
            # no NCCL process group in the process.
            if self.is_inline:
                raise RuntimeError("the host-staged exchange cannot be captured; use backend='rccl'")
            send_h, recv_h = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(recv_h, send_h, output_split_sizes=self.recv_splits, input_split_sizes=self.send_splits,
                                   group=self.group)
            recv.copy_(recv_h)
            self._work = None
            return
        inline = self.is_inline
        work = dist.all_to_all_single(
            recv, send, output_split_sizes=self.recv_splits, input_split_sizes=self.send_splits,
            group=self.group, async_op=not inline,
        )
        self._work = None if inline else work

    def wait(self):
        """Make the current stream (GPU) / the caller (CPU) wait for the halos."""
        if self._native is not None:
            from . import _lib

            _lib.check(self.lib.wx_exchange_wait(self._native, torch.cuda.current_stream(self.send_buf.device).cuda_stream),
                       "wx_exchange_wait")
            return
        if self._work is not None:
            self._work.wait()
            self._work = None
