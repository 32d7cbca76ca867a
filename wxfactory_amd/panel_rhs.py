"""Shared driver of a cubed-sphere RHS evaluation over the panels one rank owns.

Phase ordering of the reference's RHS template (rhs/rhs.py:88-118) in terms of the two HIP
kernels per panel: extrapolate+pack -> start exchange -> interior elements -> wait -> tile-edge
elements.  Accepts float64 and complex128 states (the complex twin of every plan and of the exchange
buffers is created on first use: matvec_fun's complex step, solvers/matvec.py:56-61).
"""
from typing import Dict

import torch

from . import _lib
from .exchange import PanelExchange


def _ptr_array(items):
    """Four edge buffers (tensors, or raw device addresses) -> void*[4] for the C ABI; None passes NULL."""
    import ctypes

    if items is None:
        return None
    return (ctypes.c_void_p * 4)(*[(t.data_ptr() if isinstance(t, torch.Tensor) else int(t)) for t in items])


class PanelRhs:
    def __init__(self, plans: Dict[int, object], exchange: PanelExchange = None, overlap: bool = True,
                 rank: int = 0, world_size: int = 1, group=None, device=None, edge_count: int = 0,
                 complex_arith: str = "complex", tiles_per_side: int = 1):
        """`device` / `edge_count` are only needed by a rank that owns no panel (plans == {})."""
        if complex_arith not in ("complex", "dual"):
            raise ValueError("complex_arith must be 'complex' (true complex arithmetic) or 'dual' (first order)")
        self.complex_arith = complex_arith
        self.tiles_per_side = exchange.topo.k if exchange is not None else tiles_per_side
        self.panels = sorted(plans)  # tile ids (k = 1: panel ids)
        self.overlap = overlap
        self.rank, self.world, self.group = rank, world_size, group
        first = plans[self.panels[0]] if self.panels else None
        self.device = first.device if first is not None else device
        self._plans = {first.dtype if first is not None else torch.float64: plans}
        self._ex = {}
        # shared by every exchange this object creates: {"inline": True} makes them all stream-ordered (graph capture)
        self.comm_mode = {"inline": False}
        if exchange is not None:
            exchange.mode = self.comm_mode
            self._ex[first.dtype if first is not None else torch.float64] = exchange
            self.rank, self.world, self.group = exchange.rank, exchange.world, exchange.group
        self.edge_count = first.edge_count if first is not None else edge_count
        self.panel_shape = first.shape if first is not None else None

    @property
    def reduce_group(self):
        """Who the callers' reductions run over (reduce.py): the library's communicator when the exchange is its own
        (backend "rccl": halo exchange and reductions on ONE communicator, as the reference has it), else the
        torch.distributed group of the exchange."""
        first = next(iter(self._ex.values()), None)
        comm = getattr(first, "comm", None) if first is not None and first.backend == "rccl" else None
        return comm if comm is not None else self.group

    # -- per-dtype resources
    def plans_for(self, dtype):
        if dtype not in self._plans:
            base = next(iter(self._plans.values()))
            self._plans[dtype] = {p: pl.twin(dtype, dual=self.complex_arith == "dual") for p, pl in base.items()}
        return self._plans[dtype]

    def new_exchange(self, words: int) -> PanelExchange:
        """Another exchange like the one this object was given (same ranks, tiling, backend, communicator, loopback
        rehearsal): the complex twin, the second buffer set of the stage pipeline, the value / tangent sets of the
        prepared JVP."""
        dev = self.device if self.device is not None else "cpu"
        first = next(iter(self._ex.values()), None)
        return PanelExchange(words, dev, rank=self.rank, world_size=self.world, group=self.group,
                             tiles_per_side=self.tiles_per_side, mode=self.comm_mode,
                             loopback=first.loopback if first is not None else False,
                             backend=first.backend if first is not None else "torch",
                             comm=getattr(first, "comm", None))

    def exchange_for(self, dtype):
        if dtype not in self._ex:
            self._ex[dtype] = self.new_exchange(self.edge_count * (2 if dtype.is_complex else 1))
        return self._ex[dtype]

    def set_inline_exchange(self, inline: bool = True):
        """Stream-ordered halo exchange without work handles (and therefore without the INTERIOR / BOUNDARY overlap):
        what a HIP-graph capture of an evaluation over several GPUs needs (graph.GraphedFunction sets it)."""
        self.comm_mode["inline"] = bool(inline)

    @property
    def supports_axpy(self) -> bool:
        """True when the plans can fuse linear combinations into the RHS store (rhs_axpy)."""
        return bool(self.panels) and all(hasattr(pl, "rhs_axpy") for pl in self.plans.values())

    @property
    def supports_axpy2(self) -> bool:
        """... and a second extra array (the finite-difference Jacobian operators need y and z)."""
        return self.supports_axpy and all(getattr(pl, "axpy_two", False) for pl in self.plans.values())

    @property
    def plans(self):
        return next(iter(self._plans.values()))

    @property
    def ex(self):
        return self.exchange_for(next(iter(self._plans)))

    # -- the evaluation
    def __call__(self, qs, dtype=None):
        return self._run(qs, None, None, dtype, None)

    def axpy(self, qs, ys, a: float, b: float, c: float, zs=None, d: float = 0.0):
        """a*ys + b*qs + c*R(qs) [+ d*zs] with the update fused into the RHS kernel's store (an explicit
        Runge-Kutta stage, or the finite-difference Jacobian operators of solvers/matvec.py); `ys`, `zs`
        may be None.  Same structures as __call__."""
        return self._run(qs, ys, (float(a), float(b), float(c), float(d)), None, zs)

    def _structure(self, qs):
        np_ = len(self.panels)
        kind = "dict"
        if isinstance(qs, torch.Tensor):
            per = 1
            for s in self.panel_shape:
                per *= s
            if qs.numel() == per and np_ == 1:
                kind, shape = "single", qs.shape
                qs = {self.panels[0]: qs}
            elif qs.numel() == per * np_:
                kind, shape = "stacked", qs.shape
                flat = qs.reshape((np_,) + tuple(self.panel_shape))
                qs = {p: flat[i] for i, p in enumerate(self.panels)}
            else:
                raise ValueError(f"state of {qs.numel()} values does not match {np_} panel(s) of {self.panel_shape}")
        return kind, (shape if kind != "dict" else None), qs

    # -- the reference's timing interface (rhs/rhs.py:39-40, 69-121): nine device timestamps per evaluation,
    #    `timings` = lists of the eight intervals + total, consumed by Integrator.step (integrators/integrator.py:99-106)
    timed = False  # opt-in: nine event records per evaluation

    def clear_timings(self):
        self.timestamps, self.timings = [], []

    def retrieve_last_times(self):
        """Append the intervals (seconds) between the last evaluation's timestamps, and their total.  With the
        phases fused into two kernels the nine stamps of the reference map to: 0 start, 1 extrapolation + pack
        done, 2 exchange started, 3 = 4 interior launch done (pointwise fluxes and their divergence are one kernel),
        5 exchange complete, 6 = 7 = 8 boundary (or whole-tile) launch done (Riemann, correction, forcing)."""
        ts = getattr(self, "timestamps", [])
        if not ts:
            return
        if not hasattr(self, "timings"):
            self.timings = []
        ts[-1].synchronize()
        out = [ts[i].elapsed_time(ts[i + 1]) * 1e-3 for i in range(len(ts) - 1)]
        out.append(ts[0].elapsed_time(ts[-1]) * 1e-3)
        self.timings.append(out)
        self.timestamps = []

    def _stamp(self, *slots):
        if self.timed and self.device is not None and torch.device(self.device).type == "cuda":
            if len(getattr(self, "timestamps", [])) != 9:
                self.timestamps = [None] * 9
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            for i in slots:
                self.timestamps[i] = ev

    def _phases(self, ex: PanelExchange, run):
        """The phase order of rhs/rhs.py:88-118 once the edge messages are packed: start the exchange, evaluate
        the elements that need no halo (INTERIOR) while it is in flight, then the ring (BOUNDARY); when nothing
        travels (all neighbours on this rank: the halos alias the packed buffers) one launch covers ALL.
        `run(region)` enqueues the kernels of that region on the current stream.
        The library's own exchange (backend "rccl") keeps the grouped sends / receives on the compute stream and forks
        the INTERIOR launches to the exchange's second stream instead (include/wxhip.h, STREAM CAPTURE: the arrangement
        that also records into a HIP graph on the runtime torch ships)."""
        self._stamp(1)
        if ex.needs_comm and self.overlap and not ex.is_inline:
            if ex.native and not self.timed:
                ex.fork()
                with torch.cuda.stream(ex.comm_stream):
                    run(_lib.WX_REGION_INTERIOR)
                ex.start(on_compute=True)
                run(_lib.WX_REGION_BOUNDARY)
                ex.join()
                return
            ex.start()
            self._stamp(2)
            run(_lib.WX_REGION_INTERIOR)
            self._stamp(3, 4)
            ex.wait()
            self._stamp(5)
            run(_lib.WX_REGION_BOUNDARY)
        else:
            ex.start(on_compute=True)   # nothing to overlap with: in stream order
            self._stamp(2, 3, 4)
            ex.wait()
            self._stamp(5)
            run(_lib.WX_REGION_ALL)
        self._stamp(6, 7, 8)

    def _exchange_and_launch(self, ex: PanelExchange, launch):
        """`launch(i, tile, halo_or_None, region)` enqueues one tile's kernel; see _phases."""
        def run(region):
            for i, p in enumerate(self.panels):
                launch(i, p, None if region == _lib.WX_REGION_INTERIOR else ex.halo_views(p), region)

        self._phases(ex, run)

    overlapped_entry = None   # name of the library's "pack, start, INTERIOR, wait, BOUNDARY" entry point for these plans

    def _run_native(self, plans, ex, flat, outs):
        """wx_euler3d_rhs_overlapped / wx_sw_rhs_overlapped (include/wxhip.h): the exchange behind the C ABI on its own
        communication stream, the launches of every local tile, one host call - the Python of rhs/rhs.py:88-118 gone."""
        import ctypes

        n = len(self.panels)
        for p in self.panels:
            plans[p]._check_q(flat[p])
            plans[p].faces_epoch = getattr(plans[p], "faces_epoch", 0) + 1
        handles = (ctypes.c_void_p * n)(*[plans[p]._h for p in self.panels])
        qp = (ctypes.c_void_p * n)(*[flat[p].data_ptr() for p in self.panels])
        rp = (ctypes.c_void_p * n)(*[outs[p].data_ptr() for p in self.panels])
        cs = torch.cuda.current_stream(self.device).cuda_stream
        ms = ex.comm_stream.cuda_stream if ex.comm_stream is not None else None   # the second stream: INTERIOR launches
        lib = _lib.load()
        _lib.check(getattr(lib, self.overlapped_entry)(handles, n, ex._native, qp, rp, cs, ms), self.overlapped_entry)

    def _run(self, qs, ys, coef, dtype, zs=None):
        np_ = len(self.panels)
        kind, shape, qs = self._structure(qs)
        if ys is not None:
            ys = self._structure(ys)[2]
        if zs is not None:
            zs = self._structure(zs)[2]
        if not self.panels:
            # a rank that owns no panel (ranks 6, 7 of an 8-GPU node) still takes part in the collective
            ex = self.exchange_for(dtype or torch.float64)
            if ex.needs_comm:
                ex.start(on_compute=True)
                ex.wait()
            return qs
        dtype = next(iter(qs.values())).dtype
        plans, ex = self.plans_for(dtype), self.exchange_for(dtype)
        if self.timed:
            if getattr(self, "timestamps", None):
                self.retrieve_last_times()  # rhs.py:78-79: timing of the previous call
            self._stamp(0)
        shapes = {p: q.shape for p, q in qs.items()}
        flat = {p: q.reshape(self.panel_shape) for p, q in qs.items()}
        yflat = {p: y.reshape(self.panel_shape) for p, y in ys.items()} if ys is not None else {}
        zflat = {p: z.reshape(self.panel_shape) for p, z in zs.items()} if zs is not None else {}

        def launch(p, halo, region):
            if coef is None:
                plans[p].rhs(flat[p], halo, outs[p], region)
            else:
                plans[p].rhs_axpy(flat[p], halo, yflat.get(p), outs[p], coef[0], coef[1], coef[2], region,
                                  zflat.get(p), coef[3])

        if kind == "stacked":
            out_all = torch.empty((np_,) + tuple(self.panel_shape), dtype=dtype, device=self.device)
            outs = {p: out_all[i] for i, p in enumerate(self.panels)}
        else:
            outs = {p: torch.empty_like(flat[p]) for p in self.panels}
        if (coef is None and getattr(ex, "_native", None) is not None and self.overlapped_entry and not self.timed
                and (self.overlap or not ex.needs_comm) and not ex.is_inline):
            self._run_native(plans, ex, flat, outs)   # the whole evaluation of this rank from one call of the C ABI
        else:
            for p in self.panels:
                plans[p].extrap_pack(flat[p], ex.send_views(p))
            self._exchange_and_launch(ex, lambda i, p, halo, region: launch(p, halo, region))
        if kind == "stacked":
            return out_all.reshape(shape)
        if kind == "single":
            return outs[self.panels[0]].reshape(shape)
        return {p: outs[p].reshape(shapes[p]) for p in self.panels}

    full = __call__
