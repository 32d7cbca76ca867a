"""The headline of bench.py: whole-sphere 3-D Euler R(Q), timed between barriers, ONE JSON line on rank 0."""
import argparse
import math
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

from .cpu import cpu_baseline
from .extras import (caller_extras, column_metric_extras, e7_v1_extras, epi2_kiops_e7_extras, extras, ini_size_extras,
                     rhs_benchmark_matrix)
from .roofline import ALGO_BYTES_PER_POINT, HBM_PEAK_GBS, copy_ceiling, mfma_block, pmc_traffic


def decomposition(world, H, tiles_per_side=0, whole_panels=False):
    """(k, owner of each of the 6 k^2 tiles): the reference's own tiling (process_topology.py:69-94) with the smallest
    k that spreads the tiles evenly over the ranks - k = 1 (whole panels) for 1, 2, 3, 6 GPUs, k = 2 for 4 and 8."""
    from wxfactory_amd.panels import CubeTopology, owner_of_tiles, tiles_per_side_for

    k = tiles_per_side or (1 if whole_panels else tiles_per_side_for(world))
    if H % k:
        raise SystemExit(f"H={H} is not divisible by {k} tiles per panel side")
    return k, owner_of_tiles(world, CubeTopology(k).ntiles)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # >= 50 calls ...
    ap.add_argument("--warmup", type=int, default=10)   # ... after >= 5 warm-ups
    ap.add_argument("--n", type=int, default=8, help="num_solpts (p = n-1)")
    ap.add_argument("--H", type=int, default=60, help="elements per panel side")
    ap.add_argument("--V", type=int, default=8, help="vertical elements")
    ap.add_argument("--seed", type=int, default=20250824)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--whole-panels", action="store_true", help="one tile per panel even when 6 does not divide N")
    ap.add_argument("--tiles-per-side", type=int, default=0, help="force k (6 k^2 tiles); default: chosen from N")
    ap.add_argument("--event-every", type=int, default=0,
                    help="0 (default): the timed region carries no instrumentation - `value` is what a caller gets - and the "
                         "kernels are timed in a SECOND pass of the same K steps with HIP events around every launch (the "
                         "roofline block; the two passes' step times are both in the line).  K > 0: one pass, events around "
                         "the launches of every K-th timed step (K = 1 costs the headline ~1 %: two marker packets between "
                         "kernels that would otherwise overlap their tails)")
    ap.add_argument("--compute-priority", choices=("normal", "high"), default="normal",
                    help="priority of the stream the evaluation is enqueued on (pack, exchange, BOUNDARY); the INTERIOR launches of "
                         "the overlapped evaluation go to the exchange's second stream (WXHIP_SIDE_PRIORITY=low: lowest priority)")
    ap.add_argument("--spinup", type=float, default=0.5,
                    help="seconds of untimed evaluations before the W warm-up steps (the device's clock ramp after idle)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (shallow-water S7) measurement")
    ap.add_argument("--loopback", action="store_true",
                    help="rehearsal on one GPU: route every edge message through the RCCL collective (1-rank group) and split "
                         "the evaluation into INTERIOR / BOUNDARY launches, as a multi-GPU run does")
    ap.add_argument("--exchange", choices=("rccl", "torch"), default="rccl",
                    help="halo exchange of the several-GPU path: 'rccl' = the library's own behind the C ABI (wx_exchange_*: "
                         "grouped ncclSend / ncclRecv on a communication stream, event fork / join; the whole evaluation of a "
                         "rank is one wx_euler3d_rhs_overlapped call), 'torch' = torch.distributed.all_to_all_single")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the several-rank program flow on ONE GPU: every rank uses device 0 (needs --exchange "
                         "torch - gloo through host copies -: RCCL refuses two ranks on one device); not a measurement")
    ap.add_argument("--metric", choices=("true", "synthetic"), default="true",
                    help="static metric fields: the cubed-sphere metric of the DCMIP 3-1 planet from wxfactory_amd.geometry3d "
                         "(default, SURVEY 8d) or SURVEY's seeded synthetic fields; values do not affect speed")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit("benchlib.headline: --gpus N needs one rank per GPU - start it through bench.py (or torchrun)")

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # before the first call that initialises the GPU runtime (the host driver supports dmabuf IPC only: without this RCCL's
    # peer mapping fails with hipIpcGetMemHandle: invalid argument); ranks started by an external torchrun land here too
    if world > 1 or args.loopback:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")   # a communicator that cannot connect says why ...
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # ... on stderr (RCCL's default is stdout: the ONE line's stream)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.one_device:
        if args.exchange != "torch" and world > 1:
            raise SystemExit("--one-device: RCCL refuses two ranks on one device, use --exchange torch")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # a HOST-side process group (gloo) for what happens around the measurement: the 128-byte id of the library's
        # communicator, barriers, the max over ranks, the independent route of the exchange self-check.  NO NCCL process group:
        # the halo exchange and every reduction of the data path run on the library's own communicator (wx_comm_*), and a
        # torch NCCL group would add a watchdog thread issuing HIP calls beside the captures (profiles/r05_process_group_abort.md)
        # (gloo announces its connections on STDOUT: fd 1 points at fd 2 meanwhile - this program's stdout is the ONE line)
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    from wxfactory_amd import _lib, synthetic
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    _lib.load()
    n, H, V = args.n, args.H, args.V
    ops = synthetic.dfr_ops(n)
    # the sphere is cut into 6 k^2 tiles (the reference's own decomposition, process_topology.py:69-94) with the
    # smallest k that spreads evenly over the ranks: k = 1 (whole panels) for 1, 2, 3, 6 GPUs, k = 2 for 4 and 8
    k, owner = decomposition(world, H, args.tiles_per_side, args.whole_panels)
    topo = CubeTopology(k)
    Ht = H // k
    mine = [t for t, r in enumerate(owner) if r == rank]
    plans, qs = {}, {}
    t_setup = time.perf_counter()
    for t in mine:
        if args.metric == "true":
            from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch

            panel, row, col = topo.locate(t)
            metric = metric3d_torch(CubedSphere3DTile(n, Ht, V, panel, 10000.0, 31, row=row, col=col, k=k), dev)
        else:
            metric = synthetic.euler3d_metric(n, Ht, V, t, dev, args.seed)
        plans[t] = Euler3DPlan(n, Ht, V, 31, topo.locate(t)[0], ops, metric, on_panel_edge=topo.on_panel_edge(t))
        prc = topo.locate(t)
        qs[t] = synthetic.euler3d_state(n, Ht, V, prc[0], dev, args.seed, row=prc[1], col=prc[2], k=k)   # a cut of the PANEL's state
    t_setup = time.perf_counter() - t_setup
    edge_doubles = 5 * V * Ht * n * n  # WX_EULER3D_EDGE_FIELDS
    comm = None
    exchange_report = {"backend": "none needed: one rank owns every tile, the halos alias the packed edge buffers"}

    def all_ranks(flag: bool) -> bool:
        """True when `flag` holds on every rank (one all-reduce of the process group; the decision is the same everywhere)."""
        if world == 1 or not dist.is_initialized():
            return flag
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)   # (gloo: host tensors)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def torch_exchange():
        # torch.distributed.all_to_all_single on the gloo group, staged through host copies (exchange.py): the independent
        # second route of the self-check and the fallback - slow, correct, pinned by tests/test_exchange_gloo.py
        return PanelExchange(edge_doubles, dev, rank=rank, world_size=world, tiles_per_side=k, backend="torch")

    ex = None
    dog = None
    if args.exchange == "rccl" and (world > 1 or args.loopback):
        # a watchdog over the set-up and the first evaluation of the library's exchange: a rank that never reaches the
        # collective communicator set-up, or a send without its receive, becomes a message on stderr and exit code 4
        # instead of a silent timeout
        import threading

        def hung():
            sys.stderr.write(f"bench.py rank {rank}: the RCCL exchange (communicator set-up / first grouped ncclSend + ncclRecv) "
                             "did not finish in 300 s; rerun with --exchange torch (host-staged gloo)\n")
            sys.stderr.flush()
            os._exit(4)

        dog = threading.Timer(300.0, hung)
        dog.daemon = True
        dog.start()
        # the library's own exchange (wx_comm_*, wx_exchange_*).  Its first run on several GPUs is the driver's: if the
        # communicator or the buffers cannot be set up on some rank, every rank falls back to the host-staged gloo route and
        # the line says so, instead of the whole scaling run being lost
        from wxfactory_amd.exchange import RcclComm

        why = None
        # RCCL prints its version banner on STDOUT when NCCL_DEBUG is set (at the first communicator's creation), whatever
        # NCCL_DEBUG_FILE says: fd 1 points at fd 2 for the duration of the set-up - this program's stdout is the ONE line
        sys.stdout.flush()
        keep_fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = RcclComm(rank, world, device=dev)   # (the unique id travels through the gloo group; one rank needs none)
            ex = PanelExchange(edge_doubles, dev, rank=rank, world_size=world, tiles_per_side=k, loopback=args.loopback,
                               backend="rccl", comm=comm)
        except Exception as e:   # noqa: BLE001 - reported in the line
            why = f"{type(e).__name__}: {e}"
        finally:
            sys.stdout.flush()
            os.dup2(keep_fd1, 1)
            os.close(keep_fd1)
        if all_ranks(why is None):
            exchange_report = {"backend": "rccl behind the C ABI (wx_exchange_*: grouped ncclSend / ncclRecv, event fork / join)"}
        else:
            ex, comm = None, None
            exchange_report = {"backend": "gloo all_to_all_single through host copies", "fell_back_from": "rccl behind the C ABI",
                               "reason": why or "set-up failed on another rank"}
    elif world > 1:
        exchange_report = {"backend": "gloo all_to_all_single through host copies (--exchange torch)"}
    elif args.loopback:
        raise SystemExit("--loopback rehearses the library's exchange on one GPU: it needs --exchange rccl")
    if ex is None:
        ex = torch_exchange()
    rhs = RhsEuler3D(plans, ex, overlap=not args.no_overlap)

    if getattr(ex, "_native", None) is not None:
        # self-check before anything is timed: the same state through the library's exchange and through an independent
        # route must give the same R bit for bit on every rank.  Several ranks: gloo's all_to_all_single through host copies
        # (the route tests/test_exchange_gloo.py pins against the reference's halos).  One rank in loopback mode: the
        # aliasing exchange (no message moves).
        probe = torch.stack([qs[t] for t in mine]) if mine else qs
        got = rhs(probe)
        torch.cuda.synchronize()
        if world > 1:
            rhs_t = RhsEuler3D(plans, torch_exchange(), overlap=not args.no_overlap)
            route = "gloo all_to_all_single through host copies"
        else:
            rhs_t = RhsEuler3D(plans, PanelExchange(edge_doubles, dev, rank=0, world_size=1, tiles_per_side=k))
            route = "aliasing exchange of one rank"
        want = rhs_t(probe)
        torch.cuda.synchronize()
        same = all_ranks(bool(torch.equal(got, want)) if mine else True)
        exchange_report["selfcheck"] = (f"R(Q) bit-identical to the {route} on every rank" if same else
                                        f"MISMATCH against the {route}: timed on that route instead")
        if not same:
            exchange_report["backend"] = route
            exchange_report["fell_back_from"] = "rccl behind the C ABI"
            rhs, ex = rhs_t, rhs_t.ex
        del got, want, probe
    if dog is not None:
        dog.cancel()

    # live timing of the dominant kernel: HIP events on the launch stream around every K1 / K2 launch while EV.on
    # (wxfactory_amd.rhs_euler3d.LAUNCH_EVENTS: the plans' own opt-in instrumentation - nothing is patched)
    from wxfactory_amd.rhs_euler3d import LAUNCH_EVENTS as EV

    EV.clear()
    # the state of a rank: its tiles stacked in one tensor (what a time loop holds), so that small tiles can share launches
    state = torch.stack([qs[t] for t in mine]) if mine else qs

    def barrier():
        if world > 1:
            dist.barrier()

    # the exchange behind the C ABI: a rank's whole evaluation is ONE wx_euler3d_rhs_overlapped call (whole panels; the
    # tiles of the 24-tile layout share launches through the batch instead), which the Python-side event pairs above never
    # see - the call stamps the reference's nine-slot timing row itself (wx_exchange_set_timer), one timer per timed step
    import ctypes

    native_timers = []
    one_call = (getattr(ex, "_native", None) is not None and bool(mine) and not rhs._small_tiles() and not args.no_overlap)
    if one_call:
        lib = _lib.load()
        for _ in range(args.steps):
            h = ctypes.c_void_p()
            _lib.check(lib.wx_phase_timer_create(ctypes.byref(h)), "wx_phase_timer_create")
            native_timers.append(h)

    if args.compute_priority == "high":   # everything below is enqueued on a high-priority stream (an A/B of the overlap)
        hp = torch.cuda.Stream(device=dev, priority=-1)
        hp.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(hp)

    def timed_pass(events_every):
        """K steps between barriers; events_every: 0 none, K > 0 around the launches of every K-th step."""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = None
        for i in range(args.steps):
            EV.on = bool(events_every) and i % events_every == 0
            if one_call:
                lib.wx_exchange_set_timer(ex._native, native_timers[i] if EV.on else None)
            res = rhs(state)
        torch.cuda.synchronize()
        barrier()
        took = time.perf_counter() - t0
        EV.on = False
        if one_call:
            lib.wx_exchange_set_timer(ex._native, None)
        return res, took

    # spin-up, outside every count: the device leaves its idle clocks some hundred milliseconds after work arrives - a warm-up of
    # W = 10 steps (67 ms) ended inside that ramp on one box of round 6 and the FIRST timed pass read 8.5 ms per step where the
    # second read 6.7 (profiles/r06_v1_bench.json.log against r06_v1_bench_profiled.json.log) - so evaluations run for half a
    # second first, then the W warm-up steps of the contract, then EXACTLY K timed steps
    # (an evaluation is a COLLECTIVE at N > 1: every rank must run the same number of them - the count comes from rank-wide numbers)
    out = None
    torch.cuda.synchronize()
    barrier()
    t_spin = time.perf_counter()
    for _ in range(5):
        out = rhs(state)
    torch.cuda.synchronize()
    per_eval = torch.tensor([(time.perf_counter() - t_spin) / 5.0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(per_eval, op=dist.ReduceOp.MAX)
    more = int(min(5000, max(0, math.ceil(args.spinup / max(float(per_eval.item()), 1e-6)) - 5)))
    for i in range(more):
        out = rhs(state)
        if i % 8 == 7:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    spin = 5 + more
    for _ in range(args.warmup):
        out = rhs(state)
    # the headline: EXACTLY K steps between barriers, nothing but the evaluation inside
    out, dt = timed_pass(args.event_every)
    dt_events = None
    if args.event_every == 0:
        # the kernels' own times: the same K steps again with HIP events around every launch (outside the headline)
        _, dt_events = timed_pass(1)
    ev, ev1 = EV.rhs, EV.pack
    stamped = args.event_every if args.event_every > 0 else 1
    native_rows, native_since = [], []
    if one_call:
        for i, h in enumerate(native_timers):
            if i % stamped == 0:
                row = (ctypes.c_double * 9)()
                _lib.check(lib.wx_phase_timer_elapsed(h, row), "wx_phase_timer_elapsed")
                native_rows.append(list(row))
                _lib.check(lib.wx_phase_timer_since_start(h, row), "wx_phase_timer_since_start")
                native_since.append(list(row))
            lib.wx_phase_timer_destroy(h)
    if mine:
        chk = float(out.abs().amax(dim=(1, 2, 3, 4, 5)).sum())
        if not (chk == chk and chk < float("inf")):
            raise SystemExit("non-finite RHS in the benchmark")

    tmax = torch.tensor([dt], dtype=torch.float64)   # (gloo: host tensors)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    dt_events_max = None
    if dt_events is not None:
        tmax2 = torch.tensor([dt_events], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tmax2, op=dist.ReduceOp.MAX)
        dt_events_max = float(tmax2.item())
    ranks_seen = dist.get_world_size() if world > 1 else 1
    if ranks_seen != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} ranks")
    # What makes an N-rank line checkable against the N = 1 line: the whole-sphere R of the last timed step, per variable
    # sum, sum of magnitudes and largest magnitude (every rank's tiles, all-reduced).  The sphere, its state and its
    # metric do not depend on the decomposition (the state is a cut of the panel's, synthetic.euler3d_state), and R does
    # not to 1e-13 (tests: test_result_does_not_depend_on_the_decomposition), so `sum` must agree between any two lines to
    # 1e-13 x abs_sum, max_abs to 1e-13 relative.
    local = torch.zeros((3, 5), dtype=torch.float64, device=dev)
    if mine:
        local[0] = out.sum(dim=(0, 2, 3, 4, 5))
        local[1] = out.abs().sum(dim=(0, 2, 3, 4, 5))
        local[2] = out.abs().amax(dim=(0, 2, 3, 4, 5))
    local = local.cpu()
    if world > 1:
        sums = local[:2].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        mx = local[2].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        local = torch.cat((sums, mx[None]))
    checksum = {"of": "whole-sphere R(Q) of the last timed step, per variable (rho, rho u1, rho u2, rho w, rho theta)",
                "sum": local[0].tolist(), "abs_sum": local[1].tolist(), "max_abs": local[2].tolist(),
                "agreement": "between decompositions: |sum - sum'| <= 1e-13 abs_sum, max_abs to 1e-13 relative"}

    # per-rank phase times (outside the timed region): the reference's nine RHS timestamps (rhs/rhs.py:88-118) on
    # HIP events of the launch stream, five more evaluations
    rhs.timed = True   # (the timed evaluation launches tile by tile: each phase of each tile gets its own stamps)
    rhs.clear_timings()
    for _ in range(5):
        rhs(state)
    torch.cuda.synchronize()
    rhs.retrieve_last_times()
    rhs.timed = False
    tm = rhs.timings[1:] or rhs.timings
    mean = lambda i: round(sum(t[i] for t in tm) / len(tm) * 1e3, 4) if tm else None  # noqa: E731
    mine_phase = {"rank": rank, "tiles": len(mine), "pack_ms": mean(0), "exchange_start_ms": mean(1), "interior_ms": mean(2),
                  "exchange_ms": mean(4), "boundary_ms": mean(5), "total_ms": mean(8)}
    if native_since:
        # where the two streams' phases of the overlapped evaluation lie against each other (stamps of the timed steps, ms from
        # the start of the evaluation): INTERIOR on the second stream [2, 3], the exchange complete on the compute stream at 5
        avg = lambda i: sum(r[i] for r in native_since) / len(native_since) * 1e3  # noqa: E731
        inside = sum(1 for r in native_since if 0.0 <= r[5] <= r[3])
        mine_phase["overlap"] = {"interior_start_ms": round(avg(2), 4), "interior_end_ms": round(avg(3), 4),
                                 "exchange_done_ms": round(avg(5), 4), "evaluations": len(native_since),
                                 "exchange_done_before_interior_end": inside == len(native_since),
                                 "evaluations_with_exchange_inside_interior": inside,
                                 "second_stream_priority": getattr(ex, "side_priority", None)}
    per_rank = [mine_phase]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_phase)
    barrier()

    pts_panel = V * H * H * n**3
    evals_per_s = args.steps / dt
    dof_per_s = 5 * pts_panel * 6 * evals_per_s

    # dominant-kernel roofline (rank 0's launches)
    roof = None
    if ev or native_rows:
        by_region, tiles_in_launch = {}, {}
        for a, b, region, ntl in ev:
            by_region.setdefault(region, []).append(a.elapsed_time(b) * 1e-3)
            tiles_in_launch[region] = ntl
        for row in native_rows:   # seconds between the stamps 0 1 2 3 5 8: pack, start, INTERIOR, -, join, -, -, BOUNDARY
            by_region.setdefault(_lib.WX_REGION_INTERIOR, []).append(row[2] / len(mine))
            by_region.setdefault(_lib.WX_REGION_BOUNDARY, []).append(row[7] / len(mine))
            tiles_in_launch[_lib.WX_REGION_INTERIOR] = tiles_in_launch[_lib.WX_REGION_BOUNDARY] = 1
        w = Ht - 2 if Ht > 2 else 0
        frac_of_panel = {_lib.WX_REGION_ALL: 1.0, _lib.WX_REGION_INTERIOR: (w * w) / (Ht * Ht),
                         _lib.WX_REGION_BOUNDARY: 1.0 - (w * w) / (Ht * Ht)}
        # the dominant launch shape: ALL at N=1, INTERIOR when the exchange is overlapped
        region = max(by_region, key=lambda r: sum(by_region[r]))
        tk = sum(by_region[region]) / len(by_region[region])
        # compulsory bytes of THIS launch: SURVEY 8d's 384 B/point, minus the 72 B/point of the nine rotation
        # Christoffel fields when the plan found them identically zero (non-rotating planet) and skips them
        bpp = next(iter(plans.values())).bytes_per_point if plans else ALGO_BYTES_PER_POINT
        bytes_launch = bpp * (pts_panel / (k * k)) * frac_of_panel[region] * tiles_in_launch[region]
        achieved = bytes_launch / tk / 1e9
        traffic, traffic_src = pmc_traffic(region, n, Ht, V, bpp)
        roof = {"bound": "hbm", "kernel": "euler_rhs_kernel<8,double>", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src, "launch_ms": round(tk * 1e3, 4),
                "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_point": bpp,
                "survey_bytes_per_point": ALGO_BYTES_PER_POINT,
                "region": {0: "all", 1: "interior", 2: "boundary"}[region], "tiles_per_launch": tiles_in_launch[region]}
        if ev1:
            roof["extrap_kernel_launch_ms"] = round(sum(a.elapsed_time(b) for a, b in ev1) / len(ev1), 4)
        if native_rows:
            roof["extrap_kernel_launch_ms"] = round(sum(r[0] for r in native_rows) / len(native_rows) / len(mine) * 1e3, 4)
            roof["timing_source"] = ("wx_phase_timer stamps inside wx_euler3d_rhs_overlapped (one host call per evaluation): "
                                     "the launches of a phase follow each other on the compute stream, launch time = phase / tiles")
        # the whole sweep (extrapolation kernel + exchange + fused kernel, every local tile): compulsory bytes of one
        # R(Q) of this rank's tiles (each static field and Q read once, R written once) over the step time
        sweep_bytes = bpp * (pts_panel / (k * k)) * len(mine)
        sweep = sweep_bytes / (dt / args.steps) / 1e9
        roof["sweep"] = {"algorithmic_bytes": sweep_bytes, "achieved": round(sweep, 1), "frac": round(sweep / HBM_PEAK_GBS, 4),
                         "note": "rank 0's tiles; the interface buffer's round trip through HBM and the extrapolation "
                                 "kernel's second read of Q are not compulsory bytes"}
        roof["sweep_frac"] = roof["sweep"]["frac"]
        # the north star's target (>= 50 % of the HBM roofline on the rhs_euler sweep), judged on the bytes this plan is
        # COMPELLED to move - nothing it skips is counted
        roof["sweep"]["target"] = {"source": "BASELINE.json north_star: >= 50 % of MI355X HBM roofline on the rhs_euler sweep",
                                   "bytes_per_point": bpp, "frac": roof["sweep"]["frac"],
                                   "dof_updates_per_s_this_rank": 5 * (pts_panel / (k * k)) * len(mine) / (dt / args.steps),
                                   "met": bool(roof["sweep"]["frac"] >= 0.5)}
        roof["matrix_cores"] = mfma_block(plans, n, tk, (pts_panel / (k * k)) / n**3 * frac_of_panel[region] * tiles_in_launch[region])
        roof["ceiling"] = copy_ceiling(dev)
        if roof["ceiling"] and traffic:
            on_traffic = traffic / tk / 1e9
            ceil = roof["ceiling"]
            roof["on_measured_traffic"] = {"GBps": round(on_traffic, 1), "frac_of_copy_rate": round(on_traffic / ceil["achieved"], 4),
                                           "traffic_over_algorithmic": round(traffic / bytes_launch, 4)}
            fb, wb = traffic_src.get("fetch_bytes"), traffic_src.get("write_bytes")
            rd = ceil.get("read_only", {}).get("achieved")
            if fb and wb and rd:
                # the time this launch's own mix of reads and writes would take at the measured streaming rates: its reads at
                # the read-only rate, its writes at the rate the copy's writes are left with once its reads are priced so
                n_copy = ceil["bytes_read_plus_written"] / 2.0
                t_copy_writes = ceil["launch_ms"] * 1e-3 - n_copy / (rd * 1e9)
                if t_copy_writes > 0:
                    wr = n_copy / t_copy_writes / 1e9
                    floor_s = tiles_in_launch[region] * (fb / (rd * 1e9) + wb / (wr * 1e9))
                    roof["on_measured_traffic"].update({"read_rate_GBps": rd, "implied_write_rate_GBps": round(wr, 1),
                                                        "streaming_floor_ms": round(floor_s * 1e3, 4),
                                                        "frac_of_streaming_floor": round(floor_s / tk, 4)})

    if rank == 0:
        line = {
            "metric": "DOF-updates/s (whole-sphere 3-D Euler RHS evals, cubed sphere p=7, 60x60 elem/panel)",
            "value": dof_per_s, "unit": "DOF-updates/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "spinup_evaluations_before_warmup": spin, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            **({"ms_per_step_with_launch_events": dt_events_max / args.steps * 1e3,
                "launch_events": "the kernel times of `roofline` come from a second pass of the same steps with HIP events around "
                                 "every launch; `value` and ms_per_step from the first, uninstrumented one"}
               if dt_events_max is not None else {}),
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rhs_evals_per_s": evals_per_s, "ranks_seen": ranks_seen, "per_rank": per_rank, "checksum": checksum,
            "rccl_version": _lib.load().wx_comm_rccl_version(),
            "hip_runtime_version": _lib.load().wx_hip_runtime_version(),   # what the process BOUND (inside torch: the wheel's)
            "process_group": "gloo (host side only: id bootstrap, barriers, max over ranks); no NCCL process group" if world > 1
                             else "none",
            **({"rehearsal": "--one-device: every rank on GPU 0, halos through gloo and host copies - program flow only, NOT a "
                             "measurement"} if args.one_device else {}),
            "config": {"workload": f"E7: 3-D Euler RHS, n={n} (p={n-1}), H={H}x{H} elem/panel, V={V}, 6 panels "
                                   f"({5*pts_panel*6} DOF), halo exchange included",
                       "n": n, "H": H, "V": V, "tiles": topo.ntiles, "tile_H": Ht, "tiles_per_gpu": len(mine),
                       "parallelism": f"tile-dd{min(world, topo.ntiles)}",
                       "overlap": not args.no_overlap,
                       "exchange": exchange_report,
                       "metric": "geometry3d: equiangular cubed sphere, DCMIP 3-1 planet (R/125), ztop 10 km"
                                 if args.metric == "true" else "seeded synthetic fields (SURVEY 8d)",
                       "metric_setup_s": round(t_setup, 1)},
            "roofline": roof,
        }
        if args.gpus == 1 and not args.no_extras:
            line["extra"] = extras(dev, args.seed)
            line["extra"]["euler_callers"] = caller_extras(rhs, qs)
            line["extra"]["euler_ini_sizes"] = ini_size_extras(dev, args.seed)
            if args.metric == "true" and not args.loopback:
                line["extra"]["euler_column_metric"] = column_metric_extras(plans, mine, state, out, edge_doubles, dev, k,
                                                                            dt / args.steps, args.steps)
            del rhs, qs, plans, out
            torch.cuda.empty_cache()
            line["extra"]["euler_e7_v1"] = e7_v1_extras(dev, args.seed)
            line["extra"]["epi2_kiops_e7"] = epi2_kiops_e7_extras(dev, args.seed)
            line["extra"]["rhs_benchmark_matrix"] = rhs_benchmark_matrix(dev, args.seed)
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, V, args.seed, H)
        print(json.dumps(line), flush=True)
    # teardown in dependency order: nothing in flight, then the library's exchange and communicator, then the process group
    torch.cuda.synchronize()
    barrier()
    try:
        ex.close()
    except Exception:   # noqa: BLE001
        pass
    if comm is not None:
        comm.close()   # (closes every exchange made on it first: twins, value / tangent sets, the stage pipeline's)
    if dist.is_initialized():
        dist.destroy_process_group()


