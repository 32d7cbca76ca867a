"""The library's communicator as the ONE communicator of a rank (reference: process_topology.py:259-261 for the halo
exchange, solvers/global_operations.py:14-36 / kiops.py:165-200 / pmex.py:150-173 / fgmres.py:41 for the callers' small
reductions, simulation.py:399-408 for the NaN flag): wx_comm_allreduce behind RcclComm.allreduce and reduce.py, the
refusals that replace known process-killing calls, and BASELINE config 5 over the several-rank code path - a whole KIOPS pass
(matvec + halo exchange + both reductions per Krylov vector) replayed from one HIP graph with the reference's statistics.

No torch.distributed process group exists in this process: the 128-byte id of a one-rank communicator needs no transport, and
a several-rank run hands it round over gloo or a TCPStore (tests/test_exchange_gloo.py::test_id_bootstrap_over_a_store).
Why none: profiles/r05_process_group_abort.md."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
AX = (0, 2, 3, 4, 5)


@pytest.fixture(scope="module")
def comm():
    from wxfactory_amd.exchange import RcclComm

    c = RcclComm(0, 1, device=DEV)
    c.always = True   # issue the reductions on one rank too (reduce.allreduce skips them otherwise)
    yield c
    torch.cuda.synchronize()
    c.close()


def test_no_process_group_in_this_process():
    import torch.distributed as dist

    assert not dist.is_initialized()


def test_versions_the_process_bound(comm):
    from wxfactory_amd import _lib

    lib = _lib.load()
    rt, drv = lib.wx_hip_runtime_version(), lib.wx_hip_driver_version()
    assert rt >= 60000000 and drv >= 60000000, (rt, drv)
    assert comm.hip_runtime_version == rt and comm.version >= 20000
    # inside a torch process the library runs on the wheel's runtime, whatever hipcc compiled it
    hip = torch.version.hip.split(".")
    assert rt // 10_000_000 == int(hip[0]) and (rt // 100_000) % 100 == int(hip[1]), (rt, torch.version.hip)


def test_allreduce_on_the_library_communicator(comm):
    from wxfactory_amd import _lib, reduce

    assert reduce.is_comm(comm) and reduce.world_size(comm) == 1 and reduce.capturable(comm)
    x = torch.arange(1.0, 9.0, dtype=torch.float64, device=DEV)
    for op in ("sum", "max", "min"):
        y = x.clone()
        assert reduce.allreduce(y, comm, op) is y
        torch.cuda.synchronize()
        assert torch.equal(y, x), op          # one rank: the reduction of one contribution
    flag = torch.tensor([3], dtype=torch.int32, device=DEV)     # (the NaN flag: reduced through a float64 copy)
    assert int(reduce.allreduce(flag, comm, "max").item()) == 3 and flag.dtype == torch.int32
    length = torch.tensor([2**40 + 1], dtype=torch.int64, device=DEV)
    assert int(reduce.allreduce(length, comm, "sum").item()) == 2**40 + 1
    z = torch.complex(x, -x)
    assert torch.equal(reduce.allreduce(z.clone(), comm, "sum"), z)
    # argument errors come back as a status, not a crash
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.wx_comm_allreduce(None, x.data_ptr(), 8, 0, st) != _lib.WX_OK
    assert lib.wx_comm_allreduce(comm._h, x.data_ptr(), 8, 7, st) != _lib.WX_OK and b"unknown reduction" in lib.wx_last_error()
    assert lib.wx_comm_allreduce(comm._h, None, 0, 0, st) == _lib.WX_OK   # nothing to reduce
    # as a node of a HIP graph (the capture's origin stream), replayed on fresh data
    buf = torch.zeros(4, dtype=torch.float64, device=DEV)
    out = torch.zeros(4, dtype=torch.float64, device=DEV)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            buf.mul_(2.0)
            comm.allreduce(buf, "sum")
            out.copy_(buf)
    torch.cuda.current_stream().wait_stream(side)
    comm.register_graph(g)
    for k in (1.0, 3.0):
        buf.fill_(k)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, torch.full_like(out, 2.0 * k))
    g.reset()


def test_comm_destroy_refuses_while_exchanges_live():
    from wxfactory_amd import _lib
    from wxfactory_amd.exchange import PanelExchange, RcclComm

    lib = _lib.load()
    c = RcclComm(0, 1, device=DEV)
    assert lib.wx_comm_users(c._h) == 0
    a = PanelExchange(640, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=c)
    b = PanelExchange(1280, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=c)
    assert lib.wx_comm_users(c._h) == 2
    assert lib.wx_comm_destroy(c._h) != _lib.WX_OK and b"alive" in lib.wx_last_error()   # refused: nothing was freed
    a.send_buf.fill_(1.0)
    a.start(on_compute=True)            # ... and the communicator still works
    a.wait()
    torch.cuda.synchronize()
    assert float(a.recv_buf.min()) == 1.0
    b.close()
    assert lib.wx_comm_users(c._h) == 1
    c.close()                           # closes `a` first, then destroys the communicator
    assert a._native is None and not c._h


def test_forked_exchange_under_capture_is_refused_where_it_would_kill_the_process(comm):
    """wx_exchange_start(ex, compute, comm_stream != compute) while `compute` is being captured: on HIP runtimes before 7.2
    (the 7.0.2 inside torch 2.10 wheels) hipStreamEndCapture would overflow the stack (profiles/r04_capture_crash.md).  The
    library says so with a status; the capture survives and ends cleanly."""
    from wxfactory_amd import _lib
    from wxfactory_amd.exchange import PanelExchange

    lib = _lib.load()
    ex = PanelExchange(640, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm)
    ex.send_buf.fill_(2.0)
    old_runtime = lib.wx_hip_runtime_version() < 70200000
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    status = None
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            ex.send_buf.mul_(1.0)
            if old_runtime:
                status = lib.wx_exchange_start(ex._native, side.cuda_stream, ex.comm_stream.cuda_stream)
                msg = lib.wx_last_error().decode()
            # the form that records everywhere: the group on the capture's origin stream
            ex.start(on_compute=True)
    torch.cuda.current_stream().wait_stream(side)
    if old_runtime:
        assert status != _lib.WX_OK and "EndCapture" in msg and "origin" in msg, (status, msg)
    g.replay()
    torch.cuda.synchronize()
    assert float(ex.recv_buf.min()) == 2.0 and float(ex.recv_buf.max()) == 2.0
    g.reset()
    ex.close()
    if not old_runtime:
        pytest.skip("this process bound a HIP runtime >= 7.2: the forked form records there (tests/test_exchange_rccl_gpu.py::"
                    "test_plain_c_capture_probe)")


def test_overlapped_calls_check_the_message_size_and_clean_up(comm):
    """wx_sw_rhs_overlapped / wx_euler3d_rhs_overlapped with an exchange whose messages have another size than the plans
    pack: refused before anything is written (the pack kernels would write past their send slots)."""
    from tests.gpu_util import make_plan, to_dev
    from tests.test_sw_gpu import _plan
    from tests.util import golden_sw
    from wxfactory_amd import _lib
    from wxfactory_amd.exchange import PanelExchange

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    gs = golden_sw("sw_c5_n4_h3")
    sw = {p: _plan(gs, p) for p in range(6)}
    ge = golden("euler3d_c31p_n3_h4_v2")
    eu = {p: make_plan(ge, p) for p in range(6)}
    for plans, entry, g in ((sw, lib.wx_sw_rhs_overlapped, gs), (eu, lib.wx_euler3d_rhs_overlapped, ge)):
        ex = PanelExchange(plans[0].edge_count + 8, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm)
        q = [to_dev(g.q(p)) for p in range(6)]
        r = [torch.zeros_like(x) for x in q]
        handles = (ctypes.c_void_p * 6)(*[plans[p]._h for p in range(6)])
        qp = (ctypes.c_void_p * 6)(*[x.data_ptr() for x in q])
        rp = (ctypes.c_void_p * 6)(*[x.data_ptr() for x in r])
        status = entry(handles, 6, ex._native, qp, rp, st, ex.comm_stream.cuda_stream)
        assert status != _lib.WX_OK and b"doubles per edge" in lib.wx_last_error(), lib.wx_last_error()
        torch.cuda.synchronize()
        assert all(float(x.abs().max()) == 0.0 for x in r) and float(ex.send_buf.abs().max()) == 0.0
        ex.close()
    assert lib.wx_sw_plan_dtype(sw[0]._h) == _lib.WX_F64


def test_config5_kiops_passes_with_exchange_and_reductions_replay_from_one_graph(comm):
    """BASELINE config 5 as a rank of a several-GPU run executes it, on one GPU: config/dcmip21.ini's EPI2 + KIOPS on the
    Schaer-mountain state with (a) every edge message travelling through the library's RCCL exchange (loopback), (b) the
    two reductions of every Krylov vector issued as wx_comm_allreduce on the same communicator (solvers/kiops.py:176-200:
    the several-rank code path, `_force_split`), and (c) whole Krylov passes - matvec, exchange, both reductions per
    vector - captured into ONE HIP graph each and replayed.  The adaptive controller must still take the reference's
    decisions step by step, and the states must match the reference's and the plain one-rank run's."""
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
    from wxfactory_amd.integrators import Epi
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    g = np.load(os.path.join(GOLDEN, "config5_c21_n4_h2_v3.npz"))
    n, H, V, case = (int(g[f"meta/{k}"]) for k in ("n", "H", "V", "case_number"))
    topo = topography_for_case(case, planet_for_case(case)[0])
    plans = {}
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, float(g["meta/ztop"]), case, topo=topo)
        plans[p] = Euler3DPlan(n, H, V, case, p, dfr_ops(n), metric3d_torch(t, DEV))
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm)
    rhs = RhsEuler3D(plans, ex, overlap=True)
    assert rhs.reduce_group is comm and rhs._small_tiles()
    tol = float(g["meta/tolerance"])
    stack = lambda key: np.stack([g[f"p{p}/{key}"] for p in range(6)])  # noqa: E731
    Q0 = torch.from_numpy(stack("Q")).to(DEV)
    dt, nsteps = float(g["meta/dt"]), int(g["meta/nsteps"])
    ref_stats = g["meta/kiops_stats"]
    calls = {"n": 0}
    inner = comm.allreduce

    def counting(t, op="sum"):
        calls["n"] += 1
        return inner(t, op)

    def new_epi(graphs):
        e = Epi(2, rhs, tol=tol)
        assert e.group is comm
        e._force_split, e.graph_passes = True, graphs
        return e

    def stats_of(e):
        return [int(e.solver_info[k]) for k in ("substeps", "rejected", "iterations", "exps", "krylov_size")]

    comm.allreduce = counting
    try:
        # (1) eager, step by step against the reference's statistics and states and against the plain one-rank run
        epi, plain, Q = new_epi(False), Epi(2, RhsEuler3D(plans), tol=tol), Q0
        for i in range(nsteps):
            Qn = epi.step(Q, dt)
            assert stats_of(epi) == [int(ref_stats[i][j]) for j in (0, 1, 2, 3, 5)], (i, epi.solver_info, ref_stats[i])
            ref_u = stack(f"Q{i + 1}_unfiltered")
            upd = np.abs(ref_u - Q.cpu().numpy()).max(axis=AX)
            err = np.abs(Qn.cpu().numpy() - ref_u).max(axis=AX)
            assert (err <= 1e-6 * upd + 1e-13 * np.abs(ref_u).max(axis=AX)).all(), (i, err / upd)
            Qp = plain.step(Q, dt)
            assert stats_of(plain) == stats_of(epi)
            assert float((Qn - Qp).abs().max()) <= 1e-9 * float((Qp - Q).abs().max()), i
            Q = torch.from_numpy(stack(f"Q{i + 1}")).to(DEV)   # the reference's filtered state: the start of its next step
        assert calls["n"] >= 2 * int(ref_stats[0][2])   # two reductions per Krylov vector went through the communicator
        # (2) the same first step three times with whole passes as HIP graphs: the first occurrence of a pass (j0, m) runs
        # eagerly, the second is captured - matvec, exchange, reductions - and replayed, the third replays
        gepi, outs = new_epi(True), []
        for _ in range(3):
            gepi.krylov_size = 1
            before = calls["n"]
            outs.append(gepi.step(Q0, dt))
            assert stats_of(gepi) == [int(ref_stats[0][j]) for j in (0, 1, 2, 3, 5)], (gepi.solver_info, ref_stats[0])
            called = calls["n"] - before
        ws = gepi._ws
        assert ws.captures >= 1 and ws.replays >= ws.captures + 1, (ws.captures, ws.replays)
        assert called < 2 * int(ref_stats[0][2])   # the replayed passes never came back to Python for their reductions
        ref_u = stack("Q1_unfiltered")
        upd = np.abs(ref_u - Q0.cpu().numpy()).max(axis=AX)
        for o in outs:
            err = np.abs(o.cpu().numpy() - ref_u).max(axis=AX)
            assert (err <= 1e-6 * upd + 1e-13 * np.abs(ref_u).max(axis=AX)).all(), err / upd
        assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])   # eager, captured + replayed, replayed
    finally:
        comm.allreduce = inner
    ws.release()
    del gepi, epi, rhs
    torch.cuda.synchronize()
    ex.close()


@pytest.mark.parametrize("p,taus", [(1, [1.0]), (3, [0.4, 1.0])])
def test_pmex_vectors_split_over_ranks_reduce_in_stream_order(comm, p, taus):
    """solvers/pmex.py:150-173, 194-218 with the vectors split over ranks: the (j+1) x 2 block of products and the vector's own
    norm are completed by wx_comm_allreduce inside wx_pmex_vector_split, in stream order - the host reads the Hessenberg
    columns once per pass, as on one rank.  Same decisions and vectors as the unsplit build."""
    from wxfactory_amd.solvers import pmex

    n = 50_000
    gen = torch.Generator(device=DEV).manual_seed(23 + p)
    lam = -(0.2 + 2.5 * torch.rand(n, generator=gen, device=DEV, dtype=torch.float64))
    u = torch.randn((p + 1, n), generator=gen, device=DEV, dtype=torch.float64)
    A = lambda v: lam * v  # noqa: E731
    args = dict(tol=1e-10, m_init=12, mmin=10, mmax=40)
    w_one, st_one = pmex(taus, A, u, **args)
    calls = {"n": 0}
    inner = comm.allreduce

    def counting(t, op="sum"):
        calls["n"] += 1
        return inner(t, op)

    comm.allreduce = counting
    try:
        w_split, st_split = pmex(taus, A, u, group=comm, _force_split=True, **args)
    finally:
        comm.allreduce = inner
    assert st_split[:4] == st_one[:4], (st_split, st_one)
    assert float((w_split - w_one).abs().max()) <= 1e-12 * float(w_one.abs().max())
    # the per-vector reductions never came back to Python (they are ncclAllReduce calls inside the C function): what Python
    # reduced is the start of each sub-step only (|u|, the first vector's norm)
    assert calls["n"] <= 2 + 2 * st_split[0] + 2 * st_split[1], (calls, st_split)


@pytest.mark.parametrize("p", [1, 3])
def test_pmex_on_a_rank_that_owns_nothing_still_takes_part(comm, p):
    """ranks beyond the tile count (6, 7 of an 8-GPU node) hold the p augmented components of every Krylov vector
    and NO nodal part: A(V[:0]) and the flipped u are empty tensors, whose address is null.  wx_pmex_vector_split must accept
    them (it refused, and the other ranks - already inside ncclAllReduce - would have hung), issue both reductions and build
    the augmented part: the whole solver on an empty nodal part, device pass, against the host pass on the same input."""
    from wxfactory_amd.solvers import pmex

    u = torch.zeros((p + 1, 0), device=DEV, dtype=torch.float64)
    A = lambda v: v  # noqa: E731  (never sees a component)
    args = dict(tol=1e-10, m_init=6, mmin=4, mmax=20)
    calls = {"n": 0}
    inner = comm.allreduce

    def counting(t, op="sum"):
        calls["n"] += 1
        return inner(t, op)

    comm.allreduce = counting
    try:
        w_dev, st_dev = pmex([1.0], A, u, group=comm, _force_split=True, **args)
    finally:
        comm.allreduce = inner
    assert calls["n"] > 0    # the collective decisions still went through the communicator
    os.environ["WXHIP_PMEX_DEVICE"] = "0"
    try:
        w_host, st_host = pmex([1.0], A, u, group=comm, _force_split=True, **args)
    finally:
        del os.environ["WXHIP_PMEX_DEVICE"]
    assert w_dev.shape == w_host.shape and w_dev.shape[-1] == 0
    assert st_dev[:4] == st_host[:4], (st_dev, st_host)


def test_bench_loopback_rehearsal_prints_one_line_and_checks_its_exchange(built_lib):
    """bench.py --loopback: the several-GPU path of the benchmark on one GPU - a one-rank communicator of the library's own, every
    edge message through grouped ncclSend / ncclRecv, INTERIOR beside the exchange, no process group of any kind - at a reduced
    size.  stdout must be exactly ONE JSON line (RCCL's banner goes to stderr), the exchange self-check must have passed against
    the aliasing route, and the checksum must equal the unsplit run's."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, os.path.join(root, "bench.py"), "--H", "6", "--V", "2", "--steps", "3", "--warmup", "1", "--no-extras",
              "--no-cpu-baseline"]
    lines = {}
    for name, extra in (("loopback", ["--loopback"]), ("plain", [])):
        r = subprocess.run(common + extra, capture_output=True, text=True, timeout=400, cwd=root)
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-3000:])
        out = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(out) == 1, (name, out[:5])
        lines[name] = json.loads(out[0])
    lb, pl = lines["loopback"], lines["plain"]
    assert lb["n_gpus"] == 1 and lb["ranks_seen"] == 1 and lb["process_group"] == "none"
    assert lb["config"]["exchange"]["backend"].startswith("rccl behind the C ABI"), lb["config"]["exchange"]
    assert "bit-identical" in lb["config"]["exchange"]["selfcheck"], lb["config"]["exchange"]
    assert lb["hip_runtime_version"] >= 60000000 and lb["rccl_version"] >= 20000
    for k in ("sum", "abs_sum", "max_abs"):
        a, b = np.asarray(lb["checksum"][k]), np.asarray(pl["checksum"][k])
        assert (np.abs(a - b) <= 1e-12 * np.maximum(np.asarray(pl["checksum"]["abs_sum"]), 1e-300)).all(), (k, a, b)
