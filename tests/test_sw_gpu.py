"""Parity of the shallow-water HIP path (C ABI) with the reference's golden vectors (rhs_sw.py)."""
import numpy as np
import pytest
import torch

from tests.util import SW_FIXTURES, golden_sw, make_sw_oracle, var_err, var_max  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = 1e-10
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _plan(g, p, dtype=torch.float64):
    from wxfactory_amd.rhs_sw import SwPlan

    m = {k: _dev(v) for k, v in g.sub(p, "metric").items() if k != "inv_sqrtG"}
    m.update({k: _dev(v) for k, v in g.sub(p, "topo").items()})
    m["boundary_sn"] = _dev(g[f"p{p}/geom/boundary_sn"])
    m["boundary_we"] = _dev(g[f"p{p}/geom/boundary_we"])
    return SwPlan(g.n, g.H, p, g.ops, m, dtype=dtype)


def _scale(g, p, cplx):
    o = make_sw_oracle(g, p)
    want = {}
    o.rhs(g.q(p, cplx), g.halo(p, cplx), want=want)
    return o.cancel_scale(want)


@pytest.mark.parametrize("name", SW_FIXTURES)
@pytest.mark.parametrize("cplx", [False, True])
def test_pack_and_rhs(name, cplx, built_lib):
    from wxfactory_amd.panels import NEIGHBOR, landing_edge

    g = golden_sw(name)
    dtype = torch.complex128 if cplx else torch.float64
    for p in range(6):
        plan = _plan(g, p, dtype)
        q = _dev(g.q(p, cplx))
        send = torch.zeros((4, plan.edge_count), dtype=dtype, device=DEV)
        plan.extrap_pack(q, [send[e].data_ptr() for e in range(4)])
        halo = [_dev(h) for h in g.halo(p, cplx)]
        out = torch.full_like(q, float("nan"))
        plan.rhs(q, [h.data_ptr() for h in halo], out)
        torch.cuda.synchronize()
        got = send.cpu().numpy().reshape(4, 3, g.H, g.n)
        for e in range(4):
            ref = g.halo(NEIGHBOR[p][e], cplx)[landing_edge(p, e)]
            assert np.abs(got[e] - ref).max() <= 1e-13 * np.abs(ref).max(), (p, e)
        R, ref = out.cpu().numpy(), g.r(p, cplx)
        scale = np.maximum(var_max(ref.real), _scale(g, p, cplx))
        assert (var_err(R.real, ref.real) <= TOL * scale).all(), (name, p, var_err(R.real, ref.real) / scale)
        if cplx:
            assert (var_err(R.imag, ref.imag) <= 1e-10 * var_max(ref.imag)).all()
        plan.close()


def test_whole_sphere_and_regions(built_lib):
    from wxfactory_amd import _lib
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw("sw_c5_n4_h3")  # topography
    plans = {p: _plan(g, p) for p in range(6)}
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1)
    rhs = RhsShallowWater(plans, ex)
    Rs = rhs({p: _dev(g.q(p)) for p in range(6)})
    torch.cuda.synchronize()
    for p in range(6):
        ref = g.r(p)
        assert (var_err(Rs[p].cpu().numpy(), ref) <= TOL * np.maximum(var_max(ref), _scale(g, p, False))).all()
    # INTERIOR + BOUNDARY == ALL
    p = 3
    q = _dev(g.q(p))
    a, b = torch.full_like(q, float("nan")), torch.full_like(q, float("nan"))
    plans[p].extrap_pack(q, ex.send_ptrs(p))
    plans[p].rhs(q, ex.halo_ptrs(p), a, _lib.WX_REGION_ALL)
    plans[p].rhs(q, None, b, _lib.WX_REGION_INTERIOR)
    plans[p].rhs(q, ex.halo_ptrs(p), b, _lib.WX_REGION_BOUNDARY)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("cplx", [False, True])
def test_batched_launch_equals_per_panel(cplx, built_lib):
    """wx_sw_batch_*: six panels in one launch per phase == six per-panel launches, bit for bit."""
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw("sw_c6_n5_h4")
    plans = {p: _plan(g, p) for p in range(6)}
    rhs = RhsShallowWater(plans, complex_arith="dual")
    Q = torch.stack([_dev(g.q(p, cplx)) for p in range(6)])
    Rb = rhs(Q)
    rhs.batched = False
    Rp = rhs(Q)
    torch.cuda.synchronize()
    assert torch.equal(Rb, Rp)
    for p in range(6):
        ref = g.r(p, cplx)
        scale = np.maximum(var_max(ref.real), _scale(g, p, cplx))
        assert (var_err(Rb[p].cpu().numpy().real, ref.real) <= TOL * scale).all()


def test_tvdrk3_fused_stage_equals_literal_sequence(built_lib):
    """SSP-RK3 on the shallow-water sphere (BASELINE configs 2-3): stage updates fused into the RHS
    store (batched, one launch per phase) == the reference's literal sequence of axpys."""
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.matvec import matvec_rat
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw("sw_c6_n5_h4")
    rhs = RhsShallowWater({p: _plan(g, p) for p in range(6)})
    assert rhs.supports_axpy and not rhs.supports_axpy2
    Q = torch.stack([_dev(g.q(p)) for p in range(6)])
    dt = 200.0
    a = Tvdrk3(rhs, fused=True).step(Q, dt)
    b = Tvdrk3(rhs, fused=False).step(Q, dt)
    rhs.batched = False
    c = Tvdrk3(rhs, fused=True).step(Q, dt)
    torch.cuda.synchronize()
    upd = (b - Q).abs().amax(dim=(0, 2, 3, 4))
    qmax = Q.abs().amax(dim=(0, 2, 3, 4))
    assert ((a - b).abs().amax(dim=(0, 2, 3, 4)) <= 1e-12 * upd + 1e-15 * qmax).all()  # a few ulp of the state
    assert torch.equal(a, c)
    # the FD Rosenbrock operator falls back to torch ops for shallow water (no second fused array)
    v = torch.randn_like(Q) * 1e-3 * Q.abs().amax(dim=(0, 2, 3, 4), keepdim=True)
    out = matvec_rat(v.flatten(), dt, Q, rhs(Q), rhs)
    assert torch.isfinite(out).all()


def test_hip_graph_replay_of_a_step(built_lib):
    """A whole SSP-RK3 step (6 launches + 0 torch kernels, batched fused stages) captured into one HIP graph."""
    from wxfactory_amd.graph import GraphedFunction
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw("sw_c2p_n8_h3")
    rhs = RhsShallowWater({p: _plan(g, p) for p in range(6)})
    stepper = Tvdrk3(rhs)
    Q = torch.stack([_dev(g.q(p)) for p in range(6)])
    eager = stepper.step(Q, 100.0)
    graphed = GraphedFunction(lambda q: stepper.step(q, 100.0), Q)
    out1 = graphed(Q).clone()
    out2 = graphed(eager).clone()          # new input through the static buffer
    torch.cuda.synchronize()
    assert torch.equal(out1, eager)
    assert torch.equal(out2, stepper.step(eager, 100.0))


def test_tiles_of_a_24_rank_decomposition(built_lib):
    """Shallow water on k x k tiles per panel (24 MPI ranks of the reference, mountain case): tile plans with
    on_panel_edge flags pack what the reference delivered to each neighbour and reproduce its R, and the whole
    24-tile sphere on one GPU (zero-copy exchange between tiles) does too."""
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan

    from oracle.sw2d import SW2DOracle

    g = golden_sw("sw_tiles24_c5_n4_h2")
    k = int(g["meta/k"])
    topo = CubeTopology(k)

    def _scale(g, t, cplx):  # noqa: F811 - tile-aware version of the module helper
        o = SW2DOracle(g.n, g.H, g.ops, g.sub(t, "metric"), g.sub(t, "topo"), g[f"p{t}/geom/boundary_sn"],
                       g[f"p{t}/geom/boundary_we"], panel=topo.locate(t)[0])
        want = {}
        o.rhs(g.q(t, cplx), g.halo(t, cplx), want=want)
        return o.cancel_scale(want)

    plans = {}
    for t in range(topo.ntiles):
        p, row, col = (int(x) for x in g[f"p{t}/tile/panel_row_col"])
        assert topo.tile(p, row, col) == t
        m = {kk: _dev(v) for kk, v in g.sub(t, "metric").items() if kk != "inv_sqrtG"}
        m.update({kk: _dev(v) for kk, v in g.sub(t, "topo").items()})
        m["boundary_sn"] = _dev(g[f"p{t}/geom/boundary_sn"])
        m["boundary_we"] = _dev(g[f"p{t}/geom/boundary_we"])
        plans[t] = SwPlan(g.n, g.H, p, g.ops, m, on_panel_edge=topo.on_panel_edge(t))
    for t in (0, 3, 5, 10, 17, 23):
        plan = plans[t]
        q = _dev(g.q(t))
        send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        plan.extrap_pack(q, [send[e].data_ptr() for e in range(4)])
        out = torch.full_like(q, float("nan"))
        halo = [_dev(h) for h in g.halo(t)]  # (kept alive: the call takes raw pointers)
        plan.rhs(q, [h.data_ptr() for h in halo], out)
        torch.cuda.synchronize()
        got = send.cpu().numpy().reshape(4, 3, g.H, g.n)
        for e in range(4):
            ref = g.halo(topo.neighbor(t, e))[topo.landing(t, e)]
            assert np.abs(got[e] - ref).max() <= 1e-13 * np.abs(ref).max(), (t, e)
        ref = g.r(t)
        scale = np.maximum(var_max(ref), _scale(g, t, False))
        assert (var_err(out.cpu().numpy(), ref) <= TOL * scale).all(), t
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, tiles_per_side=k)
    rhs = RhsShallowWater(plans, ex)
    Q = torch.stack([_dev(g.q(t)) for t in range(topo.ntiles)])
    Rd = rhs(Q)
    R = Rd.cpu().numpy()
    for t in range(topo.ntiles):
        ref = g.r(t)
        scale = np.maximum(var_max(ref), _scale(g, t, False))
        assert (var_err(R[t], ref) <= TOL * scale).all(), t
    # the direct form on the 24 tiles: tile edges INSIDE a panel (no rotation, no flip) and on panel edges, every line pulled
    # from the neighbour tile's nodal values - the same bits as the two-kernel form
    one = RhsShallowWater(plans, PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, tiles_per_side=k))
    one.direct = True
    got = one(Q)
    assert one._batches[torch.float64].pulls and not bool(one.ex.send_buf.any())
    assert ((got - Rd).abs() <= 1e-15 * Rd.abs().amax(dim=(0, 2, 3), keepdim=True)).all()


@pytest.mark.parametrize("name", SW_FIXTURES)
def test_direct_form_equals_the_two_kernel_form(built_lib, name, monkeypatch):
    """wx_sw_rhs_direct / wx_sw_batch_rhs_direct: no interface buffer - the RHS kernel extrapolates its own face states from
    LDS and the neighbours' from the neighbour elements' nodal values; only the ring of tile-edge lines is packed before
    the exchange.  Same arithmetic term by term: the result equals the two-kernel form to the last bits (and with it the
    reference's R), per tile and batched, with topography, R(Q) and the fused stage update, whole tiles and the
    INTERIOR / BOUNDARY launches of a travelling exchange."""
    from wxfactory_amd import _lib
    from wxfactory_amd.exchange import PanelExchange, RcclComm
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw(name)
    plans = {p: _plan(g, p) for p in range(6)}
    Q = torch.stack([_dev(g.q(p)) for p in range(6)])
    two = RhsShallowWater(plans)
    want, want_axpy = two(Q), two.axpy(Q, Q, 0.75, 0.25, 12.5)
    scale = want.abs().amax(dim=(0, 2, 3), keepdim=True)
    comm = RcclComm(0, 1, device=DEV)
    for loop in (False, True):
        ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=loop, backend="rccl" if loop else "torch",
                           comm=comm if loop else None)
        one = RhsShallowWater(plans, ex)
        one.direct = True
        got, got_axpy = one(Q), one.axpy(Q, Q, 0.75, 0.25, 12.5)
        torch.cuda.synchronize()
        assert ((got - want).abs() <= 1e-15 * scale).all(), (name, loop, float(((got - want).abs() / scale).max()))
        assert ((got_axpy - want_axpy).abs() <= 1e-15 * want_axpy.abs().amax(dim=(0, 2, 3), keepdim=True)).all(), (name, loop)
        # halos that alias the neighbours' send lines: the launch forms the tile-edge lines itself from the neighbour tiles' nodal
        # values (wx_sw_batch_direct_pulls) and nothing is packed; halos that travel: the ring pack + the exchange
        assert one._batches[torch.float64].pulls == (not loop)
        if not loop:
            assert not bool(one.ex.send_buf.any())
            # ... against the packed lines (WXHIP_SW_PULL=0 when the batch is created): the same bits, and the ring-only pack
            # wrote the same edge lines as the full extrapolation
            monkeypatch.setenv("WXHIP_SW_PULL", "0")
            packed = RhsShallowWater(plans, PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1))
            packed.direct = True
            got_p, got_axpy_p = packed(Q), packed.axpy(Q, Q, 0.75, 0.25, 12.5)
            Qc = torch.complex(Q, 1e-3 * torch.roll(Q, 1, dims=0))
            got_pc = packed(Qc)   # (a batch per dtype, created at its first call: this one too while the switch is set)
            monkeypatch.delenv("WXHIP_SW_PULL")
            assert not packed._batches[torch.float64].pulls
            twoex = RhsShallowWater(plans, PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1))
            twoex.direct = False
            twoex(Q)
            torch.cuda.synchronize()
            assert torch.equal(got_p, got) and torch.equal(got_axpy_p, got_axpy), name
            assert torch.equal(packed.ex.send_buf, twoex.ex.send_buf)
            # ... and on complex states (the complex-step matvec's dtype): pulled = packed = two kernels
            want_c, got_c = two(Qc), one(Qc)
            assert one._batches[torch.complex128].pulls and not packed._batches[torch.complex128].pulls
            assert torch.equal(got_c, got_pc), name
            sc = want_c.abs().amax(dim=(0, 2, 3), keepdim=True)
            assert ((got_c - want_c).abs() <= 1e-15 * sc).all(), (name, float(((got_c - want_c).abs() / sc).max()))
    # one tile through the plan: the per-tile entry points, halos from the fixture
    p = 3
    q = _dev(g.q(p))
    halo = [_dev(h) for h in g.halo(p, False)]
    send = torch.zeros((4, plans[p].edge_count), dtype=torch.float64, device=DEV)
    plans[p].extrap_pack_ring(q, [send[e].data_ptr() for e in range(4)])
    out = torch.full_like(q, float("nan"))
    plans[p].rhs_direct(q, [h.data_ptr() for h in halo], out, _lib.WX_REGION_ALL)
    out2 = torch.full_like(q, float("nan"))
    if g.H > 2:
        plans[p].rhs_direct(q, None, out2, _lib.WX_REGION_INTERIOR)
        plans[p].rhs_direct(q, [h.data_ptr() for h in halo], out2, _lib.WX_REGION_BOUNDARY)
    torch.cuda.synchronize()
    ref = g.r(p, False)
    sc = np.maximum(var_max(ref), _scale(g, p, False))
    assert (var_err(out.cpu().numpy(), ref) <= TOL * sc).all()
    if g.H > 2:
        assert torch.equal(out, out2)
    comm.close()


@pytest.mark.parametrize("topo", [False, True])
def test_stage_pipeline_equals_the_fused_stages(built_lib, topo):
    """wx_sw_stage / wx_sw_batch_stage: the stage's kernel extrapolates its own output (the next stage's state) to the faces
    and packs its edge lines, so the next stage has no extrapolation launch.  Three SSP-RK3 stages that way equal the
    fused stages with a separate extrapolation (RhsShallowWater.axpy) to rounding - per tile and batched, flat and with
    topography, aliasing halos and halos that travel (loopback exchange behind the C ABI: INTERIOR / BOUNDARY launches each
    write the faces of their own elements) - and a state that was modified in between is extrapolated afresh."""
    from wxfactory_amd.exchange import PanelExchange, RcclComm
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw("sw_c5_n4_h3" if topo else "sw_c2p_n8_h3")
    plans = {p: _plan(g, p) for p in range(6)}
    Q = torch.stack([_dev(g.q(p)) for p in range(6)])
    dt = 50.0
    plain = RhsShallowWater(plans)
    P1 = plain.axpy(Q, None, 0.0, 1.0, dt)
    P2 = plain.axpy(P1, Q, 0.75, 0.25, 0.25 * dt)
    P3 = plain.axpy(P2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt)
    scale = P3.abs().amax(dim=(0, 2, 3), keepdim=True)
    comm = RcclComm(0, 1, device=DEV)
    for batched in (True, False):
        for loop in (False, True):
            ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=loop, backend="rccl" if loop else "torch",
                               comm=comm if loop else None)
            piped = RhsShallowWater(plans, ex)
            piped.batched = batched
            piped.direct = False   # (the automatic choice at n = 8 is the direct form, whose fused stage is one launch already)
            Q1 = piped.stage(Q, None, 0.0, 1.0, dt)
            Q2 = piped.stage(Q1, Q, 0.75, 0.25, 0.25 * dt)
            Q3 = piped.stage(Q2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt)
            torch.cuda.synchronize()
            assert ((Q3 - P3).abs() <= 1e-14 * scale).all(), (batched, loop)
            # the prepared faces belong to Q3 as returned: an in-place change is seen (version counter) and extrapolated afresh
            Q3.mul_(1.0 + 1e-3)
            Q4 = piped.stage(Q3, None, 0.0, 1.0, dt)
            want = plain.axpy(Q3, None, 0.0, 1.0, dt)
            torch.cuda.synchronize()
            assert ((Q4 - want).abs() <= 1e-14 * scale).all(), (batched, loop)
            assert Tvdrk3(piped).pipeline
    torch.cuda.synchronize()
    comm.close()


@pytest.mark.parametrize("name", ["sw_rk3_c6_n5_h4", "sw_rk3_c6_n8_h3"])
@pytest.mark.parametrize("fused,batched", [(True, True), (True, False), (False, True), ("auto", True)])
def test_rk3_time_loop_matches_reference(name, fused, batched, built_lib):
    """The explicit time loop of BASELINE configs 2 / 3 - Tvdrk3.step + apply_filters, simulation.py:147-155 with
    integrators/tvdrk3.py:12-19 on rhs/rhs_sw.py - against the reference's own run of it: the state after one and after
    five steps (config/case6.ini at p = 4, and at p = 7 = the order of config 3)."""
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw(name)
    rhs = RhsShallowWater({p: _plan(g, p) for p in range(6)})
    rhs.batched = batched
    if fused == "auto":
        # the default: float64 at n = 8 takes the direct form (one launch per fused stage, no interface buffer), every other
        # order the stage pipeline of the two-kernel form
        stepper = Tvdrk3(rhs)
        assert rhs._use_direct(torch.float64) == (g.n == 8) and stepper.pipeline == (g.n != 8) and stepper.fused
    else:
        rhs.direct = False
        stepper = Tvdrk3(rhs, fused=fused)   # (fused: the stage pipeline - wx_sw_stage / wx_sw_batch_stage)
        assert stepper.pipeline == fused
    Q0 = torch.stack([_dev(g.q(p)) for p in range(6)])
    dt, nsteps = float(g["meta/rk3_dt"]), int(g["meta/rk3_steps"])
    stack = lambda key: np.stack([g[f"p{p}/{key}"] for p in range(6)])  # noqa: E731
    Q = Q0
    for i in range(nsteps):
        Q = stepper.step(Q, dt)
        if i in (0, nsteps - 1):
            ref = stack("rk3_1" if i == 0 else "rk3_n")
            moved = np.abs(ref - stack("Q")).max(axis=(0, 2, 3, 4))
            err = np.abs(Q.cpu().numpy() - ref).max(axis=(0, 2, 3, 4))
            assert (moved > 0).all() and (err <= 1e-9 * moved + 1e-14 * np.abs(ref).max(axis=(0, 2, 3, 4))).all(), (i, err, moved)


def test_case6_ini_integrator_epi3_with_pmex(built_lib):
    """config/case6.ini as shipped - Epi order 3 (one EPI2 start-up step, then the multistep form), exponential solver
    pmex, complex-step JVP, tolerance 1e-7, dt = 1800 s - against the reference's own run of it (integrators/epi.py,
    solvers/pmex.py on rhs/rhs_sw.py): the state after each of four steps and pmex's statistics of every step."""
    from wxfactory_amd.integrators import Epi
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw("sw_epi3_pmex_c6_n5_h4")
    assert str(g["meta/epi_solver"]) == "pmex"
    rhs = RhsShallowWater({p: _plan(g, p) for p in range(6)})
    stepper = Epi(int(g["meta/epi_order"]), rhs, tol=float(g["meta/epi_tol"]), exponential_solver="pmex")
    dt, nsteps = float(g["meta/epi_dt"]), int(g["meta/epi_steps"])
    ref_stats = g["meta/epi_solver_stats_all"][::6]   # (all six emulated ranks logged each call)
    stack = lambda key: np.stack([g[f"p{p}/{key}"] for p in range(6)])  # noqa: E731
    Q = torch.stack([_dev(g.q(p)) for p in range(6)])
    prev = stack("Q")
    for i in range(nsteps):
        Q = stepper.step(Q, dt)
        info = stepper.solver_info
        got = [info[k] for k in ("substeps", "rejected", "iterations", "exps", "krylov_size", "own_norms")]
        assert got == [int(ref_stats[i][k]) for k in (0, 1, 2, 3, 5, 6)], (i, got, ref_stats[i])
        ref = stack(f"epi_{i + 1}")
        moved = np.abs(ref - prev).max(axis=(0, 2, 3, 4))
        err = np.abs(Q.cpu().numpy() - ref).max(axis=(0, 2, 3, 4))
        # both sides solve each step to the tolerance 1e-7 (relative to the update)
        assert (moved > 0).all() and (err <= 1e-6 * moved).all(), (i, err, moved)
        prev = ref


@pytest.mark.parametrize("case", ["galewsky", 5])
def test_own_geometry_and_initial_states_against_the_oracle(built_lib, case):
    """The whole chain a run without fixtures takes - own geometry + metric (geometry.py), own initial state
    (initial_sw.py: the Galewsky jet with its bump; Williamson 5 with its mountain and the topography arrays), all six
    panels through RhsShallowWater - against the NumPy oracle on the same inputs at 1e-10, and the balance property at
    a size the oracle no longer covers: on the S7 sphere the jet without its bump is steady to truncation error."""
    from oracle.sw2d import SW2DOracle
    from wxfactory_amd import initial_sw, synthetic
    from wxfactory_amd.geometry import CubedSphereTile2D, metric2d
    from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan

    n, H = 8, 6
    ops = synthetic.dfr_ops(n)
    tiles = [CubedSphereTile2D(n, H, p) for p in range(6)]
    metrics = [metric2d(t) for t in tiles]
    if case == "galewsky":
        h0 = initial_sw.galewsky_h0(tiles[0].earth_radius, tiles[0].rotation_speed)
        states = [(initial_sw.galewsky(t, True, h0), None) for t in tiles]
    else:
        states = [initial_sw.williamson5(t, ops["diff_solpt"], ops["correction"]) for t in tiles]
    oracles = [SW2DOracle(n, H, ops, m, tp, t.boundary_sn, t.boundary_we, panel=p)
               for p, (t, m, (_, tp)) in enumerate(zip(tiles, metrics, states))]
    from oracle import cubed_sphere as cs

    itfs = [o.extrapolate(q) for o, (q, _) in zip(oracles, states)]
    recvs = cs.route([o.pack_edges(itf) for o, itf in zip(oracles, itfs)])
    terms = [{} for _ in range(6)]
    want = [o.rhs(states[p][0], recvs[p], itf=itfs[p], want=terms[p]) for p, o in enumerate(oracles)]
    plans = {p: SwPlan(n, H, p, ops, {k: _dev(v) for k, v in {**metrics[p], **(states[p][1] or {})}.items()}) for p in range(6)}
    rhs = RhsShallowWater(plans)
    got = rhs(torch.stack([_dev(q) for q, _ in states]))
    torch.cuda.synchronize()
    for p in range(6):
        R, ref = got[p].cpu().numpy(), want[p]
        scale = np.maximum(var_max(ref), SW2DOracle.cancel_scale(terms[p]))
        assert (var_err(R, ref) <= TOL * scale).all(), (case, p, var_err(R, ref) / scale)
    if case != "galewsky":
        return
    n, H = 8, 60   # S7
    tiles = [CubedSphereTile2D(n, H, p) for p in range(6)]
    plans = {p: SwPlan(n, H, p, ops, {k: _dev(v) for k, v in metric2d(t).items()}) for p, t in enumerate(tiles)}
    rhs = RhsShallowWater(plans)
    jet = torch.stack([_dev(initial_sw.galewsky(t, False, h0)) for t in tiles])
    full = torch.stack([_dev(initial_sw.galewsky(t, True, h0)) for t in tiles])
    Rj, Rf = rhs(jet), rhs(full)
    torch.cuda.synchronize()
    # S7 itself (BASELINE config 3's size: 6 x 60 x 60 elements, 1.38 M points) against the NumPy oracle, all six panels, every
    # row, at 1e-10 - through the default form of the evaluation at this size (one launch, tile-edge lines pulled by the kernel)
    assert rhs._use_direct(torch.float64)
    metrics7 = [metric2d(t) for t in tiles]
    oracles7 = [SW2DOracle(n, H, ops, m, None, t.boundary_sn, t.boundary_we, panel=p) for p, (t, m) in enumerate(zip(tiles, metrics7))]
    q7 = [initial_sw.galewsky(t, True, h0) for t in tiles]
    itf7 = [o.extrapolate(q) for o, q in zip(oracles7, q7)]
    recv7 = cs.route([o.pack_edges(itf) for o, itf in zip(oracles7, itf7)])
    for p, o in enumerate(oracles7):
        terms7 = {}
        ref = o.rhs(q7[p], recv7[p], itf=itf7[p], want=terms7)
        scale = np.maximum(var_max(ref), SW2DOracle.cancel_scale(terms7))
        err = var_err(Rf[p].cpu().numpy(), ref)
        assert (err <= TOL * scale).all(), ("S7", p, err / scale)
    assert float(Rj[:, 0].abs().max()) < 1e-7                   # m/s of depth (1.3e-3 at 8 x 8 elements per panel)
    assert 1e-3 < float(Rf[:, 0].abs().max()) < 1e-1            # the bump's gravity waves set off
    assert float((Rj[:, 1:].abs().amax(dim=(0, 2, 3, 4)) / jet[:, 1:].abs().amax(dim=(0, 2, 3, 4))).max()) < 1e-9   # 1/s


def test_galewsky_steps_conserve_mass(built_lib):
    """Twenty SSP-RK3 steps of the Galewsky jet + bump on own geometry through the stage pipeline (wx_sw_batch_stage): the
    global mass  sum_panels sum w sqrt(g) h  is conserved to rounding (the scheme is conservative: interface fluxes are
    single-valued, the exchange rotates only momentum) and the state stays finite and close to the jet."""
    from wxfactory_amd import initial_sw, synthetic
    from wxfactory_amd.geometry import CubedSphereTile2D, gauss_legendre, metric2d
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan

    n, H = 5, 8
    ops = synthetic.dfr_ops(n)
    tiles = [CubedSphereTile2D(n, H, p) for p in range(6)]
    metrics = [metric2d(t) for t in tiles]
    plans = {p: SwPlan(n, H, p, ops, {k: _dev(v) for k, v in metrics[p].items()}) for p in range(6)}
    rhs = RhsShallowWater(plans)
    h0 = initial_sw.galewsky_h0(tiles[0].earth_radius, tiles[0].rotation_speed)
    Q = torch.stack([_dev(initial_sw.galewsky(t, True, h0)) for t in tiles])
    w = gauss_legendre(n)[1]
    W = torch.stack([_dev(m["sqrtG"] * np.outer(w, w).reshape(-1)) for m in metrics])

    def mass(q):
        return float((W * q[:, 0]).sum())

    m0, q0 = mass(Q), Q.clone()
    stepper = Tvdrk3(rhs)
    for _ in range(20):
        Q = stepper.step(Q, 60.0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(Q).all())
    assert abs(mass(Q) - m0) <= 1e-12 * abs(m0), (mass(Q), m0)
    assert float((Q[:, 0] - q0[:, 0]).abs().max()) < 50.0   # 20 minutes: the bump (120 m) has begun to spread, nothing more
