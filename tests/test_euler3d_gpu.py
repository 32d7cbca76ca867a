"""Parity of the HIP path (through the C ABI) with the reference's golden vectors and the oracle.

Bar (BASELINE.json north star): fp64 agreement <= 1e-10 in the cancellation-aware norm of
SURVEY.md section 7:  max_v |R_v - Rref_v|_inf / max(|Rref_v|_inf, s_v).
"""
import numpy as np
import pytest
import torch

from tests.util import tight_tangent, EDGE_FIELDS, EULER_FIXTURES, MONOLITH_FIXTURES, golden, halo7, make_oracle, var_err, var_max

pytestmark = pytest.mark.gpu

TOL = 1e-10
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded(built_lib):
    from wxfactory_amd import _lib

    lib = _lib.load()
    assert lib.wx_device_count() >= 1, "no HIP device: the -m gpu tests need a real MI355X"
    return lib


def _scale(g, p, cplx):
    o = make_oracle(g, p)
    want = {}
    o.rhs(g.q(p, cplx), g.halo(p, cplx), want=want)
    return o.cancel_scale(want)


@pytest.mark.parametrize("name", EULER_FIXTURES)
def test_pack_matches_reference_halos(name):
    """K1: what each panel packs for its 4 edges is exactly what the reference delivered to the
    neighbour (process_topology.py rotation + flip + neighbour-alltoall routing)."""
    from tests.gpu_util import to_dev
    from wxfactory_amd.panels import NEIGHBOR, landing_edge
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden(name)
    for cplx in (False, True):
        dtype = torch.complex128 if cplx else torch.float64
        for p in g.metric_panels():
            from tests.gpu_util import make_plan

            plan = make_plan(g, p, dtype)
            q = to_dev(g.q(p, cplx))
            send = torch.zeros((4, plan.edge_count), dtype=dtype, device=DEV)
            plan.extrap_pack(q, [send[e].data_ptr() for e in range(4)])
            torch.cuda.synchronize()
            got = send.cpu().numpy().reshape(4, EDGE_FIELDS, g.V, g.H, g.n**2)
            for e in range(4):
                nb, e2 = NEIGHBOR[p][e], landing_edge(p, e)
                ref = halo7(g.halo(nb, cplx)[e2])
                for v in range(EDGE_FIELDS):  # 0-4: the reference's message (5, 6: pressure, log p in NQ=7 builds)
                    assert np.abs(got[e][v] - ref[v]).max() <= 1e-13 * np.abs(ref[v]).max(), (name, p, e, v, cplx)
            plan.close()


@pytest.mark.parametrize("name", EULER_FIXTURES)
@pytest.mark.parametrize("cplx", [False, True])
def test_rhs_matches_reference(name, cplx):
    from tests.gpu_util import make_plan, to_dev

    g = golden(name)
    dtype = torch.complex128 if cplx else torch.float64
    for p in g.metric_panels():
        plan = make_plan(g, p, dtype)
        q = to_dev(g.q(p, cplx))
        halo = [to_dev(halo7(h)) for h in g.halo(p, cplx)]
        out = torch.full_like(q, float("nan"))
        plan.extrap_pack(q, None)
        plan.rhs(q, [h.data_ptr() for h in halo], out)
        torch.cuda.synchronize()
        R = out.cpu().numpy()
        ref = g.r(p, cplx)
        s = _scale(g, p, cplx)
        scale = np.maximum(var_max(ref.real), s)
        err = var_err(R.real, ref.real)
        assert np.isfinite(R.view(np.float64)).all()
        assert (err <= TOL * scale).all(), (name, p, cplx, err / scale)
        if cplx:
            tight = tight_tangent(name)  # see tests/test_oracle_euler3d.py on max() tie-breaks
            ierr = var_err(R.imag, ref.imag) / np.maximum(var_max(ref.imag), g.eps * s * 1e-3)
            assert (ierr <= (1e-10 if tight else 1e-3)).all(), (name, p, ierr)
        plan.close()


@pytest.mark.parametrize("name", MONOLITH_FIXTURES)
def test_rhs_matches_the_reference_monolith(name):
    """the HIP path against R from the reference's one-function rhs/rhs_euler.py:158-517."""
    from tests.gpu_util import make_plan, to_dev

    g = golden(name)
    for p in g.metric_panels():
        plan = make_plan(g, p)
        q = to_dev(g.q(p))
        halo = [to_dev(halo7(h)) for h in g.halo(p)]
        out = torch.full_like(q, float("nan"))
        plan.extrap_pack(q, None)
        plan.rhs(q, [h.data_ptr() for h in halo], out)
        torch.cuda.synchronize()
        mono = g[f"p{p}/R_mono"]
        scale = np.maximum(var_max(mono), _scale(g, p, False))
        err = var_err(out.cpu().numpy(), mono)
        assert (err <= TOL * scale).all(), (name, p, err / scale)
        plan.close()


@pytest.mark.parametrize("name", ["euler3d_c31p_n3_h4_v2", "euler3d_c21_n4_h3_v4"])
def test_regions_compose(name):
    """INTERIOR + BOUNDARY launches write exactly what one ALL launch writes."""
    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd import _lib

    g = golden(name)
    p = g.metric_panels()[-1]
    plan = make_plan(g, p)
    q = to_dev(g.q(p))
    keep = [to_dev(halo7(h)) for h in g.halo(p)]
    halo = [k.data_ptr() for k in keep]
    a = torch.full_like(q, float("nan"))
    b = torch.full_like(q, float("nan"))
    plan.extrap_pack(q, None)
    plan.rhs(q, halo, a, _lib.WX_REGION_ALL)
    plan.rhs(q, None, b, _lib.WX_REGION_INTERIOR)
    plan.rhs(q, halo, b, _lib.WX_REGION_BOUNDARY)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    plan.close()


def test_whole_sphere_on_one_gpu():
    """Six panels resident on one GPU, exchange by zero-copy aliasing of the pack buffers:
    the N=1 leg of the strong-scaling benchmark."""
    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    g = golden("euler3d_c31p_n3_h4_v2")
    plans = {p: make_plan(g, p) for p in range(6)}
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1)
    rhs = RhsEuler3D(plans, ex)
    qs = {p: to_dev(g.q(p)) for p in range(6)}
    Rs = rhs(qs)
    torch.cuda.synchronize()
    for p in range(6):
        ref = g.r(p)
        err = var_err(Rs[p].cpu().numpy(), ref)
        assert (err <= TOL * np.maximum(var_max(ref), _scale(g, p, False))).all(), (p, err)
        assert Rs[p].shape == qs[p].shape and Rs[p].dtype == qs[p].dtype


@pytest.mark.parametrize("name", ["euler3d_c31p_n3_h4_v2", "euler3d_c31p_n8_h2_v2", "euler3d_c21_n4_h3_v4", "euler3d_c21p_n4_h3_v4",
                                  "euler3d_c21p_n8_h2_v2"])
def test_dual_number_arithmetic_equals_complex_step(name):
    """WX_DUAL128 (first-order arithmetic in complex128 storage) returns the reference's complex-step
    result: same real part, same tangent Im R / eps, to the 1e-10 bound."""
    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden(name)
    for p in g.metric_panels():
        plan = Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV), dtype=torch.complex128, dual=True)
        q = to_dev(g.q(p, True))
        halo = [to_dev(halo7(h)) for h in g.halo(p, True)]
        out = torch.full_like(q, float("nan"))
        plan.extrap_pack(q, None)
        plan.rhs(q, halo, out)
        torch.cuda.synchronize()
        R, ref = out.cpu().numpy(), g.r(p, True)
        s = _scale(g, p, True)
        assert (var_err(R.real, ref.real) <= TOL * np.maximum(var_max(ref.real), s)).all()
        tight = tight_tangent(name)
        ierr = var_err(R.imag, ref.imag) / np.maximum(var_max(ref.imag), g.eps * s * 1e-3)
        assert (ierr <= (1e-10 if tight else 1e-3)).all(), (name, p, ierr)
        plan.close()


def test_tiles_of_a_24_rank_decomposition():
    """k x k tiles per panel (the reference's 6 k^2 ranks): plans with on_panel_edge flags reproduce what the
    reference packed and computed on 24 MPI ranks (interior tile edges unrotated/unflipped)."""
    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden("euler3d_tiles24_n3_h2_v2")
    topo = CubeTopology(int(g["meta/k"]))
    for t in g.metric_panels():
        plan = Euler3DPlan(g.n, g.H, g.V, g.case, topo.locate(t)[0], g.ops, device_metric(g, t, DEV),
                           on_panel_edge=topo.on_panel_edge(t))
        q = to_dev(g.q(t))
        send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        plan.extrap_pack(q, list(send))
        out = torch.full_like(q, float("nan"))
        halo = [to_dev(halo7(h)) for h in g.halo(t)]  # (kept alive until the launch has run)
        plan.rhs(q, halo, out)
        torch.cuda.synchronize()
        got = send.cpu().numpy().reshape(4, EDGE_FIELDS, g.V, g.H, g.n**2)
        for e in range(4):
            ref = halo7(g.halo(topo.neighbor(t, e))[topo.landing(t, e)])
            assert np.abs(got[e] - ref).max() <= 1e-13 * np.abs(ref).max(), (t, e)
        ref = g.r(t)
        assert (var_err(out.cpu().numpy(), ref) <= TOL * np.maximum(var_max(ref), _scale_tile(g, t, topo))).all(), t
        plan.close()


def _scale_tile(g, t, topo):
    from oracle.euler3d import Euler3DOracle

    o = Euler3DOracle(g.n, g.H, g.V, g.case, g.ops, g.metric(t), g[f"p{t}/geom/boundary_sn_new"],
                      g[f"p{t}/geom/boundary_we_new"], panel=topo.locate(t)[0], on_panel_edge=topo.on_panel_edge(t))
    want = {}
    o.rhs(g.q(t), g.halo(t), want=want)
    return o.cancel_scale(want)


@pytest.mark.parametrize("name,ztop", [("euler3d_c21_n4_h3_v4", 30000.0), ("euler3d_c31p_n8_h2_v2", 10000.0),
                                       ("euler3d_c31p_n5_h2_v1", 10000.0)])
def test_rhs_with_own_geometry_and_metric(name, ztop):
    """End to end without any reference-supplied array but the state: geometry3d builds the metric of all six
    panels (Schaer mountain + sponge for case 21), the kernels evaluate R(Q), the reference's R is the check."""
    from tests.gpu_util import to_dev
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    g = golden(name)
    topo = topography_for_case(g.case, planet_for_case(g.case)[0])
    plans = {}
    for p in range(6):
        t = CubedSphere3DTile(g.n, g.H, g.V, p, ztop, g.case, topo=topo)
        plans[p] = Euler3DPlan(g.n, g.H, g.V, g.case, p, dfr_ops(g.n), metric3d_torch(t, DEV))
    Rs = RhsEuler3D(plans)({p: to_dev(g.q(p)) for p in range(6)})
    torch.cuda.synchronize()
    scales = {p: _scale(g, p, False) for p in g.metric_panels()}
    floor = np.max(np.stack(list(scales.values())), axis=0)
    for p in range(6):
        ref = g.r(p)
        err = var_err(Rs[p].cpu().numpy(), ref)
        assert (err <= TOL * np.maximum(var_max(ref), scales.get(p, floor))).all(), (p, err)


def test_metric3d_on_the_gpu_equals_numpy():
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d, planet_for_case, topography_for_case

    t = CubedSphere3DTile(4, 3, 2, 1, 30000.0, 21, topo=topography_for_case(21, planet_for_case(21)[0]))
    a, b = metric3d(t, threads=1), metric3d(t, device=DEV)
    for k in a:
        assert b[k].is_cuda and tuple(b[k].shape) == a[k].shape
        tol = 1e-11 if k == "christoffel" else 1e-13
        assert np.abs(b[k].cpu().numpy() - a[k]).max() <= tol * np.abs(a[k]).max(), k


def test_plan_skips_identically_zero_rotation_symbols():
    """Plan-time specialisation: on a non-rotating planet (every DCMIP fixture here) christoffel[:, 0:3] is
    identically zero and the kernels do not read it; the result is the same as with the loads."""
    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden("euler3d_c31p_n8_h2_v2")
    m = device_metric(g, 0, DEV)
    assert float(m["christoffel"].reshape(3, 9, -1)[:, :3].abs().max()) == 0.0
    plan = Euler3DPlan(g.n, g.H, g.V, g.case, 0, g.ops, m)
    assert plan.bytes_per_point == 8 * (5 + 5 + 1 + 6 + 18 + 1) + 24
    m2 = dict(m)
    m2["christoffel"] = m["christoffel"].clone()
    m2["christoffel"].view(3, 9, -1)[0, 0, 0] = 1e-300  # one tiny non-zero value: the full-load path
    full = Euler3DPlan(g.n, g.H, g.V, g.case, 0, g.ops, m2)
    assert full.bytes_per_point == 384.0
    q = to_dev(g.q(0))
    halo = [to_dev(h) for h in g.halo(0)]
    send = [torch.empty_like(h) for h in halo]
    a, b = torch.empty_like(q), torch.empty_like(q)
    for pl, out in ((plan, a), (full, b)):
        pl.extrap_pack(q, send)
        pl.rhs(q, halo, out)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_rhs_timing_interface():
    """RHS.timestamps / timings / retrieve_last_times / clear_timings of the reference (rhs/rhs.py:39-40, 69-121)."""
    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    g = golden("euler3d_c31p_n3_h4_v2")
    rhs = RhsEuler3D({p: make_plan(g, p) for p in range(6)})
    qs = {p: to_dev(g.q(p)) for p in range(6)}
    rhs(qs)
    assert not getattr(rhs, "timestamps", [])  # off by default
    rhs.timed = True
    rhs.clear_timings()
    for _ in range(3):
        rhs(qs)
    rhs.retrieve_last_times()
    assert len(rhs.timings) == 3
    for t in rhs.timings:
        assert len(t) == 9 and all(x >= 0.0 for x in t) and abs(sum(t[:8]) - t[8]) < 1e-4 and t[8] > 0.0
    rhs.clear_timings()
    assert rhs.timings == [] and rhs.timestamps == []


def test_phase_timer_of_the_c_abi():
    """The reference's nine-stamp timing row (rhs/rhs.py:88-118, device.elapsed) for a caller that drives the two kernels
    itself, without torch: wx_phase_timer_* on the launch stream."""
    import ctypes

    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd import _lib
    from wxfactory_amd._lib import check

    lib = _lib.load()
    g = golden("euler3d_c31p_n3_h4_v2")
    p = g.metric_panels()[0]
    plan = make_plan(g, p)
    q = to_dev(g.q(p))
    halo = [to_dev(halo7(h)) for h in g.halo(p)]
    out = torch.empty_like(q)
    t = ctypes.c_void_p()
    check(lib.wx_phase_timer_create(ctypes.byref(t)), "wx_phase_timer_create")
    st = torch.cuda.current_stream().cuda_stream
    row = (ctypes.c_double * 9)()
    assert lib.wx_phase_timer_elapsed(t, row) != 0          # nothing stamped yet
    for _ in range(2):
        check(lib.wx_phase_timer_stamp(t, 0, st), "stamp")
        plan.extrap_pack(q, None)
        check(lib.wx_phase_timer_stamp(t, 1, st), "stamp")
        check(lib.wx_phase_timer_stamp(t, 2, st), "stamp")    # (exchange posted: nothing travels here)
        plan.rhs(q, None, out, _lib.WX_REGION_INTERIOR)
        check(lib.wx_phase_timer_stamp(t, 4, st), "stamp")    # slot 3 shares its kernel with slot 4
        check(lib.wx_phase_timer_stamp(t, 5, st), "stamp")
        plan.rhs(q, halo, out, _lib.WX_REGION_BOUNDARY)
        check(lib.wx_phase_timer_stamp(t, 8, st), "stamp")    # slots 6, 7 share their kernel with slot 8
        check(lib.wx_phase_timer_elapsed(t, row), "wx_phase_timer_elapsed")
        vals = list(row)
        assert all(v >= 0.0 for v in vals) and vals[2] == 0.0 and vals[5] == 0.0 and vals[6] == 0.0
        assert vals[0] > 0.0 and vals[3] > 0.0 and vals[7] > 0.0
        assert abs(sum(vals[:8]) - vals[8]) <= 1e-4 and vals[8] > 0.0
    assert lib.wx_phase_timer_stamp(t, 9, st) != 0
    check(lib.wx_phase_timer_destroy(t), "wx_phase_timer_destroy")
    plan.close()


@pytest.mark.parametrize("name", ["euler3d_c31p_n3_h4_v2", "euler3d_c21_n4_h3_v4", "euler3d_c31p_n8_h2_v2"])
def test_batched_launch_equals_per_tile_launches(name):
    """Stacked states go through ONE launch per phase for all tiles (wx_euler3d_batch_*): identical bits to the
    per-tile launches, plain and with the fused stage update, float64 and complex128; and faster where it is meant
    to be (small tiles are launch-bound)."""
    import time

    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    g = golden(name)
    if len(g.metric_panels()) == 6:
        metrics = {p: device_metric(g, p, DEV) for p in range(6)}
    else:  # fixtures that hold the metric of a few panels only: build all six
        from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case

        ztop = 30000.0 if g.case in (21, 22) else 10000.0
        topo = topography_for_case(g.case, planet_for_case(g.case)[0])
        metrics = {p: metric3d_torch(CubedSphere3DTile(g.n, g.H, g.V, p, ztop, g.case, topo=topo), DEV) for p in range(6)}
    rhs = RhsEuler3D({p: Euler3DPlan(g.n, g.H, g.V, g.case, p, dfr_ops(g.n), metrics[p]) for p in range(6)})
    Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
    Y = Q * 0.5 + 1.0
    Qc = torch.complex(Q, 1e-8 * Y)

    def both(fn):
        rhs.batched = True
        a = fn()
        rhs.batched = False
        b = fn()
        rhs.batched = True
        return a, b

    for fn in (lambda: rhs(Q), lambda: rhs.axpy(Q, Y, 0.75, 0.25, 0.1), lambda: rhs.axpy(Q, None, 0.0, 1.0, 0.3, Y, -2.0),
               lambda: rhs(Qc)):
        a, b = both(fn)
        assert torch.equal(a, b)
    ref = np.stack([g.r(p) for p in range(6)])
    assert np.isfinite(rhs(Q).cpu().numpy()).all() and rhs(Q).shape == ref.shape



def test_whole_sphere_of_24_tiles_batched():
    """All 24 tiles of the reference's 24-rank run on one GPU, metric of every tile from geometry3d, the stacked
    state evaluated with one launch per phase for all tiles (zero-copy exchange between them): R of every tile."""
    from tests.gpu_util import to_dev
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    g = golden("euler3d_tiles24_n3_h2_v2")
    k = int(g["meta/k"])
    topo = CubeTopology(k)
    plans = {}
    for t in range(topo.ntiles):
        p, row, col = topo.locate(t)
        tile = CubedSphere3DTile(g.n, g.H, g.V, p, 10000.0, g.case, row=row, col=col, k=k)
        plans[t] = Euler3DPlan(g.n, g.H, g.V, g.case, p, dfr_ops(g.n), metric3d_torch(tile, DEV),
                               on_panel_edge=topo.on_panel_edge(t))
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, tiles_per_side=k)
    rhs = RhsEuler3D(plans, ex)
    assert rhs._small_tiles()
    Q = torch.stack([to_dev(g.q(t)) for t in range(topo.ntiles)])
    R = rhs(Q).cpu().numpy()
    rhs.batched = False
    assert np.array_equal(rhs(Q).cpu().numpy(), R)
    scales = {t: _scale_tile(g, t, topo) for t in g.metric_panels()}
    floor = np.max(np.stack(list(scales.values())), axis=0)
    for t in range(topo.ntiles):
        ref = g.r(t)
        assert (var_err(R[t], ref) <= TOL * np.maximum(var_max(ref), scales.get(t, floor))).all(), t
