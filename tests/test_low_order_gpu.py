"""The low-order one-kernel form (csrc/euler3d_brick.h; num_solpts 2..4, float64) against the two-kernel form on the same
plan, and against the reference's fixtures through every entry point that takes it: whole-tile and INTERIOR + BOUNDARY launches on
tiles whose sides the bricks do not divide, the batched launch, the fused stage update with its edge messages, the shifted state."""
import numpy as np
import pytest
import torch

from tests.util import EDGE_FIELDS, golden, halo7, make_oracle, var_err, var_max

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded(built_lib):
    from wxfactory_amd import _lib

    return _lib.load()


def _tile(n, H, V, panel=4, seed=3):
    from wxfactory_amd import synthetic
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    m = synthetic.euler3d_metric(n, H, V, panel, DEV, seed=seed)
    q = synthetic.euler3d_state(n, H, V, panel, DEV, seed=seed)
    plan = Euler3DPlan(n, H, V, 31, panel, synthetic.dfr_ops(n), m)
    assert plan.one_kernel == (n == 2)   # the default: where the form wins
    plan.set_one_kernel(True)
    return plan, q


def _halos(plan, q):
    """A consistent finite halo: the tile's own packed edges, permuted."""
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
    plan.extrap_pack(q, [send[e].data_ptr() for e in range(4)])
    torch.cuda.synchronize()
    return send, [send[e] for e in (1, 0, 3, 2)]


def _rel(a, b):
    ax = tuple(range(1, a.dim()))
    return float(((a - b).abs().amax(dim=ax) / b.abs().amax(dim=ax).clamp_min(1e-300)).max())


@pytest.mark.parametrize("n,H,V", [(2, 5, 3), (2, 8, 2), (2, 3, 1), (3, 5, 3), (3, 4, 2), (4, 5, 2), (4, 3, 1), (2, 1, 1), (4, 2, 3)])
def test_one_kernel_form_equals_the_two_kernel_form(n, H, V):
    from wxfactory_amd import _lib

    plan, q = _tile(n, H, V)
    assert plan.one_kernel
    send1, halo = _halos(plan, q)
    hp = [h.data_ptr() for h in halo]
    out1 = torch.full_like(q, float("nan"))
    plan.rhs(q, hp, out1)
    # INTERIOR + BOUNDARY write exactly what ALL writes
    out1s = torch.full_like(q, float("nan"))
    plan.rhs(q, None, out1s, _lib.WX_REGION_INTERIOR)
    plan.rhs(q, hp, out1s, _lib.WX_REGION_BOUNDARY)
    torch.cuda.synchronize()
    assert torch.equal(out1, out1s)
    # the two-kernel form on the same plan
    plan.set_one_kernel(False)
    assert not plan.one_kernel
    send2 = torch.zeros_like(send1)
    plan.extrap_pack(q, [send2[e].data_ptr() for e in range(4)])
    out2 = torch.full_like(q, float("nan"))
    plan.rhs(q, hp, out2)
    torch.cuda.synchronize()
    # (the one-kernel form and its pack kernel take their logarithms from the lean form of wx_math.h: an ulp apart from the library's)
    assert _rel(send1, send2) <= 4e-15, _rel(send1, send2)
    assert torch.isfinite(out1).all()
    assert _rel(out1, out2) <= 1e-12, _rel(out1, out2)
    # fused update and shifted state
    y, v = torch.randn_like(q), q * 1e-3 * torch.rand_like(q)
    a2 = torch.empty_like(q)
    plan.shifted_extrap_pack(q, v, 1e-4, None)
    plan.shifted_rhs_axpy(q, v, 1e-4, hp, y, a2, 0.5, 0.25, -2.0)
    plan.set_one_kernel(True)
    a1 = torch.empty_like(q)
    plan.shifted_extrap_pack(q, v, 1e-4, None)
    plan.shifted_rhs_axpy(q, v, 1e-4, hp, y, a1, 0.5, 0.25, -2.0)
    torch.cuda.synchronize()
    assert _rel(a1, a2) <= 1e-12
    plan.close()


@pytest.mark.parametrize("n", [2, 3, 4])
@pytest.mark.parametrize("filtered", [False, True])
def test_stage_update_and_next_messages(n, filtered):
    """wx_euler3d_stage with prepare_next: the stage's output and the edge messages of that output (the next stage's state) -
    one-kernel form against the two-kernel form's pipeline epilogue."""
    from wxfactory_amd import _lib, synthetic

    H, V = 5, 2
    plan, q = _tile(n, H, V)
    plan.reserve(_lib.WX_RESERVE_STAGE)
    if filtered:
        f = np.eye(n) * 0.97 + 0.03 / n
        plan.set_exp_filter(f)
    _, halo = _halos(plan, q)
    hp = [h.data_ptr() for h in halo]
    y = torch.randn_like(q) * 1e-3 + q
    res = {}
    for form in (True, False):
        plan.set_one_kernel(form)
        ns = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        out = torch.empty_like(q)
        plan.extrap_pack_slot(q, None, 0)
        plan.stage(q, hp, y, None, out, 0.75, 0.25, 1e-3, 0.0, _lib.WX_REGION_ALL, 0, [ns[e].data_ptr() for e in range(4)],
                   2 if filtered else 1, flag.data_ptr() if filtered else 0)
        torch.cuda.synchronize()
        res[form] = (out, ns, int(flag.item()))
    assert res[True][2] == 0 and res[False][2] == 0
    assert _rel(res[True][0], res[False][0]) <= 1e-12
    assert _rel(res[True][1], res[False][1]) <= 1e-12
    plan.close()


@pytest.mark.parametrize("name", ["euler3d_c31p_n2_h4_v3", "euler3d_c31p_n3_h4_v2", "euler3d_c21_n4_h3_v4", "euler3d_c21p_n4_h3_v4"])
def test_one_kernel_form_on_the_low_order_fixtures(name):
    """Every low-order fixture of the reference (orders 2, 3, 4; DCMIP 3-1 and 2-1 with its mountain and sponge) through the
    one-kernel form: per panel with the halos the reference delivered (whole-tile and INTERIOR + BOUNDARY launches), and - where
    the fixture holds all six panels - the whole sphere through RhsEuler3D (batched launch, exchange by aliasing); R at 1e-10."""
    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd import _lib
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    g = golden(name)
    panels = g.metric_panels()
    plans = {p: make_plan(g, p) for p in panels}
    for pl in plans.values():
        pl.set_one_kernel(True)
    assert all(pl.one_kernel for pl in plans.values())

    def check(R, p, what):
        o = make_oracle(g, p)
        want = {}
        o.rhs(g.q(p), g.halo(p), want=want)
        scale = np.maximum(var_max(g.r(p)), o.cancel_scale(want))
        err = var_err(R, g.r(p))
        assert (err <= 1e-10 * scale).all(), (name, what, p, err / scale)

    for p in panels:
        q = to_dev(g.q(p))
        halo = [to_dev(halo7(h)) for h in g.halo(p)]
        hp = [h.data_ptr() for h in halo]
        out = torch.full_like(q, float("nan"))
        plans[p].extrap_pack(q, None)
        plans[p].rhs(q, hp, out)
        split = torch.full_like(q, float("nan"))
        plans[p].rhs(q, None, split, _lib.WX_REGION_INTERIOR)
        plans[p].rhs(q, hp, split, _lib.WX_REGION_BOUNDARY)
        torch.cuda.synchronize()
        assert torch.equal(out, split)
        check(out.cpu().numpy(), p, "tile")
    if len(panels) == 6:
        rhs = RhsEuler3D(plans)
        R = rhs(torch.stack([to_dev(g.q(p)) for p in panels])).cpu().numpy()
        for i, p in enumerate(panels):
            check(R[i], p, "sphere")


def test_ros2_step_at_order_2_with_device_passes(monkeypatch):
    """config/dcmip31.ini's order (num_solpts 2) on a small sphere of own geometry: the Rosenbrock-2 + fgmres step with the
    device passes - whose operator is the ONE-KERNEL form here, its shifted state scaled from device memory - against the host
    loop: same iterations, same solution."""
    from wxfactory_amd import synthetic
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.integrators import Ros2
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    n, H, V = 2, 6, 3
    plans, q = {}, []
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
        plans[p] = Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metric3d_torch(t, DEV))
        q.append(torch.from_numpy(initial_state(t)).to(DEV))
    assert all(pl.one_kernel for pl in plans.values())
    rhs = RhsEuler3D(plans)
    Q = torch.stack(q)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WXHIP_FGMRES_VECTOR", mode)
        ros = Ros2(rhs, tol=1e-10)
        res[mode] = (ros.step(Q, 30.0), ros.solver_info)
    dev, host = res["1"][1], res["0"][1]
    assert dev["device_passes"] > 0 and host["device_passes"] == 0
    assert dev["flag"] == 0 and dev["iterations"] == host["iterations"], (dev["iterations"], host["iterations"])
    ax = (0, 2, 3, 4, 5)
    upd = (res["0"][0] - Q).abs().amax(dim=ax)
    assert (((res["1"][0] - res["0"][0]).abs().amax(dim=ax)) <= 1e-7 * upd).all()   # (both solved to 1e-10 of |b|)
    # the same device passes with the Gram-Schmidt step from ONE launch (a barrier of resident workgroups inside it): every step
    # is the three launches' bit for bit, so the whole solve is
    monkeypatch.setenv("WXHIP_FGMRES_VECTOR", "1")
    monkeypatch.setenv("WXHIP_FGMRES_ONE_LAUNCH", "1")
    ros = Ros2(rhs, tol=1e-10)
    one = ros.step(Q, 30.0)
    assert ros.solver_info["flag"] == 0 and ros.solver_info["iterations"] == dev["iterations"]
    assert torch.equal(one, res["1"][0])


def test_the_lean_logarithm_is_good_to_an_ulp():
    """wx_math.h: lean_log (the logarithm of the one-kernel form) against numpy's over the range the path feeds it - densities,
    rho theta, 1e-8 ... 1e9: within 1 ulp of the correctly rounded value wherever |log x| is not tiny, 2e-16 absolute near 1."""
    from wxfactory_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(5)
    x = np.concatenate([10.0 ** rng.uniform(-8, 9, 400_000), 1.0 + rng.uniform(-0.3, 0.45, 100_000), np.array([1.0, 2.0, 0.5, 1e5, 287.05])])
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.empty_like(xd)
    _lib.check(lib.wx_lean_log(xd.data_ptr(), yd.data_ptr(), x.size, torch.cuda.current_stream().cuda_stream), "wx_lean_log")
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    ref = np.log(x.astype(np.longdouble))
    err = np.abs(y.astype(np.longdouble) - ref)
    ulp = np.spacing(np.abs(np.asarray(ref, dtype=np.float64)))
    assert float(np.max(err / np.maximum(ulp, 2.2e-16))) <= 1.0, float(np.max(err / np.maximum(ulp, 2.2e-16)))
    assert y[-5] == 0.0
    # a state that has blown up must show in R (the NaN flag of the integrators): what the library's log returns there
    bad = torch.tensor([-1.0, -1e-300, 0.0, float("nan"), -float("inf")], dtype=torch.float64, device=DEV)
    out = torch.empty_like(bad)
    _lib.check(lib.wx_lean_log(bad.data_ptr(), out.data_ptr(), bad.numel(), torch.cuda.current_stream().cuda_stream), "wx_lean_log")
    o = out.cpu().numpy()
    assert np.isnan(o[0]) and np.isnan(o[1]) and o[2] == -np.inf and np.isnan(o[3]) and np.isnan(o[4])


def test_tile_edge_states_pulled_from_the_neighbour_tiles(monkeypatch):
    """One rank owns the sphere: the one-kernel form forms the tile-edge states itself from the neighbour tiles' nodal values
    (the sender's extrapolation, rotation, flip) - no pack launch.  On the reference's all-panel fixture (order 3, the form
    selected by hand) against the reference's R of every panel - every panel edge's rotation and flip - and against the same
    evaluation through packed edge messages, plain and on the shifted state of the finite-difference products; and at order 2 (the
    default form) on a sphere of own geometry."""
    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd import synthetic
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = golden("euler3d_c31p_n3_h4_v2")
    panels = g.metric_panels()
    assert len(panels) == 6
    Q = torch.stack([to_dev(g.q(p)) for p in panels])
    v = Q * 1e-3 * torch.rand_like(Q)

    def both_ways(make_plans, Q, v):
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("WXHIP_BRICK_PULLS", mode)
            plans = make_plans()
            for pl in plans.values():
                pl.set_one_kernel(True)
            rhs = RhsEuler3D(plans)
            R = rhs(Q)
            bt = rhs._batch_for(torch.float64, rhs.plans_for(torch.float64), rhs.exchange_for(torch.float64))
            assert bt.pulls == (mode == "1")
            S = rhs.shifted_axpy(Q, v, 1e-4, v, 1.0, 0.0, -2.0, R, 2.0)
            torch.cuda.synchronize()
            res[mode] = (R, S)
        assert _rel(res["1"][0], res["0"][0]) <= 1e-12 and _rel(res["1"][1], res["0"][1]) <= 1e-12
        return res["1"][0]

    R = both_ways(lambda: {p: make_plan(g, p) for p in panels}, Q, v).cpu().numpy()
    for i, p in enumerate(panels):
        o = make_oracle(g, p)
        want = {}
        o.rhs(g.q(p), g.halo(p), want=want)
        scale = np.maximum(var_max(g.r(p)), o.cancel_scale(want))
        err = var_err(R[i], g.r(p))
        assert (err <= 1e-10 * scale).all(), (p, err / scale)

    n, H, V = 2, 5, 3
    tiles = [CubedSphere3DTile(n, H, V, p, 10000.0, 31) for p in range(6)]
    metrics = [metric3d_torch(t, DEV) for t in tiles]
    Q2 = torch.stack([torch.from_numpy(initial_state(t)).to(DEV) for t in tiles])
    Q2 = Q2 * (1.0 + 0.01 * (torch.rand_like(Q2) - 0.5))
    both_ways(lambda: {p: Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metrics[p]) for p in range(6)}, Q2, Q2 * 1e-3)


@pytest.mark.parametrize("pulls", ["1", "0"])
def test_one_kernel_form_on_the_24_tile_decomposition(pulls, monkeypatch):
    """The layout of 4 and 8 GPUs - k x k tiles per panel, interior tile edges unrotated and unflipped, panel edges as ever - on
    one rank in the one-kernel form (order 3: selected by hand), tile-edge states pulled from the neighbour tiles or read from
    packed messages: R of every tile against what the reference computed on 24 MPI ranks (1e-10); per-tile launches agree with
    the batched one."""
    from tests.gpu_util import to_dev
    from tests.test_euler3d_gpu import _scale_tile
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    monkeypatch.setenv("WXHIP_BRICK_PULLS", pulls)
    g = golden("euler3d_tiles24_n3_h2_v2")
    k = int(g["meta/k"])
    topo = CubeTopology(k)
    plans = {}
    for t in range(topo.ntiles):
        p, row, col = topo.locate(t)
        tile = CubedSphere3DTile(g.n, g.H, g.V, p, 10000.0, g.case, row=row, col=col, k=k)
        plans[t] = Euler3DPlan(g.n, g.H, g.V, g.case, p, dfr_ops(g.n), metric3d_torch(tile, DEV), on_panel_edge=topo.on_panel_edge(t))
        plans[t].set_one_kernel(True)
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, tiles_per_side=k)
    rhs = RhsEuler3D(plans, ex)
    assert rhs._small_tiles()
    Q = torch.stack([to_dev(g.q(t)) for t in range(topo.ntiles)])
    Rd = rhs(Q)
    assert rhs._batch_for(torch.float64, rhs.plans_for(torch.float64), ex).pulls == (pulls == "1")
    R = Rd.cpu().numpy()
    rhs.batched = False
    assert _rel(rhs(Q), Rd) <= 1e-12
    scales = {t: _scale_tile(g, t, topo) for t in g.metric_panels()}
    floor = np.max(np.stack(list(scales.values())), axis=0)
    for t in range(topo.ntiles):
        ref = g.r(t)
        assert (var_err(R[t], ref) <= 1e-10 * np.maximum(var_max(ref), scales.get(t, floor))).all(), t


@pytest.mark.parametrize("n,H,V", [(2, 5, 3), (2, 4, 2), (3, 4, 2), (4, 3, 2)])
def test_one_kernel_form_of_the_complex_step_product(n, H, V):
    """wx_euler3d_jvp on a dual plan in the one-kernel form (csrc/euler3d_brick_jvp.h) against the two-kernel form on the same plan:
    whole tile and INTERIOR + BOUNDARY, dual edge messages from the pack kernel."""
    from wxfactory_amd import _lib, synthetic
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    m = synthetic.euler3d_metric(n, H, V, 4, DEV, seed=5)
    q = synthetic.euler3d_state(n, H, V, 4, DEV, seed=5)
    v = q * 1e-2 * (torch.rand_like(q) - 0.5)
    plan = Euler3DPlan(n, H, V, 31, 4, synthetic.dfr_ops(n), m, dtype=torch.complex128, dual=True)
    assert plan.one_kernel == (n == 2)
    eps = 1.4901161193847656e-08
    res = {}
    for form in (True, False):
        plan.set_one_kernel(form)
        send = torch.zeros((4, plan.edge_count), dtype=torch.complex128, device=DEV)
        plan.jvp_extrap_pack(q, v, eps, [send[e].data_ptr() for e in range(4)])
        halo = [send[e] for e in (1, 0, 3, 2)]
        hp = [h.data_ptr() for h in halo]
        out = torch.full_like(q, float("nan"))
        plan.jvp(q, v, eps, hp, out, 1.0 / eps)
        split = torch.full_like(q, float("nan"))
        plan.jvp(q, v, eps, None, split, 1.0 / eps, _lib.WX_REGION_INTERIOR)
        plan.jvp(q, v, eps, hp, split, 1.0 / eps, _lib.WX_REGION_BOUNDARY)
        torch.cuda.synchronize()
        assert torch.equal(out, split)
        res[form] = (out, torch.view_as_real(send).clone())
    assert torch.isfinite(res[True][0]).all()
    assert _rel(res[True][1], res[False][1]) <= 1e-13
    assert _rel(res[True][0], res[False][0]) <= 1e-11, _rel(res[True][0], res[False][0])
    plan.close()


def test_complex_step_matvec_and_kiops_on_an_order_2_sphere(monkeypatch):
    """Order 2 (config/dcmip31.ini's) on a small sphere of own geometry: matvec_fun's complex step through the one-kernel dual form
    (batched, tile-edge states pulled or packed) against the two-kernel dual kernels, and KIOPS over it (the one-call Krylov
    vector of wx_euler3d_batch_kiops_vector): the same adaptive decisions."""
    from wxfactory_amd import synthetic
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.matvec import ComplexStepOperator, matvec_fun
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.solvers import kiops

    n, H, V = 2, 6, 3
    tiles = [CubedSphere3DTile(n, H, V, p, 10000.0, 31) for p in range(6)]
    metrics = [metric3d_torch(t, DEV) for t in tiles]
    Q = torch.stack([torch.from_numpy(initial_state(t)).to(DEV) for t in tiles])
    Q = Q * (1.0 + 0.01 * (torch.rand_like(Q) - 0.5))
    v = (torch.rand_like(Q) - 0.5).flatten()
    res = {}
    for mode in ("pull", "pack", "two"):
        monkeypatch.setenv("WXHIP_BRICK_PULLS", "1" if mode == "pull" else "0")
        monkeypatch.setenv("WXHIP_DIRECT", "0" if mode == "two" else "1")
        plans = {p: Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metrics[p]) for p in range(6)}
        rhs = RhsEuler3D(plans)
        R = rhs(Q)
        jv = matvec_fun(v, 30.0, Q, R, rhs, "complex")
        dualplans = rhs._jvp_plans()
        assert all(pl.one_kernel == (mode != "two") for pl in dualplans.values())
        vec = torch.zeros((2, R.numel()), dtype=torch.float64, device=DEV)
        vec[1] = R.flatten()
        op = ComplexStepOperator(30.0, Q, R, rhs)
        phiv, stats = kiops([1], op, vec, tol=1e-7, m_init=1, mmin=10, mmax=64)
        torch.cuda.synchronize()
        res[mode] = (jv, phiv, [int(stats[i]) for i in (0, 1, 2, 3, 5)])
    for mode in ("pull", "pack"):
        assert _rel(res[mode][0].reshape(Q.shape), res["two"][0].reshape(Q.shape)) <= 1e-10, mode
        assert res[mode][2] == res["two"][2], (mode, res[mode][2], res["two"][2])
        assert _rel(res[mode][1].reshape(-1, 1), res["two"][1].reshape(-1, 1)) <= 1e-6
