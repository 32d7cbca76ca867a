"""Every templated order n = 2..8 of the HIP kernels against the (reference-pinned) CPU oracles on seeded
synthetic tiles - the golden fixtures cover n = 3, 4, 5, 8; this closes the gap for the other
instantiations, for ragged sizes (H = 1, 2: every element touches several tile edges; V = 1: both
vertical walls in one element) and for the region split."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _euler_case(n, H, V, panel, case=31, seed=11):
    from oracle.euler3d import Euler3DOracle
    from wxfactory_amd import synthetic

    ops = synthetic.dfr_ops(n)
    m = synthetic.euler3d_metric(n, H, V, panel, "cpu", seed=seed, damping=case in (21, 22))
    q = synthetic.euler3d_state(n, H, V, panel, "cpu", seed=seed)
    om = {"sqrtG_new": m["sqrtG"].numpy(), "inv_sqrtG_new": (1.0 / m["sqrtG"]).numpy(),
          "h_contra_new": m["h_contra"].numpy(), "christoffel": m["christoffel"].numpy(),
          "inv_dzdeta_new": m["inv_dzdeta"].numpy()}
    for d in "ijk":
        om[f"sqrtG_itf_{d}_new"] = m[f"sqrtG_itf_{d}"].numpy()
        om[f"h_contra_itf_{d}_new"] = m[f"h_contra_itf_{d}"].numpy()
    if case in (21, 22):
        om["damp_coef"], om["damp_uref"] = m["damp_coef"].numpy(), m["damp_uref"].numpy()
    bsn = np.tile(m["boundary_sn"].numpy().reshape(H, 1, n), (1, n, 1))
    o = Euler3DOracle(n, H, V, case, ops, om, bsn, bsn, panel=panel)
    return ops, m, q, o


@pytest.mark.parametrize("n,H,V,panel,case", [(2, 3, 2, 1, 31), (5, 3, 2, 2, 31), (6, 2, 2, 5, 21), (7, 2, 1, 3, 31),
                                               (8, 1, 1, 4, 31), (4, 5, 3, 0, 22), (3, 2, 1, 5, 12)])
def test_euler3d_all_orders(n, H, V, panel, case, built_lib):
    from wxfactory_amd import _lib
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    ops, m, q, o = _euler_case(n, H, V, panel, case)
    itf = o.extrapolate(q.numpy())
    sends = o.pack_edges(itf)
    halo = [sends[1], sends[0], sends[3], sends[2]]
    want = {}
    ref = o.rhs(q.numpy(), halo, itf=itf, want=want)
    plan = Euler3DPlan(n, H, V, case, panel, ops, {k: v.to(DEV) for k, v in m.items()})
    qd = q.to(DEV)
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
    plan.extrap_pack(qd, list(send))
    hd = [torch.from_numpy(np.ascontiguousarray(h)).to(DEV) for h in halo]
    out = torch.full_like(qd, float("nan"))
    if H > 2:
        plan.rhs(qd, None, out, _lib.WX_REGION_INTERIOR)
        plan.rhs(qd, hd, out, _lib.WX_REGION_BOUNDARY)
    else:
        plan.rhs(qd, hd, out, _lib.WX_REGION_ALL)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    for e in range(4):
        assert np.abs(send[e].cpu().numpy().reshape(sends[e].shape) - sends[e]).max() <= 1e-13 * np.abs(sends[e]).max()
    if case < 13:  # advection-only cases: the reference zeroes every row (rhs_dfr.py:307-313)
        assert np.abs(ref).max() == 0 and np.abs(got).max() == 0
        return
    ax = (1, 2, 3, 4)
    scale = np.maximum(np.abs(ref).max(axis=ax), o.cancel_scale(want))
    assert (np.abs(got - ref).max(axis=ax) <= 1e-10 * scale).all(), np.abs(got - ref).max(axis=ax) / scale


@pytest.mark.parametrize("n,H,panel,topo", [(2, 3, 1, False), (3, 2, 2, True), (6, 3, 5, False), (7, 1, 3, True), (8, 4, 0, True)])
def test_sw_all_orders(n, H, panel, topo, built_lib):
    from oracle.sw2d import SW2DOracle
    from wxfactory_amd import synthetic
    from wxfactory_amd.rhs_sw import SwPlan

    ops = synthetic.dfr_ops(n)
    m = synthetic.sw_metric(n, H, panel, "cpu", seed=5, topo=topo)
    q = synthetic.sw_state(n, H, panel, "cpu", seed=5)
    om = {k: v.numpy() for k, v in m.items()}
    tp = {k: om[k] for k in ("hsurf", "dzdx1", "dzdx2", "hsurf_itf_i", "hsurf_itf_j")} if topo else None
    o = SW2DOracle(n, H, ops, om, tp, om["boundary_sn"], om["boundary_we"], panel=panel)
    itf = o.extrapolate(q.numpy())
    sends = o.pack_edges(itf)
    halo = [sends[1], sends[0], sends[3], sends[2]]
    want = {}
    ref = o.rhs(q.numpy(), halo, itf=itf, want=want)
    plan = SwPlan(n, H, panel, ops, {k: v.to(DEV) for k, v in m.items()})
    qd = q.to(DEV)
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
    plan.extrap_pack(qd, list(send))
    hd = [torch.from_numpy(np.ascontiguousarray(h)).to(DEV) for h in halo]
    out = torch.full_like(qd, float("nan"))
    plan.rhs(qd, hd, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for e in range(4):
        assert np.abs(send[e].cpu().numpy().reshape(sends[e].shape) - sends[e]).max() <= 1e-13 * np.abs(sends[e]).max()
    ax = (1, 2, 3)
    scale = np.maximum(np.abs(ref).max(axis=ax), o.cancel_scale(want))
    assert (np.abs(got - ref).max(axis=ax) <= 1e-10 * scale).all(), np.abs(got - ref).max(axis=ax) / scale


@pytest.mark.parametrize("n,H,V", [(8, 60, 8), (5, 12, 3)])
def test_mass_conservation_at_full_size(n, H, V):
    """Size-independent property at the BENCHMARK's size (E7: n = 8, 60 x 60 x 8 elements per panel, 442 M DOF):
    the scheme is conservative, so the quadrature of sqrtG * R[rho] over the closed sphere telescopes to zero -
    every interface flux, including the rotated / flipped ones across panel edges, must be the same number seen
    from both sides.  Own geometry, own initial state (DCMIP 3-1) plus a seeded perturbation."""
    import numpy as np

    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    dev = "cuda:0"
    w1 = np.polynomial.legendre.leggauss(n)[1]
    w3 = torch.from_numpy(np.einsum("k,j,i->kji", w1, w1, w1).reshape(-1)).to(dev)
    plans, qs, sg = {}, {}, {}
    gen = torch.Generator(device=dev).manual_seed(11)
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
        m = metric3d_torch(t, dev)
        sg[p] = m["sqrtG"]
        plans[p] = Euler3DPlan(n, H, V, 31, p, dfr_ops(n), m)
        q = torch.from_numpy(initial_state(t)).to(dev)
        qs[p] = q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5))
    R = RhsEuler3D(plans)(qs)
    total, gross = 0.0, 0.0
    for p in range(6):
        dens = sg[p] * R[p][0] * w3
        total += float(dens.sum())
        gross += float(dens.abs().sum())
    assert gross > 0 and abs(total) <= 1e-11 * gross, (total, gross)

    # ... and so one pipelined SSP-RK3 step (fused stage updates, faces written by the stage kernels) keeps the
    # total mass:  sum w sqrtG rho  before == after, to rounding
    from wxfactory_amd.integrators import Tvdrk3

    Q = torch.stack([qs[p] for p in range(6)])
    SG = torch.stack([sg[p] for p in range(6)])
    mass = lambda X: float((SG * X[:, 0] * w3).sum())  # noqa: E731
    stepper = Tvdrk3(RhsEuler3D(plans))
    assert stepper.pipeline
    Q1 = stepper.step(Q, 0.02)
    Q2 = stepper.step(Q1, 0.02)  # second step starts from faces prepared by the first
    m0, m2 = mass(Q), mass(Q2)
    assert float((Q2 - Q).abs().max()) > 0 and abs(m2 - m0) <= 1e-12 * abs(m0), (m0, m2)


def test_shallow_water_mass_conservation_at_s7():
    """The S7 workload of BASELINE.json's galewsky line at FULL size (shallow water, n = 8, 60 x 60 elements per panel,
    6 panels, 4.15 M DOF) through its size-independent property: the continuity row is in flux form, so the quadrature
    of sqrtG * R[h] over the closed sphere telescopes to zero - every AUSM interface flux, across rotated and flipped
    panel edges too, is the same number from both sides.  Own cubed-sphere metric, seeded state; one evaluation with
    all panels in one launch per phase and one through the per-panel launches."""
    import numpy as np

    from wxfactory_amd import synthetic
    from wxfactory_amd.geometry import CubedSphereTile2D, metric2d_torch
    from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan

    dev = "cuda:0"
    n, H = 8, 60
    w1 = np.polynomial.legendre.leggauss(n)[1]
    w2 = torch.from_numpy(np.einsum("j,i->ji", w1, w1).reshape(-1)).to(dev)
    plans, sg = {}, []
    for p in range(6):
        m = metric2d_torch(CubedSphereTile2D(n, H, p, phi0=0.7853981633974483), dev)
        sg.append(m["sqrtG"])
        plans[p] = SwPlan(n, H, p, synthetic.dfr_ops(n), m)
    Q = torch.stack([synthetic.sw_state(n, H, p, dev, 5) for p in range(6)])
    rhs = RhsShallowWater(plans)
    SG = torch.stack(sg).reshape(6, H, H, n * n)
    for batched in (True, False):
        rhs.batched = batched
        R = rhs(Q)
        dens = SG * R[:, 0].reshape(6, H, H, n * n) * w2
        total, gross = float(dens.sum()), float(dens.abs().sum())
        assert gross > 0 and abs(total) <= 1e-11 * gross, (batched, total, gross)


def test_checkpoint_round_trip_from_device_tensors(tmp_path):
    """SURVEY 8f-4 on the GPU: the stacked states of the 24-tile layout (device tensors) -> global array ->
    the reference's file format -> back to device tiles, bit for bit; the global array is the same one the
    6-tile layout of the same sphere produces (rank-count independence, tests/unit/restart/test_restart.py:107-151)."""
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.state import distribute_cube, gather_cube, load_state, save_state

    dev = "cuda:0"
    n3, V, H = 27, 2, 4
    gen = torch.Generator(device=dev).manual_seed(3)
    glob = torch.rand((6, 5, V, H, H, n3), generator=gen, device=dev, dtype=torch.float64)
    tiles24 = distribute_cube(glob, tiles_per_side=2, device=dev)
    assert tiles24.is_cuda and tiles24.shape == (24, 5, V, H // 2, H // 2, n3)
    topo = CubeTopology(2)
    p, r, c = topo.locate(13)
    assert torch.equal(tiles24[13], glob[p, :, :, r * 2:(r + 1) * 2, c * 2:(c + 1) * 2, :])
    back = gather_cube(tiles24, tiles_per_side=2)
    assert back.is_cuda and torch.equal(back, glob)
    assert torch.equal(gather_cube(distribute_cube(glob, tiles_per_side=1, device=dev)), glob)
    f = tmp_path / "state.npy"
    save_state(back, "v", "[General]\nequations = Euler\n", str(f))
    state, version, config = load_state(str(f))
    assert version == "v" and "equations = Euler" in config
    again = distribute_cube(state, tiles_per_side=2, device=dev)
    assert again.is_cuda and torch.equal(again, tiles24)


@pytest.mark.parametrize("n,H,V", [(3, 1, 1), (4, 2, 1), (8, 1, 2)])
def test_prepared_jvp_on_ragged_tiles(n, H, V):
    """The prepared complex-step JVP (face values cached, tangents per product) on the smallest tiles - every lateral
    face a halo (H = 1), every vertical face a wall (V = 1) - equals the unprepared one bit for bit."""
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    dev = "cuda:0"
    plans, qs = {}, []
    gen = torch.Generator(device=dev).manual_seed(5)
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
        plans[p] = Euler3DPlan(n, H, V, 31, p, dfr_ops(n), metric3d_torch(t, dev))
        q = torch.from_numpy(np.array(initial_state(t))).to(dev)
        qs.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5)))
    Q = torch.stack(qs)
    rhs = RhsEuler3D(plans)
    rhs.batched = False   # per-tile launches: the path the prepared JVP lives on
    R = rhs(Q)
    v = (torch.rand(Q.shape, generator=gen, device=dev, dtype=Q.dtype) - 0.5) * Q.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
    plain = matvec_fun(v.flatten(), 2.0, Q, R, rhs, "complex")
    assert rhs.jvp_prepare(Q)
    prepared = matvec_fun(v.flatten(), 2.0, Q, R, rhs, "complex")
    rhs.jvp_release()
    assert torch.isfinite(plain).all() and float(plain.abs().max()) > 0
    assert torch.equal(plain, prepared)


@pytest.fixture(scope="module")
def one_rank_comm():
    """one communicator for the module (a one-rank RCCL communicator needs no process group)"""
    from wxfactory_amd.exchange import RcclComm

    c = RcclComm(0, 1, device="cuda:0")
    yield c
    torch.cuda.synchronize()
    c.close()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("n,H,V,column", [(2, 5, 3, False), (3, 4, 2, False), (4, 3, 2, False), (5, 3, 1, False), (6, 3, 2, False),
                                          (7, 3, 1, False), (8, 3, 2, False), (8, 3, 2, True), (5, 2, 2, True), (8, 1, 2, False)])
def test_partial_products_of_the_jvp_store_fill_exactly_their_slots(n, H, V, column, overlap, one_rank_comm):
    """The a x + b z store of the prepared JVP with the stored vector's products (KIOPS on long vectors): every launch writes
    one pair of partial sums per workgroup, wx_euler3d_jvp_workgroups of them - for every order (one or several elements per
    workgroup, 1-D and 3-D grids), whole-tile and INTERIOR / BOUNDARY launches, general and column kernels: exactly the
    counted slots are written (a wrong count would write past the buffer), and they sum to the products."""
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.matvec import ComplexStepOperator
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    dev = "cuda:0"
    plans, qs = {}, []
    gen = torch.Generator(device=dev).manual_seed(11)
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
        plans[p] = Euler3DPlan(n, H, V, 31, p, dfr_ops(n), metric3d_torch(t, dev), column_metric="auto" if column else False)
        assert plans[p].column_metric == column
        q = torch.from_numpy(np.array(initial_state(t))).to(dev)
        qs.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5)))
    Q = torch.stack(qs)
    if overlap:   # every edge message through a one-rank RCCL communicator: INTERIOR / BOUNDARY launches, as on several GPUs
        from wxfactory_amd.exchange import PanelExchange

        ex = PanelExchange(plans[0].edge_count, dev, rank=0, world_size=1, loopback=True, backend="rccl", comm=one_rank_comm)
        rhs = RhsEuler3D(plans, ex, overlap=True)
    else:
        rhs = RhsEuler3D(plans)
    rhs.batched = False
    R = rhs(Q)
    v = ((torch.rand(Q.shape, generator=gen, device=dev, dtype=Q.dtype) - 0.5) * Q.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)).flatten()
    op = ComplexStepOperator(2.0, Q, R, rhs)
    assert rhs.jvp_fuses_store(Q)
    plain = op(v)
    z = R.flatten().contiguous()
    rows = [Q.flatten().contiguous(), v.contiguous()]
    coef = torch.tensor([0.5, 2.0], dtype=torch.float64, device=dev)
    out = torch.empty_like(z)
    part, count = op.axpy_into(v, out, z, coef[0:1].data_ptr(), coef[1:2].data_ptr(), rows)   # (allocates the buffer)
    part.fill_(float("nan"))
    part2, count2 = op.axpy_into(v, out, z, coef[0:1].data_ptr(), coef[1:2].data_ptr(), rows)
    torch.cuda.synchronize()
    rhs.jvp_release()
    assert part2 is part and count2 == count and 0 < 2 * count <= part.numel()
    assert torch.isfinite(part[: 2 * count]).all() and torch.isnan(part[2 * count:]).all()
    want = 0.5 * plain + 2.0 * z
    assert float((out - want).abs().max()) <= 1e-14 * float(want.abs().max())
    got = part[: 2 * count].view(count, 2).sum(dim=0)
    for k in range(2):
        assert abs(float(got[k]) - float(torch.dot(rows[k], out))) <= 1e-12 * float(rows[k].norm() * out.norm()), k


@pytest.mark.parametrize("n,Htot,V", [(4, 4, 2), (8, 4, 1)])
def test_result_does_not_depend_on_the_decomposition(n, Htot, V):
    """The reference's restart tests require a 6-rank and a 24-rank run of the same case to agree to 1e-15
    (tests/unit/restart/test_restart.py:107-151): the RHS, and a time loop built on it, must not depend on how the
    sphere is cut into tiles.  Same global state through 6 whole panels and through 24 tiles (interior tile edges
    exchange unrotated / unflipped, panel edges as ever; own geometry per tile): R(Q), the complex-step matvec and three
    pipelined SSP-RK3 steps, compared after gathering into the global layout of the checkpoint format."""
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.state import distribute_cube, gather_cube
    from wxfactory_amd.synthetic import dfr_ops

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(77)
    runs = {}
    glob = None
    for k in (1, 2):
        topo = CubeTopology(k)
        Ht = Htot // k
        plans = {}
        for t in range(topo.ntiles):
            p, row, col = topo.locate(t)
            tile = CubedSphere3DTile(n, Ht, V, p, 10000.0, 31, row=row, col=col, k=k)
            plans[t] = Euler3DPlan(n, Ht, V, 31, p, dfr_ops(n), metric3d_torch(tile, dev), on_panel_edge=topo.on_panel_edge(t))
            if k == 1 and glob is None:
                pass
        if glob is None:   # the global state, once, from the whole-panel tiles
            qs = []
            for p in range(6):
                q = torch.from_numpy(np.array(initial_state(CubedSphere3DTile(n, Htot, V, p, 10000.0, 31)))).to(dev)
                qs.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5)))
            glob = torch.stack(qs)
            vglob = (torch.rand(glob.shape, generator=gen, device=dev, dtype=glob.dtype) - 0.5) * \
                glob.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True) * 1e-3
        ex = PanelExchange(plans[0].edge_count, dev, rank=0, world_size=1, tiles_per_side=k)
        rhs = RhsEuler3D(plans, ex)
        Q = distribute_cube(glob, tiles_per_side=k, device=dev)
        v = distribute_cube(vglob, tiles_per_side=k, device=dev)
        R = rhs(Q)
        J = matvec_fun(v.flatten(), 2.0, Q, R, rhs, "complex").reshape(Q.shape)
        stepper = Tvdrk3(rhs)
        Qn = Q
        for _ in range(3):
            Qn = stepper.step(Qn, 0.02)
        runs[k] = tuple(gather_cube(x.contiguous(), tiles_per_side=k) for x in (R, J, Qn))
    for a, b, what in zip(runs[1], runs[2], ("R", "J v", "Q after 3 steps")):
        scale = a.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
        err = ((a - b).abs() / scale).amax(dim=(0, 2, 3, 4, 5))
        assert float(a.abs().max()) > 0 and (err <= 1e-13).all(), (what, err)


def test_fifty_filtered_steps_at_benchmark_resolution_conserve_mass():
    """A time loop of the benchmark's horizontal resolution (n = 8, 60 x 60 elements per panel, one vertical element,
    55 M DOF): fifty pipelined SSP-RK3 steps, each followed by the exponential filter fused into the last stage, NaN flag
    raised in the same kernels.  Properties that do not depend on the size: the total mass  sum w sqrtG rho  stays put to
    rounding (flux form + a filter that leaves the element mean alone), the state stays finite, the flag stays down."""
    from wxfactory_amd.filters import ExpFilter3D, NanFlag, make_filter
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.integrators import StepLoop, Tvdrk3
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    dev = "cuda:0"
    n, H, V = 8, 60, 1
    w1 = np.polynomial.legendre.leggauss(n)[1]
    w3 = torch.from_numpy(np.einsum("k,j,i->kji", w1, w1, w1).reshape(-1)).to(dev)
    plans, qs, sg = {}, [], []
    gen = torch.Generator(device=dev).manual_seed(5)
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
        m = metric3d_torch(t, dev)
        sg.append(m["sqrtG"])
        plans[p] = Euler3DPlan(n, H, V, 31, p, dfr_ops(n), m)
        q = torch.from_numpy(np.array(initial_state(t))).to(dev)
        qs.append(q * (1.0 + 0.005 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5)))
    Q = torch.stack(qs)
    SG = torch.stack(sg)
    rhs = RhsEuler3D(plans)
    rhs.batched = False   # per-panel launches: the stage kernels with prepared faces and the fused filter
    flag = NanFlag(dev)
    F = make_filter(1e-3, 4, 0.5, np.polynomial.legendre.leggauss(n)[0])
    loop = StepLoop(Tvdrk3(rhs), ExpFilter3D(F, sg), flag, check_every=10)
    assert loop.fused
    mass = lambda X: float((SG * X[:, 0] * w3).sum())  # noqa: E731
    m0 = mass(Q)
    Qn = loop.run(Q, 0.01, 50)
    flag.raise_if_set()
    assert torch.isfinite(Qn).all() and float((Qn - Q).abs().max()) > 0
    assert abs(mass(Qn) - m0) <= 1e-11 * abs(m0), (m0, mass(Qn))
