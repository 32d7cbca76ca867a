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
