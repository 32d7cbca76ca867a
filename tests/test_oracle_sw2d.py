"""Shallow-water CPU oracle (oracle/sw2d.py) against golden vectors produced by the reference
(rhs_sw.py on 6 emulated ranks: Williamson cases 6, 5 (topography) and 2)."""
import numpy as np
import pytest

from oracle import cubed_sphere as cs
from tests.util import SW_FIXTURES, golden_sw, make_sw_oracle, var_err, var_max

TOL = 1e-10


def test_kronecker_meaning_2d():
    for name in SW_FIXTURES:
        g = golden_sw(name)
        o = make_sw_oracle(g, 0)
        u, f = g["kron/u"], g["kron/f"]
        for d, nm in enumerate(("derivative_x", "derivative_y")):
            np.testing.assert_allclose(o.deriv(u, d), g["kron/" + nm], rtol=0, atol=2e-13 * g.n**2)
        for d, nm in enumerate(("extrap_x", "extrap_y")):
            np.testing.assert_allclose(o.extrap(u, d), g["kron/" + nm], rtol=0, atol=1e-14 * g.n)
        for d, nm in enumerate(("correction_WE", "correction_SN")):
            np.testing.assert_allclose(o.correct(f, d), g["kron/" + nm], rtol=0, atol=1e-13 * g.n**2)


@pytest.mark.parametrize("name", SW_FIXTURES)
def test_exchange(name):
    g = golden_sw(name)
    sends = []
    for p in range(6):
        o = make_sw_oracle(g, p)
        sends.append(o.pack_edges(o.extrapolate(g.q(p))))
    recvs = cs.route(sends)
    for p in range(6):
        for e in range(4):
            ref = g.halo(p)[e]
            assert np.abs(recvs[p][e] - ref).max() <= 1e-14 * np.abs(ref).max(), (p, e)


@pytest.mark.parametrize("name", SW_FIXTURES)
@pytest.mark.parametrize("cplx", [False, True])
def test_rhs_matches_reference(name, cplx):
    g = golden_sw(name)
    for p in range(6):
        o = make_sw_oracle(g, p)
        want = {}
        R = o.rhs(g.q(p, cplx), g.halo(p, cplx), want=want)
        ref = g.r(p, cplx)
        scale = np.maximum(var_max(ref.real), o.cancel_scale(want))
        assert (var_err(R.real, ref.real) <= TOL * scale).all()
        if cplx:
            assert (var_err(R.imag, ref.imag) <= 1e-10 * var_max(ref.imag)).all()


def test_oracle_on_tiles_of_a_24_rank_run():
    """The restatement against the reference run on 24 MPI ranks (2 x 2 tiles per panel, mountain case): every
    tile's R from its own metric / topography arrays and the halos the reference delivered."""
    from oracle.sw2d import SW2DOracle
    from tests.util import golden_sw, var_err, var_max

    g = golden_sw("sw_tiles24_c5_n4_h2")
    assert int(g["meta/k"]) == 2 and g.H == 2
    for t in range(24):
        p = int(g[f"p{t}/tile/panel_row_col"][0])
        o = SW2DOracle(g.n, g.H, g.ops, g.sub(t, "metric"), g.sub(t, "topo"), g[f"p{t}/geom/boundary_sn"],
                       g[f"p{t}/geom/boundary_we"], panel=p)
        ref = g.r(t)
        assert (var_err(o.rhs(g.q(t), g.halo(t)), ref) <= 1e-12 * var_max(ref)).all(), t  # (R is a small difference of large terms)
