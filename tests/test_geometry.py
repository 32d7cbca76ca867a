"""Setup-time geometry/metric restatement (wxfactory_amd/geometry.py) against the metric arrays the
reference built (geometry/cubed_sphere_2d.py + metric2d.py) for the shallow-water fixtures: rotated grid
(phi0 = pi/4), all six panels, n = 4, 5, 8."""
import numpy as np
import pytest

from tests.util import SW_FIXTURES, golden_sw


@pytest.mark.parametrize("name", SW_FIXTURES)
def test_metric2d_matches_reference(name):
    from wxfactory_amd.geometry import CubedSphereTile2D, metric2d

    g = golden_sw(name)
    lam, phi, alp = (float(x) for x in g["meta/grid_rotation"])
    for p in range(6):
        tile = CubedSphereTile2D(g.n, g.H, p, lambda0=lam, phi0=phi, alpha0=alp)
        m = metric2d(tile)
        ref = g.sub(p, "metric")
        np.testing.assert_allclose(tile.boundary_sn, g[f"p{p}/geom/boundary_sn"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(tile.boundary_we, g[f"p{p}/geom/boundary_we"], rtol=1e-14, atol=1e-15)
        for k, v in m.items():
            if k.startswith("boundary"):
                continue
            r = ref[k]
            assert v.shape == r.shape, (k, v.shape, r.shape)
            scale = np.abs(r).max()
            assert np.abs(v - r).max() <= 1e-13 * scale + 1e-300, (name, p, k, np.abs(v - r).max() / scale)


def test_solution_points_and_operators_match_reference():
    from wxfactory_amd.geometry import gauss_legendre
    from wxfactory_amd.synthetic import dfr_ops

    for name in SW_FIXTURES:
        g = golden_sw(name)
        x, w = gauss_legendre(g.n)
        np.testing.assert_allclose(x, g.ops["solution_points"], rtol=0, atol=1e-14)
        np.testing.assert_allclose(w, g.ops["glweights"], rtol=0, atol=1e-14)
        o = dfr_ops(g.n)
        for k in ("extrap_neg", "extrap_pos", "diff_solpt", "correction", "highfilter"):
            assert np.abs(o[k] - g.ops[k]).max() <= 2e-13 * max(1.0, np.abs(g.ops[k]).max()), (name, k)
