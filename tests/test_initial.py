"""Initial states (init/dcmip.py restated in wxfactory_amd/initial.py) against the UNPERTURBED states the
reference's init_state_vars produced for the fixtures (cases 31 and 21, every panel)."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN
from wxfactory_amd.geometry3d import CubedSphere3DTile, planet_for_case, topography_for_case
from wxfactory_amd.initial import initial_state


@pytest.mark.parametrize("name,ztop", [("euler3d_c31_n3_h4_v2", 10000.0), ("euler3d_c31_n8_h2_v2", 10000.0),
                                       ("euler3d_c21_n4_h3_v4", 30000.0)])
def test_initial_state_matches_reference(name, ztop):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, H, V, case = (int(g[f"meta/{k}"]) for k in ("n", "H", "V", "case_number"))
    topo = topography_for_case(case, planet_for_case(case)[0])
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, ztop, case, topo=topo)
        Q, ref = initial_state(t), g[f"p{p}/Q"]
        assert Q.shape == ref.shape
        scale = np.abs(ref).max(axis=(1, 2, 3, 4))
        scale[3] = scale[1]  # rho*w is identically zero in both cases
        err = np.abs(Q - ref).max(axis=(1, 2, 3, 4))
        assert (err <= 1e-12 * scale).all(), (p, err / scale)


def test_unsupported_case():
    with pytest.raises(ValueError):
        initial_state(CubedSphere3DTile(3, 2, 2, 0, 10000.0, 77))
