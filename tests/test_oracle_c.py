"""The C++/OpenMP CPU restatement (oracle/c/euler3d_port.cpp, the cpu_baseline of bench.py) against the golden
vectors produced by the reference itself, with the bound and norm of tests/test_oracle_euler3d.py."""
import numpy as np
import pytest

from oracle import cubed_sphere as cs
from oracle.c_port import Euler3DPortC
from tests.util import EULER_FIXTURES, MONOLITH_FIXTURES, golden, make_oracle, var_err, var_max

TOL = 1e-10


def make_port(g, p, threads=2):
    return Euler3DPortC(g.n, g.H, g.V, g.case, g.ops, g.metric(p), g[f"p{p}/geom/boundary_sn_new"],
                        g[f"p{p}/geom/boundary_we_new"], panel=p, threads=threads)


@pytest.mark.parametrize("name", EULER_FIXTURES)
def test_port_faces_and_exchange(name):
    g = golden(name)
    sends = []
    for p in range(6):
        o = make_port(g, p)
        sends.append(o.pack_edges(o.extrapolate(g.q(p))))
    recvs = cs.route(sends)
    for p in range(6):
        for e, got in enumerate(recvs[p]):
            ref = g.halo(p)[e]
            assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (p, e)


@pytest.mark.parametrize("name", EULER_FIXTURES + MONOLITH_FIXTURES)
def test_port_rhs_matches_reference(name):
    g = golden(name)
    for p in g.metric_panels():
        o, ref_o = make_port(g, p), make_oracle(g, p)
        want = {}
        ref_o.rhs(g.q(p), g.halo(p), want=want)   # only for the cancellation scale s_v of the norm
        R = o.rhs(g.q(p), g.halo(p))
        ref = g.r(p)
        scale = np.maximum(var_max(ref), ref_o.cancel_scale(want))
        err = var_err(R, ref)
        assert (err <= TOL * scale).all(), (name, p, err / scale)


def test_thread_count_does_not_change_the_result():
    g = golden("euler3d_c31p_n8_h2_v2")
    p = g.metric_panels()[0]
    a = make_port(g, p, threads=1).rhs(g.q(p), g.halo(p))
    b = make_port(g, p, threads=4).rhs(g.q(p), g.halo(p))
    assert np.array_equal(a, b)
