"""The C++/OpenMP CPU restatement (oracle/c/euler3d_port.cpp, the cpu_baseline of bench.py) against the golden
vectors produced by the reference itself, with the bound and norm of tests/test_oracle_euler3d.py."""
import numpy as np
import pytest

from oracle import cubed_sphere as cs
from oracle.c_port import Euler3DPortC
from tests.util import tight_tangent, EULER_FIXTURES, MONOLITH_FIXTURES, golden, make_oracle, var_err, var_max

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


@pytest.mark.parametrize("name", ["euler3d_c31p_n3_h4_v2", "euler3d_c31p_n8_h2_v2", "euler3d_c21_n4_h3_v4", "euler3d_c31p_n5_h2_v1",
                                  "euler3d_c21p_n4_h3_v4", "euler3d_c21p_n8_h2_v2"])
def test_port_complex_step_matches_reference(name):
    """The complex128 instantiation against R(Q + i eps V) of the reference (solvers/matvec.py:56-61 on rhs_dfr.py):
    faces and routed halos, real part at the 1e-10 bound, tangent Im R at 1e-10 of its own size."""
    g = golden(name)
    for p in g.metric_panels():   # (the perturbation V is stored for these panels only)
        o = make_port(g, p)
        sends = o.pack_edges(o.extrapolate(g.q(p, True)))
        for e, got in enumerate(sends):
            ref = g.halo(cs.NEIGHBOR[p][e], True)[cs.landing_edge(p, e)]
            assert np.abs(got.real - ref.real).max() <= 1e-13 * np.abs(ref.real).max(), (p, e)
            assert np.abs(got.imag - ref.imag).max() <= 1e-12 * np.abs(ref.imag).max(), (p, e)
    tight = tight_tangent(name)   # see tests/test_oracle_euler3d.py on numpy.maximum's tie-break on symmetric states
    for p in g.metric_panels():
        o, ref_o = make_port(g, p), make_oracle(g, p)
        want = {}
        ref_o.rhs(g.q(p, True), g.halo(p, True), want=want)
        R = o.rhs(g.q(p, True), g.halo(p, True))
        ref = g.r(p, True)
        s = ref_o.cancel_scale(want)
        assert (var_err(R.real, ref.real) <= TOL * np.maximum(var_max(ref.real), s)).all(), (name, p)
        ierr = var_err(R.imag, ref.imag) / np.maximum(var_max(ref.imag), g.eps * s * 1e-3)
        assert (ierr <= (1e-10 if tight else 1e-3)).all(), (name, p, ierr)


def test_thread_count_does_not_change_the_result():
    g = golden("euler3d_c31p_n8_h2_v2")
    p = g.metric_panels()[0]
    a = make_port(g, p, threads=1).rhs(g.q(p), g.halo(p))
    b = make_port(g, p, threads=4).rhs(g.q(p), g.halo(p))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["euler3d_c31p_n8_h2_v2", "euler3d_c21p_n4_h3_v4", "euler3d_c31p_n5_h2_v1", "euler3d_c31p_n2_h4_v3"])
def test_vectorised_float64_path_equals_the_generic_one(name):
    """The float64 entry points run a path written for the vector units (structure-of-arrays scratch, `omp simd` loops,
    exp / log through libmvec: <= 4 ulp) - what bench.py times as the CPU baseline.  Same arithmetic as the scalar,
    type-generic instantiation that the complex entry points use: faces and R agree to rounding - 1e-12 of the cancellation scale:
    which elements fall into the scalar remainder loops depends on the alignment of the arrays - (and both are held to the
    reference's values above)."""
    g = golden(name)
    for p in g.metric_panels():
        fast, slow = make_port(g, p), make_port(g, p)
        slow.generic = True
        fa, fb = fast.extrapolate(g.q(p)), slow.extrapolate(g.q(p))
        for a, b in zip(fa, fb):
            assert np.abs(a - b).max() <= 1e-14 * np.abs(b).max()
        Ra, Rb = fast.rhs(g.q(p), g.halo(p), itf=fa), slow.rhs(g.q(p), g.halo(p), itf=fb)
        want = {}
        make_oracle(g, p).rhs(g.q(p), g.halo(p), want=want)
        scale = np.maximum(var_max(Rb), make_oracle(g, p).cancel_scale(want))
        assert (var_err(Ra, Rb) <= 1e-12 * scale).all(), (name, p, var_err(Ra, Rb) / scale)
