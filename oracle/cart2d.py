"""CPU oracle: 2-D Cartesian Euler RHS (the reference's plumbing case), NumPy.

TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this; the product path never does.  Parity status: PINNED against golden vectors produced by
running the reference (its Python RHS + its own native pde_cpp kernels), tests/test_oracle_cart2d.py.

Restates reference wx_factory/rhs/rhs_dfr.py:8-45 (RHSDirecFluxReconstruction),
pde/pde_euler_cartesian.py:24-48 and the native kernels it calls:
  pointwise()  pde/kernels/pointwise_flux.hpp:3-31
  riemann()    pde/kernels/riemann_flux.hpp:5-80 (AUSM) + boundary_flux.hpp:3-24 (solid walls),
               loop structure of pde/interface.cpp:127-238
Layout: q (4, nz, nx, n^2), point p = kl*n + il; interface (4, nz, nx, 2n) [:n] minus, [n:] plus.
"""
import numpy

gravity = 9.80616
p0 = 100000.0
Rd = 287.05
cpd = 1005.46
cvd = cpd - Rd
heat_capacity_ratio = cpd / cvd


def pointwise(q):
    rho, ru, rw, rt = q
    inv = 1.0 / rho
    u, w = ru * inv, rw * inv
    p = p0 * numpy.exp(heat_capacity_ratio * numpy.log((Rd * (1.0 / p0)) * rt))
    f1 = numpy.stack([ru, ru * u + p, ru * w, rt * u])
    f3 = numpy.stack([rw, rw * u, rw * w + p, rt * w])
    return f1, f3


def _ausm(qL, qR, d):
    invL, invR = 1.0 / qL[0], 1.0 / qR[0]
    uL, wL, uR, wR = qL[1] * invL, qL[2] * invL, qR[1] * invR, qR[2] * invR
    pL = p0 * numpy.power(qL[3] * Rd * (1.0 / p0), heat_capacity_ratio)
    pR = p0 * numpy.power(qR[3] * Rd * (1.0 / p0), heat_capacity_ratio)
    aL = numpy.sqrt(heat_capacity_ratio * pL * invL)
    aR = numpy.sqrt(heat_capacity_ratio * pR * invR)
    vL, vR = (uL, uR) if d == 0 else (wL, wR)
    ML = vL / aL + 1.0
    MR = vR / aR - 1.0
    M = 0.25 * (ML * ML - MR * MR)
    Mmax = numpy.maximum(0.0, M) * aL
    Mmin = numpy.minimum(0.0, M) * aR
    f = numpy.empty_like(qL)
    f[0] = qL[0] * Mmax + qR[0] * Mmin
    pf = 0.5 * (ML * pL - MR * pR)
    if d == 0:
        f[1] = pf
        f[2] = qL[2] * Mmax + qR[2] * Mmin
    else:
        f[1] = qL[1] * Mmax + qR[1] * Mmin
        f[2] = pf
    f[3] = qL[3] * Mmax + qR[3] * Mmin
    return f


def riemann(qi1, qi3, n):
    """Common fluxes (interior AUSM, wall pressure at the four domain sides); entries the reference
    never writes stay zero."""
    f1 = numpy.zeros_like(qi1)
    f3 = numpy.zeros_like(qi3)
    c = _ausm(qi1[:, :, :-1, n:], qi1[:, :, 1:, :n], 0)
    f1[:, :, :-1, n:] = c
    f1[:, :, 1:, :n] = c
    c = _ausm(qi3[:, :-1, :, n:], qi3[:, 1:, :, :n], 1)
    f3[:, :-1, :, n:] = c
    f3[:, 1:, :, :n] = c

    def wall(rt):
        return p0 * numpy.power(rt * Rd * (1.0 / p0), heat_capacity_ratio)

    for sl in ((slice(None), 0, slice(None, n)), (slice(None), -1, slice(n, None))):
        f1[(slice(None),) + sl] = 0.0
        f1[(1,) + sl] = wall(qi1[(3,) + sl])
    for sl in ((0, slice(None), slice(None, n)), (-1, slice(None), slice(n, None))):
        f3[(slice(None),) + sl] = 0.0
        f3[(2,) + sl] = wall(qi3[(3,) + sl])
    return f1, f3


class Cart2DOracle:
    def __init__(self, n, nx, nz, dx1, dx3, ops):
        self.n, self.nx, self.nz, self.dx1, self.dx3 = n, nx, nz, dx1, dx3
        self.em = numpy.asarray(ops["extrap_neg"], dtype=float)
        self.ep = numpy.asarray(ops["extrap_pos"], dtype=float)
        self.D = numpy.asarray(ops["diff_solpt"], dtype=float)
        self.C = numpy.asarray(ops["correction"], dtype=float)

    def _el(self, a):
        return a.reshape(a.shape[:-1] + (self.n, self.n))

    def extrap(self, a, d):
        e = self._el(a)
        sub = "...ki,i->...k" if d == 0 else "...ki,k->...i"
        return numpy.concatenate((numpy.einsum(sub, e, self.em), numpy.einsum(sub, e, self.ep)), axis=-1)

    def deriv(self, a, d):
        e = self._el(a)
        r = numpy.einsum("ab,...kb->...ka", self.D, e) if d == 0 else numpy.einsum("ab,...bi->...ai", self.D, e)
        return r.reshape(a.shape)

    def correct(self, f, d):
        n = self.n
        lo, hi = f[..., :n], f[..., n:]
        cm, cp = self.C[:, 0], self.C[:, 1]
        r = (lo[..., :, None] * cm + hi[..., :, None] * cp) if d == 0 else \
            (lo[..., None, :] * cm[:, None] + hi[..., None, :] * cp[:, None])
        return r.reshape(f.shape[:-1] + (n * n,))

    def rhs(self, q, want=None):
        qi1, qi3 = self.extrap(q, 0), self.extrap(q, 1)
        f1, f3 = pointwise(q)
        fi1, fi3 = riemann(qi1, qi3, self.n)
        d1 = (self.deriv(f1, 0) + self.correct(fi1, 0)) * (-2.0 / self.dx1)
        d3 = (self.deriv(f3, 1) + self.correct(fi3, 1)) * (-2.0 / self.dx3)
        r = d1 + d3
        r[2] -= q[0] * gravity
        if want is not None:
            want.update(dict(qi1=qi1, qi3=qi3, f1=f1, f3=f3, fi1=fi1, fi3=fi3, d1=d1, d3=d3))
        return r
