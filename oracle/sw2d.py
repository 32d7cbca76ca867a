"""CPU oracle: shallow-water RHS on one cubed-sphere panel, sum-factorised NumPy.

TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this; the product path (wxfactory_amd/) never does.
Parity status: PINNED against golden vectors produced by running the reference
(oracle/refharness/gen_golden.py, tests/test_oracle_sw2d.py).

Restates reference wx_factory/rhs/rhs_sw.py:58-240 (`RhsShallowWater.__compute_rhs__`):
  extrapolate()  :76-90     (h + hsurf is what gets extrapolated and exchanged)
  pack_edges()   :103-117 + process_topology.py:269-386 (rotation of (hu1, hu2), flip)
  rhs()          pointwise fluxes :120-131, derivatives :134-135, halo fill :138-155,
                 AUSM common fluxes in i then j :157-207, corrections :210-211,
                 Coriolis / metric / topography forcing :213-235, assembly :238

Layouts (geometry/cubed_sphere_2d.py:134-170): Q (3, H, H, n^2), point p = jl*n + il;
interface arrays halo-padded (3, H, H+2, 2n) / (3, H+2, H, 2n), [:n] minus side, [n:] plus side;
halo faces (3, H, n) per edge S, N, W, E in receiver-local ordering.
"""
import numpy

from . import cubed_sphere as cs

gravity = 9.80616  # common/definitions.py:5
H_, HU1, HU2 = 0, 1, 2


class SW2DOracle:
    def __init__(self, n, H, ops, metric, topo=None, boundary_sn=None, boundary_we=None, panel=0):
        self.n, self.H = n, H
        self.em = numpy.asarray(ops["extrap_neg"], dtype=float)
        self.ep = numpy.asarray(ops["extrap_pos"], dtype=float)
        self.D = numpy.asarray(ops["diff_solpt"], dtype=float)
        self.C = numpy.asarray(ops["correction"], dtype=float)
        self.m = metric
        self.topo = topo if topo else None
        self.boundary_sn, self.boundary_we = boundary_sn, boundary_we
        self.panel = panel

    def _el(self, a):
        return a.reshape(a.shape[:-1] + (self.n, self.n))

    def deriv(self, a, d):
        e = self._el(a)
        r = numpy.einsum("ab,...jb->...ja", self.D, e) if d == 0 else numpy.einsum("ab,...bi->...ai", self.D, e)
        return r.reshape(a.shape)

    def extrap(self, a, d):
        e = self._el(a)
        sub = "...ji,i->...j" if d == 0 else "...ji,j->...i"
        return numpy.concatenate((numpy.einsum(sub, e, self.em), numpy.einsum(sub, e, self.ep)), axis=-1)

    def correct(self, f, d):
        n = self.n
        lo, hi = f[..., :n], f[..., n:]
        cm, cp = self.C[:, 0], self.C[:, 1]
        if d == 0:
            r = lo[..., :, None] * cm + hi[..., :, None] * cp
        else:
            r = lo[..., None, :] * cm[:, None] + hi[..., None, :] * cp[:, None]
        return r.reshape(f.shape[:-1] + (n * n,))

    # ------------------------------------------------------------------ phase 1-2
    def extrapolate(self, q):
        qu = q.copy()
        if self.topo is not None:
            qu[H_] = qu[H_] + self.topo["hsurf"]
        return [self.extrap(qu, 0), self.extrap(qu, 1)]

    def pack_edges(self, itf):
        n, H = self.n, self.H
        qi, qj = itf
        raw = [qj[:, 0, :, :n], qj[:, -1, :, n:], qi[:, :, 0, :n], qi[:, :, -1, n:]]  # S N W E, each (3, H, n)
        out = []
        for e in range(4):
            a = raw[e].reshape(3, H * n).copy()
            bd = self.boundary_sn if e < 2 else self.boundary_we
            a[HU1], a[HU2] = cs.rotate(self.panel, e, a[HU1], a[HU2], bd)
            if cs.FLIP[self.panel][e]:
                a = numpy.flip(a, axis=-1)
            out.append(numpy.ascontiguousarray(a).reshape(3, H, n))
        return out

    # ------------------------------------------------------------------ the rest
    def rhs(self, q, halo, itf=None, want=None):
        m, n, H = self.m, self.n, self.H
        if itf is None:
            itf = self.extrapolate(q)
        dt = q.dtype
        vi = numpy.zeros((3, H, H + 2, 2 * n), dtype=dt)
        vj = numpy.zeros((3, H + 2, H, 2 * n), dtype=dt)
        vi[:, :, 1:-1, :] = itf[0]
        vj[:, 1:-1, :, :] = itf[1]
        s, nn, w, e = halo
        vj[:, 0, :, n:] = s
        vj[:, -1, :, :n] = nn
        vi[:, :, 0, n:] = w
        vi[:, :, -1, :n] = e
        if self.topo is not None:
            vi[H_] -= self.topo["hsurf_itf_i"]
            vj[H_] -= self.topo["hsurf_itf_j"]

        u1 = q[HU1] / q[H_]
        u2 = q[HU2] / q[H_]
        sg = m["sqrtG"]
        hsq = q[H_] ** 2
        f1 = numpy.empty_like(q)
        f2 = numpy.empty_like(q)
        f1[H_] = sg * q[HU1]
        f2[H_] = sg * q[HU2]
        f1[HU1] = sg * (q[HU1] * u1 + 0.5 * gravity * m["H_contra_11"] * hsq)
        f2[HU1] = sg * (q[HU1] * u2 + 0.5 * gravity * m["H_contra_12"] * hsq)
        f1[HU2] = sg * (q[HU2] * u1 + 0.5 * gravity * m["H_contra_21"] * hsq)
        f2[HU2] = sg * (q[HU2] * u2 + 0.5 * gravity * m["H_contra_22"] * hsq)
        df1 = self.deriv(f1, 0)
        df2 = self.deriv(f2, 1)

        def ausm(v, sgi, hdd, hod, un_idx, axis):
            L = [slice(None)] * v.ndim
            R = [slice(None)] * v.ndim
            L[axis], L[-1] = slice(None, -1), slice(n, None)
            R[axis], R[-1] = slice(1, None), slice(None, n)
            L, R = tuple(L), tuple(R)
            sL, sR = L[1:], R[1:]
            with numpy.errstate(all="ignore"):
                a = numpy.sqrt(gravity * v[H_] * hdd)
                tmp = v[H_] * a
                mach = numpy.where(tmp != 0.0, v[un_idx] / numpy.where(tmp != 0.0, tmp, 1.0), 0.0)
            M = 0.25 * ((mach[sL] + 1.0) ** 2 - (mach[sR] - 1.0) ** 2)
            flux = numpy.zeros_like(v)
            flux[L] = sgi[sL] * (numpy.maximum(0.0, M) * a[sL] * v[L] + numpy.minimum(0.0, M) * a[sR] * v[R])
            pdd = sgi * (0.5 * gravity) * hdd * v[H_] ** 2
            pod = sgi * (0.5 * gravity) * hod * v[H_] ** 2
            # pressure part lands on the normal component with hdd and on the other with hod
            other = HU2 if un_idx == HU1 else HU1
            flux[un_idx][sL] += 0.5 * ((1.0 + mach[sL]) * pdd[sL] + (1.0 - mach[sR]) * pdd[sR])
            flux[other][sL] += 0.5 * ((1.0 + mach[sL]) * pod[sL] + (1.0 - mach[sR]) * pod[sR])
            flux[R] = flux[L]
            return flux

        fi = ausm(vi, m["sqrtG_itf_i"], m["H_contra_11_itf_i"], m["H_contra_21_itf_i"], HU1, -2)
        fj = ausm(vj, m["sqrtG_itf_j"], m["H_contra_22_itf_j"], m["H_contra_12_itf_j"], HU2, -3)
        df1 = df1 + self.correct(fi[:, :, 1:-1, :], 0)
        df2 = df2 + self.correct(fj[:, 1:-1, :, :], 1)

        if self.topo is None:
            dz1 = dz2 = 0.0
        else:
            dz1, dz2 = self.topo["dzdx1"], self.topo["dzdx2"]
        forcing = numpy.zeros_like(q)
        forcing[HU1] = (
            2.0 * (m["christoffel_1_01"] * q[HU1] + m["christoffel_1_02"] * q[HU2])
            + m["christoffel_1_11"] * q[HU1] * u1
            + 2.0 * m["christoffel_1_12"] * q[HU1] * u2
            + gravity * q[H_] * (m["H_contra_11"] * dz1 + m["H_contra_12"] * dz2)
        )
        forcing[HU2] = (
            2.0 * (m["christoffel_2_01"] * q[HU1] + m["christoffel_2_02"] * q[HU2])
            + 2.0 * m["christoffel_2_12"] * q[HU1] * u2
            + m["christoffel_2_22"] * q[HU2] * u2
            + gravity * q[H_] * (m["H_contra_21"] * dz1 + m["H_contra_22"] * dz2)
        )
        inv_sg = m["inv_sqrtG"] if "inv_sqrtG" in m else 1.0 / sg
        R = inv_sg * (-df1 - df2) - forcing
        if want is not None:
            want.update(dict(df1=df1, df2=df2, forcing=forcing, inv_sg=inv_sg, fi=fi, fj=fj, vi=vi, vj=vj))
        return R

    @staticmethod
    def cancel_scale(want):
        ax = (1, 2, 3)
        s = numpy.abs(want["forcing"].real).max(axis=ax)
        for k in ("df1", "df2"):
            s = numpy.maximum(s, numpy.abs((want["inv_sg"] * want[k]).real).max(axis=ax))
        return s


def sphere_rhs(oracles, qs):
    itfs = [o.extrapolate(q) for o, q in zip(oracles, qs)]
    recvs = cs.route([o.pack_edges(itf) for o, itf in zip(oracles, itfs)])
    return [o.rhs(q, recvs[p], itf=itfs[p]) for p, (o, q) in enumerate(zip(oracles, qs))]
