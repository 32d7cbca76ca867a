"""Setup-time 3-D geometry and topography-following metric of the cubed sphere.

Restates, as NumPy setup code that runs once per tile (its outputs are the static inputs of the Euler RHS
kernels, `wx_euler3d_metric` in include/wxhip.h):
  * the tile's computational coordinates (x1, x2, eta), planet scaling per test case, terrain-following
    height  z = z_s + (ztop - z_s) eta                      geometry/cubed_sphere_3d.py:19-503
  * gnomonic -> Cartesian -> (lon, lat)                      cubed_sphere_3d.py:597-700, sphere.py:25-45
  * the Schaer mountain of DCMIP 2-1 / 2-2                   init/dcmip.py:557-605
  * Metric3DTopo.build_metric                                geometry/metric3d.py:21-1157:
      dz/dx1, dz/dx2, dz/deta by DFR differentiation; their interface values as the two-sided average,
      horizontally through a (buggy-as-shipped, reproduced) 2-D contravariant conversion and the panel-edge
      vector exchange; covariant / contravariant metric and sqrt(g) at nodes and on the three interface
      families; the rotation Christoffel symbols in closed form; the 18 spatial ones from the pointwise
      27 x 27 system  (sqrtG h^ab)_,c = sqrtG (G^d_cd h^ab - G^a_dc h^db - G^b_cd h^ad)
  * the element-blocked / halo-padded layouts the RHS reads  cubed_sphere_3d.py:759-897

Design differences from the reference (results equal to rounding, pinned in tests/test_geometry3d.py against
the metric arrays of the reference-generated fixtures, with and without topography):
  * no communication: what a neighbouring tile would send across an edge is recomputed locally from a
    one-element-deep strip of that neighbour (heights are analytic in (lon, lat, eta)), so the setup of a
    tile needs nothing but its own parameters - any decomposition, any rank count;
  * work arrays are (nk, nj, ni) grids, converted to the kernels' layouts once at the end;
  * the pointwise 27 x 27 system for the spatial Christoffel symbols is inverted in closed form (default;
    ~100x faster than LAPACK on 27 x 27 blocks, which the reference materialises all at once: 86 GB per panel at
    E7); `christoffel="solve"` runs the reference's literal procedure in bounded slabs as a cross-check.
"""
import math
import os
from typing import Callable, Dict, Optional

import numpy

from .geometry import EARTH_RADIUS, ROTATION_SPEED, gauss_legendre, panel_centre
from .panels import FLIP, NEIGHBOR, landing_edge
from .synthetic import dfr_ops

SOUTH, NORTH, WEST, EAST = 0, 1, 2, 3

# convert_contra of process_topology.py:137-175 as coefficients, c = 2X/(1+X^2):
#   b1 = m0 a1 + m1 a2 + c (m2 a1 + m3 a2),  b2 = m4 a1 + m5 a2 + c (m6 a1 + m7 a2)
_WE = ((1, 0, 0, 0, 0, 1, 1, 0), (1, 0, 0, 0, 0, 1, -1, 0))
ROT_CONTRA = (
    ((1, 0, 0, 1, 0, 1, 0, 0), (1, 0, 0, -1, 0, 1, 0, 0)) + _WE,
    ((0, 1, 0, 0, -1, 0, 0, -1), (0, -1, 0, 0, 1, 0, 0, -1)) + _WE,
    ((-1, 0, 0, -1, 0, -1, 0, 0), (-1, 0, 0, 1, 0, -1, 0, 0)) + _WE,
    ((0, -1, 0, 0, 1, 0, 0, 1), (0, 1, 0, 0, -1, 0, 0, 1)) + _WE,
    ((1, 0, 0, 1, 0, 1, 0, 0), (-1, 0, 0, 1, 0, -1, 0, 0), (0, -1, -1, 0, 1, 0, 0, 0), (0, 1, -1, 0, -1, 0, 0, 0)),
    ((-1, 0, 0, -1, 0, -1, 0, 0), (1, 0, 0, -1, 0, 1, 0, 0), (0, 1, 1, 0, -1, 0, 0, 0), (0, -1, 1, 0, 1, 0, 0, 0)),
)


def planet_for_case(case_number: int):
    """(earth_radius, rotation_speed) after the small-planet scalings of cubed_sphere_3d.py:405-425."""
    scaling, rotating = 1.0, 1.0
    if case_number == 31:
        scaling, rotating = 125.0, 0.0
    elif case_number == 20:
        rotating = 0.0
    elif case_number in (21, 22):
        scaling, rotating = 500.0, 0.0
    return EARTH_RADIUS / scaling, ROTATION_SPEED * rotating / scaling


def schar_mountain(earth_radius: float, lambdam=math.pi / 4.0, phim=0.0, h0=250.0, Dm=5000.0, Dxi=4000.0) -> Callable:
    """Surface height of DCMIP 2-1/2-2 as a function of (lon, lat) (init/dcmip.py:557-593)."""

    def topo(lon, lat):
        r = earth_radius * numpy.arccos(math.sin(phim) * numpy.sin(lat) + math.cos(phim) * numpy.cos(lat) * numpy.cos(lon - lambdam))
        return h0 * numpy.exp(-(r**2) / Dm**2) * numpy.cos(numpy.pi * r / Dxi) ** 2

    return topo


def topography_for_case(case_number: int, earth_radius: float) -> Optional[Callable]:
    return schar_mountain(earth_radius) if case_number in (21, 22) else None


class CubedSphere3DTile:
    """Coordinates of a rectangular block of elements of one panel.

    `H` elements per tile side, `k` x `k` tiles per panel (tile (row, col), as ProcessTopology numbers them);
    `elems=(j0, nj_el, i0, ni_el)` instead selects an arbitrary element block of the panel (used for the
    one-element-deep neighbour strips)."""

    def __init__(self, n: int, H: int, V: int, panel: int, ztop: float, case_number: int = 31,
                 depth_approx: str = "shallow", row: int = 0, col: int = 0, k: int = 1, lambda0: float = 0.0,
                 phi0: float = 0.0, alpha0: float = 0.0, topo: Optional[Callable] = None, elems=None):
        if depth_approx not in ("deep", "shallow"):
            raise ValueError(f"Invalid Euler atmosphere depth approximation ({depth_approx})")
        self.n, self.H, self.V, self.panel, self.k, self.row, self.col = n, H, V, panel, k, row, col
        self.ztop, self.case_number, self.deep = ztop, case_number, depth_approx == "deep"
        self.rotation = (lambda0, phi0, alpha0)
        self.per_side = H * k
        self.earth_radius, self.rotation_speed = planet_for_case(case_number)
        self.topo = topo
        self.lon_p, self.lat_p, self.angle_p = panel_centre(panel, lambda0, phi0, alpha0)
        j0, nje, i0, nie = elems if elems is not None else (row * H, H, col * H, H)
        self.j0, self.nje, self.i0, self.nie = j0, nje, i0, nie
        pts, _ = gauss_legendre(n)
        self.solution_points = pts
        d = (math.pi / 2.0) / self.per_side
        self.delta_x1 = self.delta_x2 = d
        self.delta_eta = 1.0 / V
        ref = 0.5 * (1.0 + pts)
        self.x1_itf = -math.pi / 4.0 + d * (i0 + numpy.arange(nie + 1))
        self.x2_itf = -math.pi / 4.0 + d * (j0 + numpy.arange(nje + 1))
        if elems is None:  # bit-compatible with the reference's linspace over the tile
            self.x1_itf = numpy.linspace(-math.pi / 4 + col * H * d, -math.pi / 4 + col * H * d + H * d, H + 1)
            self.x2_itf = numpy.linspace(-math.pi / 4 + row * H * d, -math.pi / 4 + row * H * d + H * d, H + 1)
        self.eta_itf = numpy.linspace(0.0, 1.0, V + 1)
        self.x1 = numpy.repeat(self.x1_itf[:-1], n) + numpy.tile(d * ref, nie)
        self.x2 = numpy.repeat(self.x2_itf[:-1], n) + numpy.tile(d * ref, nje)
        self.eta = numpy.repeat(self.eta_itf[:-1], n) + numpy.tile(self.delta_eta * ref, V)
        self.ni, self.nj, self.nk = nie * n, nje * n, V * n
        self.boundary_sn = numpy.tan(self.x1)  # X along the south / north edges
        self.boundary_we = numpy.tan(self.x2)  # Y along the west / east edges

    # -- physical coordinates
    def lonlat(self, X, Y):
        """(lon in [0, 2pi), lat) of gnomonic points of this panel (cubed_sphere_3d.py:628-700)."""
        lp, tp, ap = self.lon_p, self.lat_p, self.angle_p
        cl, sl, ct, st, ca, sa = math.cos(lp), math.sin(lp), math.cos(tp), math.sin(tp), math.cos(ap), math.sin(ap)
        # the radius cancels in both angles except for rounding; keep the reference's scaling
        s = self.earth_radius / numpy.sqrt(1.0 + X**2 + Y**2)
        cx = s * (cl * ct + X * (cl * st * sa - sl * ca) - Y * (cl * st * ca + sl * sa))
        cy = s * (sl * ct + X * (sl * st * sa + cl * ca) - Y * (sl * st * ca - cl * sa))
        cz = s * (st - X * ct * sa + Y * ct * ca)
        lon = numpy.arctan2(cy, cx)
        lon = numpy.where(lon < 0.0, lon + 2.0 * math.pi, lon)
        return lon, numpy.arctan2(cz, numpy.hypot(cx, cy))

    def surface(self, x1, x2):
        """Surface height on the outer product x2 (rows) x x1 (columns)."""
        X, Y = numpy.meshgrid(numpy.tan(x1), numpy.tan(x2))
        if self.topo is None:
            return numpy.zeros(X.shape)
        lon, lat = self.lonlat(X, Y)
        return self.topo(lon, lat)

    def heights(self):
        """Terrain-following heights at nodes and on the three interface families (apply_topography)."""
        zt = self.ztop
        zb = self.surface(self.x1, self.x2)                # (nj, ni)
        zb_i = self.surface(self.x1_itf, self.x2)          # (nj, nie+1)
        zb_j = self.surface(self.x1, self.x2_itf)          # (nje+1, ni)
        e = self.eta[:, None, None]
        return {
            "int": zb[None] + (zt - zb[None]) * e,
            "itf_i": zb_i[None] + (zt - zb_i[None]) * e,
            "itf_j": zb_j[None] + (zt - zb_j[None]) * e,
            "itf_k": zb[None] + (zt - zb[None]) * self.eta_itf[:, None, None],
        }

    # -- layout conversions (cubed_sphere_3d.py:759-897)
    def to_blocked(self, a):
        n, V, Hj, Hi = self.n, self.V, self.nje, self.nie
        lead = a.shape[:-3]
        t = a.reshape(lead + (V, n, Hj, n, Hi, n))
        t = numpy.moveaxis(t, (-5, -3), (-3, -2))  # (V, Hj, Hi, n, n, n)
        return numpy.ascontiguousarray(t).reshape(lead + (V, Hj, Hi, n**3))

    def to_itf_i(self, a):
        """(.., nk, nj, Hi+1) -> (.., V, Hj, Hi+2, 2 n^2); face point (kl, jl)."""
        n, V, Hj, Hi = self.n, self.V, self.nje, self.nie
        lead = a.shape[:-3]
        t = numpy.moveaxis(a.reshape(lead + (V, n, Hj, n, Hi + 1)), (-4, -2), (-2, -1)).reshape(lead + (V, Hj, Hi + 1, n * n))
        out = numpy.zeros(lead + (V, Hj, Hi + 2, 2 * n * n))
        out[..., 1:, : n * n] = t
        out[..., :-1, n * n:] = t
        return out

    def to_itf_j(self, a):
        """(.., nk, Hj+1, ni) -> (.., V, Hj+2, Hi, 2 n^2); face point (kl, il)."""
        n, V, Hj, Hi = self.n, self.V, self.nje, self.nie
        lead = a.shape[:-3]
        t = numpy.moveaxis(a.reshape(lead + (V, n, Hj + 1, Hi, n)), -4, -2).reshape(lead + (V, Hj + 1, Hi, n * n))
        out = numpy.zeros(lead + (V, Hj + 2, Hi, 2 * n * n))
        out[..., 1:, :, : n * n] = t
        out[..., :-1, :, n * n:] = t
        return out

    def to_itf_k(self, a):
        """(.., V+1, nj, ni) -> (.., V+2, Hj, Hi, 2 n^2); face point (jl, il)."""
        n, V, Hj, Hi = self.n, self.V, self.nje, self.nie
        lead = a.shape[:-3]
        t = numpy.swapaxes(a.reshape(lead + (V + 1, Hj, n, Hi, n)), -3, -2).reshape(lead + (V + 1, Hj, Hi, n * n))
        out = numpy.zeros(lead + (V + 2, Hj, Hi, 2 * n * n))
        out[..., 1:, :, :, : n * n] = t
        out[..., :-1, :, :, n * n:] = t
        return out


class _Dfr:
    """The 1-D DFR pieces applied along one axis of (nk, nj, ni) grids (operators.py:263-533)."""

    def __init__(self, n: int):
        o = dfr_ops(n)
        self.n = n
        self.D, self.C = o["diff_solpt"], o["correction"]
        self.em, self.ep = o["extrap_neg"], o["extrap_pos"]

    def _split(self, f, axis):
        f = numpy.moveaxis(f, axis, -1)
        return f.reshape(f.shape[:-1] + (f.shape[-1] // self.n, self.n))

    def comma(self, f, lo, hi, axis):
        """d f / d(reference coordinate) along `axis`; lo / hi = element-boundary values with the element
        index in place of the point index on that axis."""
        fe = self._split(f, axis)                                   # (.., nel, n)
        lo, hi = numpy.moveaxis(lo, axis, -1), numpy.moveaxis(hi, axis, -1)
        out = fe @ self.D.T + lo[..., None] * self.C[:, 0] + hi[..., None] * self.C[:, 1]
        return numpy.moveaxis(out.reshape(out.shape[:-2] + (-1,)), -1, axis)

    def extrap(self, f, axis):
        """(minus, plus) element-boundary values, element index on `axis`."""
        fe = self._split(f, axis)
        return numpy.moveaxis(fe @ self.em, -1, axis), numpy.moveaxis(fe @ self.ep, -1, axis)


def _metric2d_itf(X, Y):
    """The temporary 2-D metric of metric3d.py:213-233 used to ship dz/dx across panels as a contravariant
    vector.  The [0][1] entry divides by a SUM where the true metric has a product - kept as shipped, since the
    interface metric of every reference run carries it."""
    d2 = 1.0 + X**2 + Y**2
    con = numpy.empty((2, 2) + X.shape)
    con[0, 0] = d2 / (1 + X**2)
    con[0, 1] = con[1, 0] = d2 * X * Y / ((1 + X**2) + (1 + Y**2))
    con[1, 1] = d2 / (1 + Y**2)
    cov = numpy.empty((2, 2) + X.shape)
    cov[0, 0] = (1 + X**2) ** 2 * (1 + Y**2) / d2**2
    cov[0, 1] = cov[1, 0] = -X * Y * (1 + X**2) * (1 + Y**2) / d2**2
    cov[1, 1] = (1 + X**2) * (1 + Y**2) ** 2 / d2**2
    return con, cov


class _Slopes:
    """dz/dx1, dz/dx2, dz/deta at the nodes of a tile and the element-side values of their contravariant form
    on the i and j interfaces (metric3d.py:106-126, 262-346)."""

    def __init__(self, t: CubedSphere3DTile, dfr: _Dfr):
        self.t = t
        h = t.heights()
        self.h = h
        hi, hj, hk = h["itf_i"], h["itf_j"], h["itf_k"]
        self.d1 = dfr.comma(h["int"], hi[:, :, :-1], hi[:, :, 1:], 2) * 2 / t.delta_x1
        self.d2 = dfr.comma(h["int"], hj[:, :-1, :], hj[:, 1:, :], 1) * 2 / t.delta_x2
        self.d3 = dfr.comma(h["int"], hk[:-1], hk[1:], 0) * 2 / t.delta_eta
        # gnomonic coordinates of the interface points
        self.Xi, self.Yi = numpy.meshgrid(numpy.tan(t.x1_itf), numpy.tan(t.x2))   # (nj, nie+1)
        self.Xj, self.Yj = numpy.meshgrid(numpy.tan(t.x1), numpy.tan(t.x2_itf))   # (nje+1, ni)
        self.con_i, self.cov_i = _metric2d_itf(self.Xi, self.Yi)
        self.con_j, self.cov_j = _metric2d_itf(self.Xj, self.Yj)
        # element-side values: [component][side 0 = minus, 1 = plus] -> (nk, nj, nie) / (nk, nje, ni)
        e1, e2, e3 = (dfr.extrap(self.d1, 2), dfr.extrap(self.d2, 2), dfr.extrap(self.d3, 2))
        ci = self.con_i
        self.side_i = [[ci[0, 0][None, :, s:t.nie + s] * e1[s] + ci[0, 1][None, :, s:t.nie + s] * e2[s] for s in (0, 1)],
                       [ci[1, 0][None, :, s:t.nie + s] * e1[s] + ci[1, 1][None, :, s:t.nie + s] * e2[s] for s in (0, 1)],
                       [e3[0], e3[1]]]
        e1, e2, e3 = (dfr.extrap(self.d1, 1), dfr.extrap(self.d2, 1), dfr.extrap(self.d3, 1))
        cj = self.con_j
        self.side_j = [[cj[0, 0][None, s:t.nje + s, :] * e1[s] + cj[0, 1][None, s:t.nje + s, :] * e2[s] for s in (0, 1)],
                       [cj[1, 0][None, s:t.nje + s, :] * e1[s] + cj[1, 1][None, s:t.nje + s, :] * e2[s] for s in (0, 1)],
                       [e3[0], e3[1]]]
        self.ext_k = [dfr.extrap(d, 0) for d in (self.d1, self.d2, self.d3)]

    def edge(self, e: int):
        """The three components on the tile's outward side of edge e, each (nk, edge length)."""
        if e == SOUTH:
            return [c[0][:, 0, :] for c in self.side_j]
        if e == NORTH:
            return [c[1][:, -1, :] for c in self.side_j]
        if e == WEST:
            return [c[0][:, :, 0] for c in self.side_i]
        return [c[1][:, :, -1] for c in self.side_i]


def _rotate_contra(panel, e, a1, a2, X):
    m = ROT_CONTRA[panel][e]
    c = 2.0 * X / (1.0 + X**2)
    return (m[0] * a1 + m[1] * a2 + c * (m[2] * a1 + m[3] * a2), m[4] * a1 + m[5] * a2 + c * (m[6] * a1 + m[7] * a2))


def _neighbour_message(t: CubedSphere3DTile, e: int, dfr: _Dfr):
    """What the tile across edge e of `t` sends to it (start_exchange_vectors, process_topology.py:322-386):
    that tile's outward element-side values on the shared edge, rotated into t's basis and flipped where
    the two panels run in opposite directions - computed from a one-element-deep strip of the neighbour."""
    P, n = t.per_side, t.n
    common = dict(ztop=t.ztop, case_number=t.case_number, depth_approx="deep" if t.deep else "shallow", k=t.k,
                  lambda0=t.rotation[0], phi0=t.rotation[1], alpha0=t.rotation[2], topo=t.topo)
    lo = (t.j0, t.j0 + t.nje - 1, t.i0, t.i0 + t.nie - 1)
    on_panel_edge = (lo[0] == 0, lo[1] == P - 1, lo[2] == 0, lo[3] == P - 1)[e]
    if not on_panel_edge:  # same panel: the adjacent row / column of elements, no rotation, no flip
        if e in (SOUTH, NORTH):
            j = t.j0 - 1 if e == SOUTH else t.j0 + t.nje
            strip = CubedSphere3DTile(n, t.H, t.V, t.panel, elems=(j, 1, t.i0, t.nie), **common)
        else:
            i = t.i0 - 1 if e == WEST else t.i0 + t.nie
            strip = CubedSphere3DTile(n, t.H, t.V, t.panel, elems=(t.j0, t.nje, i, 1), **common)
        return _Slopes(strip, dfr).edge(e ^ 1)
    q, eq = NEIGHBOR[t.panel][e], landing_edge(t.panel, e)
    a0, cnt = (t.i0, t.nie) if e in (SOUTH, NORTH) else (t.j0, t.nje)
    if FLIP[q][eq]:
        a0 = P - a0 - cnt
    far = 0 if eq in (SOUTH, WEST) else P - 1
    elems = (far, 1, a0, cnt) if eq in (SOUTH, NORTH) else (a0, cnt, far, 1)
    strip = CubedSphere3DTile(n, t.H, t.V, q, elems=elems, **common)
    c1, c2, c3 = _Slopes(strip, dfr).edge(eq)
    X = strip.boundary_sn if eq in (SOUTH, NORTH) else strip.boundary_we
    b1, b2 = _rotate_contra(q, eq, c1, c2, X[None, :])
    msg = [b1, b2, c3]
    if FLIP[q][eq]:
        msg = [m[:, ::-1] for m in msg]
    return msg


_PAIRS = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))  # unique components of a symmetric 3 x 3 tensor


def _hcontra_sqrtg(X, Y, R, d1, d2, d3, dx, dy, de, A, deep):
    """Contravariant metric (its six unique components, keyed (a, b), a <= b) and sqrt(g) in reference-element
    units (metric3d.py:519-646; the covariant metric the reference also builds is not read by the RHS).
    `A` = planet radius (shallow atmosphere) - the deep form uses the local radius R instead."""
    rad = R if deep else A
    delsq = 1 + X**2 + Y**2
    h = {}
    h[0, 0] = (4 / dx**2) * (delsq / (rad**2 * (1 + X**2))) + 0.0 * d1
    h[0, 1] = (4 / dx / dy) * (X * Y * delsq / (rad**2 * (1 + X**2) * (1 + Y**2))) + 0.0 * d1
    h[0, 2] = (4 / dx / de) * (
        -(d1 * delsq / (rad**2 * (1 + X**2)) + d2 * delsq * X * Y / (rad**2 * (1 + X**2) * (1 + Y**2))) / d3)
    h[1, 1] = (4 / dy**2) * (delsq / (rad**2 * (1 + Y**2))) + 0.0 * d1
    h[1, 2] = (4 / dy / de) * (
        -(d1 * X * Y * delsq / (rad**2 * (1 + X**2) * (1 + Y**2)) + d2 * delsq / (rad**2 * (1 + Y**2))) / d3)
    h[2, 2] = (4 / de**2) * (1 + d1**2 * delsq / (rad**2 * (1 + X**2))
                             + 2 * d1 * d2 * X * Y * delsq / (rad**2 * (1 + X**2) * (1 + Y**2))
                             + d2**2 * delsq / (rad**2 * (1 + Y**2))) / d3**2
    rootg = (dx / 2) * (dy / 2) * (de / 2) * rad**2 * (1 + X**2) * (1 + Y**2) * abs(d3) / delsq**1.5
    return h, rootg


def _sym(h, a, b):
    return h[(a, b) if a <= b else (b, a)]


def _space_christoffel_solve(T, h, sg):
    """The 18 spatial Christoffel symbols exactly as the reference obtains them (metric3d.py:905-957): the
    pointwise 27 x 27 system  (sqrtG h^ab)_,c = sqrtG (G^d_cd h^ab - G^a_dc h^db - G^b_cd h^ad)  solved with
    LAPACK.  ~40 us per point: kept as the cross-check of the closed form below.  T[(a,b)][c] = the left side."""
    shp = sg.shape
    lhs = numpy.zeros(shp + (3, 3, 3, 3, 3, 3))
    rhs = numpy.empty(shp + (3, 3, 3))
    for a in range(3):
        for b in range(3):
            for c in range(3):
                rhs[..., a, b, c] = _sym(T, a, b)[c]
                for d in range(3):
                    lhs[..., a, b, c, d, c, d] += sg * _sym(h, a, b)
                    lhs[..., a, b, c, a, d, c] -= sg * _sym(h, d, b)
                    lhs[..., a, b, c, b, c, d] -= sg * _sym(h, a, d)
    sol = numpy.linalg.solve(lhs.reshape(shp + (27, 27)), rhs.reshape(shp + (27, 1)))[..., 0].reshape(shp + (3, 3, 3))
    return {(d, b, c): sol[..., d, b, c] for d in range(3) for (b, c) in _PAIRS}


def _space_christoffel_closed(T, h, sg):
    """The same symbols by inverting that linear relation by hand.  With T^ab_c = (sqrtG h^ab)_,c / sqrtG and
    g = (h^..)^-1:   G^d_cd = g_ab T^ab_c =: t_c;   S^ab_c = t_c h^ab - T^ab_c = G^a_dc h^db + G^b_dc h^da;
    lowering both indices, S_abc = G_a,bc + G_b,ac, hence G_a,bc = (S_abc + S_acb - S_bca) / 2 and
    G^d_bc = h^da G_a,bc.  The 27 x 27 system is non-singular, so this symmetric solution is its solution."""
    H = lambda a, b: _sym(h, a, b)  # noqa: E731
    det = (H(0, 0) * (H(1, 1) * H(2, 2) - H(1, 2) * H(1, 2)) - H(0, 1) * (H(0, 1) * H(2, 2) - H(1, 2) * H(0, 2))
           + H(0, 2) * (H(0, 1) * H(1, 2) - H(1, 1) * H(0, 2)))
    inv = 1.0 / det
    g = {}
    for a, b in _PAIRS:  # adjugate / det
        a1, a2, b1, b2 = (a + 1) % 3, (a + 2) % 3, (b + 1) % 3, (b + 2) % 3
        g[a, b] = (H(a1, b1) * H(a2, b2) - H(a1, b2) * H(a2, b1)) * inv
    G = lambda a, b: _sym(g, a, b)  # noqa: E731
    Sl = {}  # S_abc, symmetric in (a, b): keyed (a, b, c) with a <= b
    for c in range(3):
        tc = sum((1.0 if a == b else 2.0) * g[a, b] * (T[a, b][c] / sg) for a, b in _PAIRS)
        S = {ab: tc * h[ab] - T[ab][c] / sg for ab in _PAIRS}
        M = [[sum(G(a, p) * _sym(S, p, q) for p in range(3)) for q in range(3)] for a in range(3)]  # (g S)_a^q
        for a, b in _PAIRS:
            Sl[a, b, c] = sum(M[a][q] * G(q, b) for q in range(3))
    SL = lambda a, b, c: Sl[(a, b, c) if a <= b else (b, a, c)]  # noqa: E731
    low = {(a, b, c): 0.5 * (SL(a, b, c) + SL(a, c, b) - SL(b, c, a)) for a in range(3) for (b, c) in _PAIRS}
    return {(d, b, c): sum(H(d, a) * low[a, b, c] for a in range(3)) for d in range(3) for (b, c) in _PAIRS}


def metric3d(t: CubedSphere3DTile, christoffel: str = "closed", threads: Optional[int] = None, device=None,
             rows_per_block: Optional[int] = None):
    """Every static array `wx_euler3d_metric` needs (minus the case-21/22 sponge fields), named as in
    include/wxhip.h, in the kernels' layouts.  Metric3DTopo.build_metric for one tile.

    The slopes (three fields) are computed for the whole tile with NumPy; everything after them is local to an
    element, so the metric, its derivatives and the Christoffel symbols are produced one block of
    (vertical layer, element rows) at a time and written straight into the final layouts:
      * device=None: NumPy, cache-sized blocks (one element row) spread over `threads`; returns ndarrays;
      * device=<torch device>: the same block code on torch tensors (whole layers per block); returns tensors
        on that device - the 4.6 GB of an E7 panel are produced where the kernels read them."""
    if christoffel not in ("closed", "solve"):
        raise ValueError("christoffel must be 'closed' or 'solve'")
    use_torch = device is not None
    if use_torch and christoffel != "closed":
        raise ValueError("the LAPACK cross-check runs on the NumPy path only")
    dfr = _Dfr(t.n)
    sl = _Slopes(t, dfr)
    n, nj, ni, Hi, Hj, V = t.n, t.nj, t.ni, t.nie, t.nje, t.V
    n2, n3 = n * n, n * n * n
    dx, dy, de = t.delta_x1, t.delta_x2, t.delta_eta
    A = t.earth_radius

    # -- interface slopes: average of the two element sides; halo sides come from the neighbours
    halo = [_neighbour_message(t, e, dfr) for e in range(4)]

    def faces(side, axis, lo_msg, hi_msg):
        """(3, .., nfaces, ..): 0.5 (plus side of the element below + minus side of the element above)."""
        out = []
        for c in range(3):
            minus, plus = side[c]
            lo = numpy.expand_dims(lo_msg[c], axis)
            hi = numpy.expand_dims(hi_msg[c], axis)
            below = numpy.concatenate((lo, plus), axis=axis)    # plus side of element f-1 (halo for f = 0)
            above = numpy.concatenate((minus, hi), axis=axis)   # minus side of element f (halo for the last face)
            out.append(0.5 * below + 0.5 * above)
        return out

    a_i = faces(sl.side_i, 2, halo[WEST], halo[EAST])     # (nk, nj, Hi+1) x 3
    a_j = faces(sl.side_j, 1, halo[SOUTH], halo[NORTH])   # (nk, Hj+1, ni) x 3
    di = [sl.cov_i[0, 0][None] * a_i[0] + sl.cov_i[0, 1][None] * a_i[1],
          sl.cov_i[1, 0][None] * a_i[0] + sl.cov_i[1, 1][None] * a_i[1], a_i[2]]
    dj = [sl.cov_j[0, 0][None] * a_j[0] + sl.cov_j[0, 1][None] * a_j[1],
          sl.cov_j[1, 0][None] * a_j[0] + sl.cov_j[1, 1][None] * a_j[1], a_j[2]]
    dk = []
    for lo, hi in sl.ext_k:  # vertical: one-sided at the bottom and the top, average inside
        f = numpy.empty((V + 1, nj, ni))
        f[0], f[-1] = lo[0], hi[-1]
        f[1:-1] = 0.5 * (lo[1:] + hi[:-1])
        dk.append(f)
    del a_i, a_j, halo

    # -- array namespace of the block stage
    if use_torch:
        import torch

        dev = torch.device(device)
        on = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).to(dev)  # noqa: E731
        empty = lambda shape: torch.empty(shape, dtype=torch.float64, device=dev)  # noqa: E731
        zeros = lambda shape: torch.zeros(shape, dtype=torch.float64, device=dev)  # noqa: E731
        perm = lambda a, *ax: a.permute(*ax)  # noqa: E731
    else:
        on = lambda a: a  # noqa: E731
        empty, zeros = numpy.empty, numpy.zeros
        perm = lambda a, *ax: a.transpose(*ax)  # noqa: E731
    h = {k_: on(v) for k_, v in sl.h.items()}
    d_int = [on(sl.d1), on(sl.d2), on(sl.d3)]
    di, dj, dk = [on(a) for a in di], [on(a) for a in dj], [on(a) for a in dk]
    tx1, tx2 = on(numpy.tan(t.x1)), on(numpy.tan(t.x2))
    tx1f, tx2f = on(numpy.tan(t.x1_itf)), on(numpy.tan(t.x2_itf))
    D, C0, C1 = on(dfr.D), on(dfr.C[:, 0].copy()), on(dfr.C[:, 1].copy())
    DT = on(dfr.D.T.copy())
    sphi, cphi = math.sin(t.lat_p), math.cos(t.lat_p)
    salp, calp = math.sin(t.angle_p), math.cos(t.angle_p)
    Om = t.rotation_speed

    O = {
        "sqrtG": empty((V, Hj, Hi, n3)), "h_contra": empty((3, 3, V, Hj, Hi, n3)),
        "christoffel": zeros((3, 9, V, Hj, Hi, n3)), "inv_dzdeta": empty((V, Hj, Hi, n3)),
        "sqrtG_itf_i": zeros((V, Hj, Hi + 2, 2 * n2)), "h_contra_itf_i": zeros((3, 3, V, Hj, Hi + 2, 2 * n2)),
        "sqrtG_itf_j": zeros((V, Hj + 2, Hi, 2 * n2)), "h_contra_itf_j": zeros((3, 3, V, Hj + 2, Hi, 2 * n2)),
        "sqrtG_itf_k": zeros((V + 2, Hj, Hi, 2 * n2)), "h_contra_itf_k": zeros((3, 3, V + 2, Hj, Hi, 2 * n2)),
    }

    def put(name, idx, value):          # scalar field, or the symmetric tensor {(a, b): field}
        if isinstance(value, dict):
            for (a, b), v in value.items():
                O["h_contra" + name][(a, b) + idx] = v
                if a != b:
                    O["h_contra" + name][(b, a) + idx] = v
        else:
            O["sqrtG" + name][idx] = value

    def both(fn, hc, sg):
        return {ab: fn(v) for ab, v in hc.items()}, fn(sg)

    def block(spec):
        kv, r0, r1 = spec                      # vertical layer, element rows [r0, r1)
        nr = r1 - r0
        ks, js = slice(kv * n, (kv + 1) * n), slice(r0 * n, r1 * n)
        X = tx1[None, None, None, :]
        Y = tx2[js].reshape(nr, n)[None, :, :, None]
        nodes = lambda a: a[ks, js].reshape(n, nr, n, -1)  # noqa: E731  [k, row, j, i]
        d1, d2, d3 = (nodes(a) for a in d_int)
        R_int = nodes(h["int"]) + A
        hc, sg = _hcontra_sqrtg(X, Y, R_int, d1, d2, d3, dx, dy, de, A, t.deep)
        hci, sgi = _hcontra_sqrtg(tx1f[None, None, None, :], Y, nodes(h["itf_i"]) + A, nodes(di[0]), nodes(di[1]),
                                  nodes(di[2]), dx, dy, de, A, t.deep)                      # [k, row, j, face]
        fj = slice(r0, r1 + 1)
        hcj, sgj = _hcontra_sqrtg(tx1[None, None, :], tx2f[fj][None, :, None], h["itf_j"][ks, fj] + A, dj[0][ks, fj],
                                  dj[1][ks, fj], dj[2][ks, fj], dx, dy, de, A, t.deep)      # [k, face, i]
        fk = slice(kv, kv + 2)
        kf = lambda a: a[fk, js].reshape(2, nr, n, ni)  # noqa: E731
        hck, sgk = _hcontra_sqrtg(X, Y, kf(h["itf_k"]) + A, kf(dk[0]), kf(dk[1]), kf(dk[2]), dx, dy, de, A, t.deep)
        # (sqrtG h^ab)_,c in reference-element units (operators.grad, operators.py:635-700), unique (a, b) only
        T = {}
        for ab in _PAIRS:
            f, fi, fjj, fkk = hc[ab] * sg, hci[ab] * sgi, hcj[ab] * sgj, hck[ab] * sgk
            gi = (f.reshape(n, nr, n, Hi, n) @ DT + fi[..., :-1, None] * C0 + fi[..., 1:, None] * C1).reshape(n, nr, n, ni)
            gj = D @ f + C0[None, None, :, None] * fjj[:, :-1, None, :] + C1[None, None, :, None] * fjj[:, 1:, None, :]
            gk = (D @ f.reshape(n, -1)).reshape(f.shape) + C0[:, None, None, None] * fkk[0][None] + C1[:, None, None, None] * fkk[1][None]
            T[ab] = (gi, gj, gk)
        gam = (_space_christoffel_closed if christoffel == "closed" else _space_christoffel_solve)(T, hc, sg)

        rows = slice(r0, r1)
        blocked = lambda a: perm(a.reshape(n, nr, n, Hi, n), 1, 3, 0, 2, 4).reshape(nr, Hi, n3)  # noqa: E731
        O["sqrtG"][kv, rows] = blocked(sg)
        O["inv_dzdeta"][kv, rows] = blocked(1 / d3 * 2 / de)
        for (a, b), v in hc.items():
            O["h_contra"][a, b, kv, rows] = blocked(v)
            if a != b:
                O["h_contra"][b, a, kv, rows] = O["h_contra"][a, b, kv, rows]
        for (d, b, c), v in gam.items():
            O["christoffel"][d, 3 + _PAIRS.index((b, c)), kv, rows] = blocked(v)
        if Om != 0.0:  # rotation part in closed form (metric3d.py:661-836)
            R = R_int if t.deep else A
            rot1 = sphi - X * cphi * salp + Y * cphi * calp
            rot2 = (1 + X**2) * cphi * calp - Y * sphi + X * Y * cphi * salp
            rot3 = (1 + Y**2) * cphi * salp + X * sphi + X * Y * cphi * calp
            dsq = 1 + X**2 + Y**2
            c101 = Om * X * Y / dsq * rot1 + d1 * Om / (R * (1 + X**2)) * rot2
            c102 = -Om * (-(1 + Y**2) / dsq) * rot1 + d2 * Om / (R * (1 + X**2)) * rot2
            c103 = d3 * Om / (R * (1 + X**2)) * rot2
            c201 = Om * (1 + X**2) / dsq * rot1 + d1 * Om / (R * (1 + Y**2)) * rot3
            c202 = -Om * X * Y / dsq * rot2 + d2 * Om / (R * (1 + Y**2)) * rot3
            c203 = d3 * Om / (R * (1 + Y**2)) * rot3
            c301 = -(d3**-1) * (d1 * c101 + d2 * c201 + R / dsq * Om * (1 + X**2) * (cphi * calp - Y * sphi))
            c302 = -(d3**-1) * (d1 * c102 + d2 * c202 + R / dsq * Om * (1 + Y**2) * (cphi * salp + X * sphi))
            c303 = -d1 * Om / (R * (1 + X**2)) * rot2 - d2 * Om / (R * (1 + Y**2)) * rot3
            scale, half = (2 / dx, 2 / dy, 2 / de), (dx / 2, dy / 2, de / 2)
            for i, row in enumerate(((c101, c102, c103), (c201, c202, c203), (c301, c302, c303))):
                for j in range(3):
                    O["christoffel"][i, j, kv, rows] = blocked(row[j] * (scale[i] * half[j]))
        # interfaces: face f fills the minus slot of padded element f+1 and the plus slot of padded element f
        lo, hi = slice(0, n2), slice(n2, None)
        ti, si = both(lambda a: perm(a, 1, 3, 0, 2).reshape(nr, Hi + 1, n2), hci, sgi)          # point (kl, jl)
        tj, sj = both(lambda a: perm(a.reshape(n, nr + 1, Hi, n), 1, 2, 0, 3).reshape(nr + 1, Hi, n2), hcj, sgj)  # (kl, il)
        tk, sk = both(lambda a: perm(a.reshape(2, nr, n, Hi, n), 0, 1, 3, 2, 4).reshape(2, nr, Hi, n2), hck, sgk)  # (jl, il)
        for val in (ti, si):
            put("_itf_i", (kv, rows, slice(1, None), lo), val)
            put("_itf_i", (kv, rows, slice(0, -1), hi), val)
        for val in (tj, sj):
            put("_itf_j", (kv, slice(r0 + 1, r1 + 2), slice(None), lo), val)
            put("_itf_j", (kv, slice(r0, r1 + 1), slice(None), hi), val)
        for f in (0, 1):
            for val in ({ab: v[f] for ab, v in tk.items()}, sk[f]):
                put("_itf_k", (kv + f + 1, rows, slice(None), lo), val)
                put("_itf_k", (kv + f, rows, slice(None), hi), val)

    if rows_per_block is None:
        rows_per_block = Hj if use_torch else 1
    blocks = [(kv, r0, min(Hj, r0 + rows_per_block)) for kv in range(V) for r0 in range(0, Hj, rows_per_block)]
    nthreads = 1 if use_torch else (threads if threads is not None else min(4, os.cpu_count() or 1))
    if nthreads > 1 and len(blocks) > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            list(pool.map(block, blocks))
    else:
        for bl in blocks:
            block(bl)
    O["boundary_sn"], O["boundary_we"] = on(t.boundary_sn.copy()), on(t.boundary_we.copy())
    return O


def schar_damping_fields(t: CubedSphere3DTile, shear: Optional[bool] = None) -> Dict[str, numpy.ndarray]:
    """The static factors of the Rayleigh sponge of DCMIP 2-1 / 2-2 (init/dcmip.py:676-757), as the RHS
    kernel takes them:  forcing_i += damp_coef * rho * (u^i - damp_uref_i)  above 20 km.
    damp_coef (V, H, H, n^3); damp_uref (3, ...) = contravariant reference wind (zero where the sponge is off)."""
    if shear is None:
        shear = t.case_number == 22
    T0, Ueq, Zh, tau0, gravity = 300.0, 20.0, 20000.0, 25.0, 9.80616
    Cs = 2.5e-4 if shear else 0.0
    Xg, Yg = numpy.meshgrid(numpy.tan(t.x1), numpy.tan(t.x2))
    _, lat = t.lonlat(Xg, Yg)
    z = t.heights()["int"]
    X, Y, lat = Xg[None], Yg[None], lat[None]
    coef = 1.0 / tau0 * numpy.sin(numpy.pi / 2 * (z - Zh) / (t.ztop - Zh)) ** 2
    coef = numpy.where(z <= Zh, 0.0, coef)
    Tref = T0 * (1 - Cs * Ueq**2 / gravity * numpy.sin(lat) ** 2)
    uref = Ueq * numpy.cos(lat) * (2 * T0 / Tref * Cs * z + Tref / T0) ** 0.5
    # wind2contra_2d with v = 0 (cubed_sphere_3d.py:1033-1104)
    rad = (t.earth_radius + z) if t.deep else t.earth_radius
    lambda_dot = uref / (rad * numpy.cos(lat))
    ct, st, ca, sa = math.cos(t.lat_p), math.sin(t.lat_p), math.cos(t.angle_p), math.sin(t.angle_p)
    dx1dlon = ct * ca + (X * Y * ct * sa - Y * st) / (1.0 + X**2)
    dx2dlon = (X * Y * ct * ca + X * st) / (1.0 + Y**2) + ct * sa
    u1 = dx1dlon * lambda_dot * 2.0 / t.delta_x1
    u2 = dx2dlon * lambda_dot * 2.0 / t.delta_x2
    on = coef != 0.0
    uref3 = numpy.stack((numpy.where(on, u1, 0.0), numpy.where(on, u2, 0.0), numpy.zeros_like(coef)))
    return {"damp_coef": numpy.ascontiguousarray(t.to_blocked(coef)), "damp_uref": numpy.ascontiguousarray(t.to_blocked(uref3))}


def metric3d_torch(t: CubedSphere3DTile, device, **kw):
    """metric3d on `device` (plus the sponge fields of cases 21/22): what Euler3DPlan takes as `metric`."""
    import torch

    m = metric3d(t, device=device, **kw)
    if t.case_number in (21, 22):
        m.update({k_: torch.from_numpy(v).to(device) for k_, v in schar_damping_fields(t).items()})
    return m
