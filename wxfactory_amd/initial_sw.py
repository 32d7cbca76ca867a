"""Initial states of the shallow-water test cases on a 2-D cubed-sphere tile (setup-time NumPy; element-blocked layout).

What the reference builds in wx_factory/init/shallow_water_test.py and init/init_state_vars.py for `equations =
shallow_water`, restated for `geometry.CubedSphereTile2D`:
  * Williamson et al. (1992) case 2 - steady zonal geostrophic flow            shallow_water_test.py:141-160
  * case 5 - zonal flow over an isolated mountain, with the topography arrays the RHS reads (surface height at the
    nodes and on the element faces, its two DFR derivatives)                   shallow_water_test.py:163-226
  * case 6 - Rossby-Haurwitz wave, wavenumber 4                                shallow_water_test.py:229-285
Cases 5 and 6 are pinned against the states the reference itself produced (tests/golden/sw_c5_*, sw_c6_*:
tests/test_initial_sw.py), on the rotated grid of those fixtures, all six panels.
  * Galewsky et al. (2004) barotropic jet: OWN implementation.  The reference's case_galewsky
    (shallow_water_test.py:288-365) cannot run (`v` is read before it is set and the loop body indexes a
    stale array), so there is nothing to pin against; this one follows the paper: the zonal jet u(lat) of its eq. (2),
    the height in gradient-wind balance with it by Gauss-Legendre quadrature of eq. (3) (mean depth 10 km through h0),
    the bump of eq. (4).  Tests: the jet alone is a steady state of the discrete equations to truncation error (and
    the error falls with the order), the global mean depth is 10 km, the bump has the paper's amplitude and place.
State layout: (3, H, H, n^2) = (h, h u^1, h u^2) with contravariant velocities in reference-element units, as
rhs/rhs_sw.py expects.
"""
import math
from typing import Dict, Optional, Tuple

import numpy

from .geometry import CubedSphereTile2D

GRAVITY = 9.80616          # common/definitions.py:11
DAY_IN_SECS = 86400.0      # common/definitions.py


def lonlat(g: CubedSphereTile2D, X, Y) -> Tuple[numpy.ndarray, numpy.ndarray]:
    """(lon in [0, 2 pi), lat) of gnomonic points (X, Y) of the tile's panel (cubed_sphere_2d.py:378-447, sphere.py:25-45)."""
    cl, sl = math.cos(g.lon_p), math.sin(g.lon_p)
    ct, st = math.cos(g.lat_p), math.sin(g.lat_p)
    ca, sa = math.cos(g.angle_p), math.sin(g.angle_p)
    s = g.earth_radius / numpy.sqrt(1.0 + X**2 + Y**2)
    cx = s * (cl * ct + X * (cl * st * sa - sl * ca) - Y * (cl * st * ca + sl * sa))
    cy = s * (sl * ct + X * (sl * st * sa + cl * ca) - Y * (sl * st * ca - cl * sa))
    cz = s * (st - X * ct * sa + Y * ct * ca)
    lon = numpy.arctan2(cy, cx)
    lon = numpy.where(lon < 0.0, lon + 2.0 * math.pi, lon)
    return lon, numpy.arctan2(cz, numpy.hypot(cx, cy))


def wind2contra(g: CubedSphereTile2D, u, v, lat=None):
    """Zonal / meridional wind (m/s) at the nodes -> contravariant components in reference-element units
    (cubed_sphere_2d.py:564-627)."""
    X, Y = g.X, g.Y
    if lat is None:
        lat = lonlat(g, X, Y)[1]
    lambda_dot = u / (g.earth_radius * numpy.cos(lat))
    phi_dot = v / g.earth_radius
    ct, st, ca, sa = math.cos(g.lat_p), math.sin(g.lat_p), math.cos(g.angle_p), math.sin(g.angle_p)
    denom = numpy.sqrt((ct + X * st * sa - Y * st * ca) ** 2 + (X * ca + Y * sa) ** 2)
    d2 = 1.0 + X**2 + Y**2
    dx1dlon = ct * ca + (X * Y * ct * sa - Y * st) / (1.0 + X**2)
    dx2dlon = (X * Y * ct * ca + X * st) / (1.0 + Y**2) + ct * sa
    dx1dlat = -d2 * ((ct * sa + X * st) / (1.0 + X**2)) / denom
    dx2dlat = d2 * ((ct * ca - Y * st) / (1.0 + Y**2)) / denom
    u1 = (dx1dlon * lambda_dot + dx1dlat * phi_dot) * 2.0 / g.delta_x1
    u2 = (dx2dlon * lambda_dot + dx2dlat * phi_dot) * 2.0 / g.delta_x2
    return u1, u2


def _state(h, u1, u2) -> numpy.ndarray:
    return numpy.ascontiguousarray(numpy.stack((h, h * u1, h * u2)), dtype=numpy.float64)


def williamson2(g: CubedSphereTile2D) -> numpy.ndarray:
    """Steady nonlinear zonal geostrophic flow: u = u0 cos(lat), g h = g h0 - (a Omega u0 + u0^2 / 2) sin^2(lat)."""
    lon, lat = lonlat(g, g.X, g.Y)
    u0 = 2.0 * math.pi * g.earth_radius / (12.0 * DAY_IN_SECS)
    h = (29400.0 - (g.earth_radius * g.rotation_speed * u0 + 0.5 * u0**2) * numpy.sin(lat) ** 2) / GRAVITY
    u1, u2 = wind2contra(g, u0 * numpy.cos(lat), 0.0, lat)
    return _state(h, u1, u2)


def _dfr_derivatives(g: CubedSphereTile2D, f, f_itf_i, f_itf_j, diff_solpt, correction):
    """d f / d(reference coordinate) in both directions of a nodal field with given face values: the interior
    derivative plus the boundary columns applied to the faces (operators.py:190-208 with the 1-D pieces)."""
    n, H = g.n, g.H
    fe = f.reshape(H, H, n, n)                                     # [ej, ei, jl, il]
    D, C = numpy.asarray(diff_solpt), numpy.asarray(correction)    # (n, n): d/dx at node r from node c; (n, 2): from the two faces
    # faces of element (ej, ei): west = plus slot of padded index ei, east = minus slot of padded index ei + 2
    west, east = f_itf_i[:, :-2, n:], f_itf_i[:, 2:, :n]           # (H, H, n): indexed by jl
    south, north = f_itf_j[:-2, :, n:], f_itf_j[2:, :, :n]         # (H, H, n): indexed by il
    d1 = numpy.einsum("rc,abjc->abjr", D, fe) + west[..., None] * C[:, 0] + east[..., None] * C[:, 1]
    d2 = (numpy.einsum("rc,abci->abri", D, fe) + south[:, :, None, :] * C[:, 0][None, None, :, None]
          + north[:, :, None, :] * C[:, 1][None, None, :, None])
    return d1.reshape(H, H, n * n), d2.reshape(H, H, n * n)


def williamson5(g: CubedSphereTile2D, diff_solpt, correction) -> Tuple[numpy.ndarray, Dict[str, numpy.ndarray]]:
    """Zonal flow (20 m/s) over an isolated conical mountain of 2000 m at (3 pi / 2, pi / 6); returns the state and the
    topography arrays of the RHS (`wx_sw_topography`: hsurf, dzdx1, dzdx2, hsurf_itf_i, hsurf_itf_j).
    diff_solpt (n, n), correction (n, 2): the 1-D DFR pieces (operators.py:86-99, 144-148)."""
    u0, h0, hs0, rr = 20.0, 5960.0, 2000.0, math.pi / 9.0
    lon_m, lat_m = 3.0 * math.pi / 2.0, math.pi / 6.0
    lon, lat = lonlat(g, g.X, g.Y)
    u1, u2 = wind2contra(g, u0 * numpy.cos(lat), 0.0, lat)
    h_star = (GRAVITY * h0 - (g.earth_radius * g.rotation_speed * u0 + 0.5 * u0**2) * numpy.sin(lat) ** 2) / GRAVITY

    def cone(lo, la):
        return hs0 * (1.0 - numpy.sqrt(numpy.minimum(rr**2, (lo - lon_m) ** 2 + (la - lat_m) ** 2)) / rr)

    hsurf = cone(lon, lat)
    hs_i = cone(*lonlat(g, g.X_itf_i, g.Y_itf_i))
    hs_j = cone(*lonlat(g, g.X_itf_j, g.Y_itf_j))
    n = g.n
    hs_i[:, 0, :n] = 0.0      # the two outermost slots of the padded layout hold nothing (shallow_water_test.py:204-207)
    hs_i[:, -1, n:] = 0.0
    hs_j[0, :, :n] = 0.0
    hs_j[-1, :, n:] = 0.0
    dzdx1, dzdx2 = _dfr_derivatives(g, hsurf, hs_i, hs_j, diff_solpt, correction)
    topo = {"hsurf": hsurf, "dzdx1": dzdx1, "dzdx2": dzdx2, "hsurf_itf_i": hs_i, "hsurf_itf_j": hs_j}
    return _state(h_star - hsurf, u1, u2), {k: numpy.ascontiguousarray(v) for k, v in topo.items()}


def williamson6(g: CubedSphereTile2D) -> numpy.ndarray:
    """Rossby-Haurwitz wave of wavenumber R = 4 (omega = K = 7.848e-6 1/s, h0 = 8000 m)."""
    R, om, K, h0 = 4, 7.848e-6, 7.848e-6, 8000.0
    a, Om = g.earth_radius, g.rotation_speed
    lon, lat = lonlat(g, g.X, g.Y)
    c, s = numpy.cos(lat), numpy.sin(lat)
    A = om / 2.0 * (2.0 * Om + om) * c**2 + K**2 / 4.0 * c ** (2 * R) * (
        (R + 1) * c**2 + (2.0 * R**2 - R - 2.0) - 2.0 * R**2 * c ** (-2))
    B = 2.0 * (Om + om) * K / ((R + 1) * (R + 2)) * c**R * ((R**2 + 2 * R + 2) - (R + 1) ** 2 * c**2)
    C = K**2 / 4.0 * c ** (2 * R) * ((R + 1) * c**2 - (R + 2.0))
    h = h0 + (a**2 * A + a**2 * B * numpy.cos(R * lon) + a**2 * C * numpy.cos(2.0 * R * lon)) / GRAVITY
    u = a * om * c + a * K * c ** (R - 1) * (R * s**2 - c**2) * numpy.cos(R * lon)
    v = -a * K * R * c ** (R - 1) * s * numpy.sin(R * lon)
    u1, u2 = wind2contra(g, u, v, lat)
    return _state(h, u1, u2)


# ---- Galewsky, Scott & Polvani (2004), Tellus 56A: an initial-value problem for testing numerical models of the global
# shallow-water equations.  Own implementation (see the module docstring).
GALEWSKY_UMAX = 80.0
GALEWSKY_LAT0 = math.pi / 7.0
GALEWSKY_LAT1 = math.pi / 2.0 - math.pi / 7.0


def galewsky_jet(lat):
    """u(lat) of eq. (2): u_max / e_n * exp(1 / ((lat - lat0)(lat - lat1))) between lat0 and lat1, zero outside."""
    lat = numpy.asarray(lat, dtype=float)
    en = math.exp(-4.0 / (GALEWSKY_LAT1 - GALEWSKY_LAT0) ** 2)
    inside = (lat > GALEWSKY_LAT0) & (lat < GALEWSKY_LAT1)
    x = numpy.where(inside, (lat - GALEWSKY_LAT0) * (lat - GALEWSKY_LAT1), -1.0)
    return numpy.where(inside, GALEWSKY_UMAX / en * numpy.exp(1.0 / x), 0.0)


def _balance_integral(lat, earth_radius, rotation_speed, order: int = 32, pieces: int = 8):
    """int_{lat0}^{lat} a u (2 Omega sin + u tan / a) dlat' at every entry of `lat` (the integrand vanishes outside the
    jet): composite Gauss-Legendre quadrature on [lat0, min(lat, lat1)], `pieces` x `order` points - the integrand is
    C-infinity but flat to all orders at both ends, which a composite rule handles to rounding (8 x 32 against 32 x 64
    points: 1e-15 relative)."""
    lat = numpy.asarray(lat, dtype=float)
    hi = numpy.clip(lat, GALEWSKY_LAT0, GALEWSKY_LAT1)
    x, w = numpy.polynomial.legendre.leggauss(order)
    out = numpy.zeros_like(lat)
    width = (hi - GALEWSKY_LAT0) / pieces
    for k in range(pieces):
        lo_k = GALEWSKY_LAT0 + k * width
        pts = lo_k[..., None] + 0.5 * width[..., None] * (x + 1.0)
        u = galewsky_jet(pts)
        f = earth_radius * u * (2.0 * rotation_speed * numpy.sin(pts) + u * numpy.tan(pts) / earth_radius)
        out += 0.5 * width * (f * w).sum(axis=-1)
    return out


def galewsky_h0(earth_radius: float, rotation_speed: float, mean_depth: float = 10000.0) -> float:
    """The constant of eq. (3) that makes the global mean layer depth `mean_depth` (the paper's 10 km)."""
    x, w = numpy.polynomial.legendre.leggauss(400)
    lat = 0.5 * math.pi * x                      # mean over the sphere = 1/2 int h cos(lat) dlat
    integral = _balance_integral(lat, earth_radius, rotation_speed)
    mean_deficit = 0.5 * (0.5 * math.pi) * float((w * numpy.cos(lat) * integral).sum()) / GRAVITY
    return mean_depth + mean_deficit


def galewsky(g: CubedSphereTile2D, perturbation: bool = True, h0: Optional[float] = None) -> numpy.ndarray:
    """The barotropically unstable mid-latitude jet in gradient-wind balance, plus (perturbation=True) the height bump
    120 m cos(lat) exp(-(lon / alpha)^2) exp(-((lat2 - lat) / beta)^2), alpha = 1/3, beta = 1/15, lat2 = pi / 4,
    lon in (-pi, pi]."""
    lon, lat = lonlat(g, g.X, g.Y)
    if h0 is None:
        h0 = galewsky_h0(g.earth_radius, g.rotation_speed)
    h = h0 - _balance_integral(lat, g.earth_radius, g.rotation_speed) / GRAVITY
    if perturbation:
        lon_c = numpy.where(lon > math.pi, lon - 2.0 * math.pi, lon)
        h = h + 120.0 * numpy.cos(lat) * numpy.exp(-((lon_c * 3.0) ** 2)) * numpy.exp(-(((math.pi / 4.0 - lat) * 15.0) ** 2))
    u1, u2 = wind2contra(g, galewsky_jet(lat), 0.0, lat)
    return _state(h, u1, u2)


def initial_state_sw(g: CubedSphereTile2D, case_number: int, diff_solpt=None, correction=None):
    """(Q, topography or None) for `case_number` as init/init_state_vars.py selects it (2, 5, 6; 8 = Galewsky)."""
    if case_number == 2:
        return williamson2(g), None
    if case_number == 5:
        if diff_solpt is None or correction is None:
            raise ValueError("case 5 needs the 1-D DFR pieces diff_solpt and correction for the topography derivatives")
        return williamson5(g, diff_solpt, correction)
    if case_number == 6:
        return williamson6(g), None
    if case_number == 8:
        return galewsky(g), None
    raise ValueError(f"no shallow-water initial state for case {case_number}")
