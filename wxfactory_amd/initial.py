"""Initial states of the 3-D Euler test cases, in the kernels' layout (setup-time NumPy).

Restates the two DCMIP cases of BASELINE.json's configs 4 and 5 so that a run needs nothing from the reference:
  dcmip_gravity_wave   DCMIP 3-1, non-hydrostatic gravity waves on a small planet    init/dcmip.py:763-886
  dcmip_schar_waves    DCMIP 2-1 / 2-2, flow over a Schaer-type mountain             init/dcmip.py:551-673
and the assembly of the conserved state (rho, rho u^1, rho u^2, rho w, rho theta)     init/initialize.py:114-120.
Pinned in tests/test_initial.py against the unperturbed states of the reference-generated fixtures.
"""
import math

import numpy

from .geometry3d import CubedSphere3DTile

GRAVITY, P0, RD, CPD = 9.80616, 100000.0, 287.05, 1005.46   # common/definitions.py:5-12


def _nodes(t: CubedSphere3DTile):
    """(X, Y, lon, lat, z) at the nodes, grid layout (nk, nj, ni)."""
    Xg, Yg = numpy.meshgrid(numpy.tan(t.x1), numpy.tan(t.x2))
    lon, lat = t.lonlat(Xg, Yg)
    z = t.heights()["int"]
    return Xg[None], Yg[None], lon[None], lat[None], z


def wind2contra_2d(t: CubedSphere3DTile, u, v, X, Y, lat, z):
    """Zonal / meridional wind (m/s) -> contravariant components in reference-element units
    (cubed_sphere_3d.py:1033-1104)."""
    rad = (t.earth_radius + z) if t.deep else t.earth_radius
    lambda_dot = u / (rad * numpy.cos(lat))
    phi_dot = v / rad
    ct, st, ca, sa = math.cos(t.lat_p), math.sin(t.lat_p), math.cos(t.angle_p), math.sin(t.angle_p)
    denom = numpy.sqrt((ct + X * st * sa - Y * st * ca) ** 2 + (X * ca + Y * sa) ** 2)
    d2 = 1.0 + X**2 + Y**2
    dx1dlon = ct * ca + (X * Y * ct * sa - Y * st) / (1.0 + X**2)
    dx2dlon = (X * Y * ct * ca + X * st) / (1.0 + Y**2) + ct * sa
    dx1dlat = -d2 * ((ct * sa + X * st) / (1.0 + X**2)) / denom
    dx2dlat = d2 * ((ct * ca - Y * st) / (1.0 + Y**2)) / denom
    u1 = (dx1dlon * lambda_dot + dx1dlat * phi_dot) * 2.0 / t.delta_x1
    u2 = (dx2dlon * lambda_dot + dx2dlat * phi_dot) * 2.0 / t.delta_x2
    return u1, u2


def _state(t, rho, u1, u2, w, theta):
    Q = numpy.stack((rho, rho * u1, rho * u2, rho * w, rho * theta))
    return numpy.ascontiguousarray(t.to_blocked(numpy.broadcast_to(Q, (5,) + (t.nk, t.nj, t.ni))))


def dcmip_gravity_wave(t: CubedSphere3DTile) -> numpy.ndarray:
    """DCMIP 3-1 (case 31): balanced zonal flow on the R/125 planet + a potential-temperature perturbation."""
    X, Y, lon, lat, z = _nodes(t)
    u0, Teq, Peq = 20.0, 300.0, 100000.0
    lambdac, d, phic, delta_theta, Lz = 2.0 * math.pi / 3.0, 5000.0, 0.0, 1.0, 20000.0
    N2 = 0.01 * 0.01
    bigG = GRAVITY**2 / (N2 * CPD)
    kappa, inv_kappa = RD / CPD, CPD / RD
    u = u0 * numpy.cos(lat) + 0.0 * z
    u1, u2 = wind2contra_2d(t, u, numpy.zeros_like(u), X, Y, lat, z)
    spin = u0 + 2.0 * t.rotation_speed * t.earth_radius
    TS = bigG + (Teq - bigG) * numpy.exp(-(u0 * N2 / (4.0 * GRAVITY**2)) * spin * (numpy.cos(2.0 * lat) - 1.0))
    ps = Peq * numpy.exp((u0 / (4.0 * bigG * RD)) * spin * (numpy.cos(2.0 * lat) - 1.0)) * (TS / Teq) ** inv_kappa
    p = ps * ((bigG / TS) * numpy.exp(-N2 * z / GRAVITY) + 1.0 - (bigG / TS)) ** inv_kappa
    t_mean = bigG * (1.0 - numpy.exp(N2 * z / GRAVITY)) + TS * numpy.exp(N2 * z / GRAVITY)
    theta_base = t_mean * (P0 / p) ** kappa
    rho = p / (RD * t_mean)
    r = t.earth_radius * numpy.arccos(numpy.sin(lat) * math.sin(phic) + numpy.cos(lat) * math.cos(phic) * numpy.cos(lon - lambdac))
    s = d**2 / (d**2 + r**2)
    theta = theta_base + delta_theta * s * numpy.sin(2.0 * math.pi * z / Lz)
    return _state(t, rho, u1, u2, numpy.zeros_like(rho), theta)


def dcmip_schar_waves(t: CubedSphere3DTile, shear: bool = False) -> numpy.ndarray:
    """DCMIP 2-1 (case 21) / 2-2 (case 22, sheared): isothermal flow over the Schaer mountain; the tile must
    carry the mountain (geometry3d.topography_for_case)."""
    X, Y, lon, lat, z = _nodes(t)
    T0, Ueq, Peq = 300.0, 20.0, 100000.0
    Cs = 2.5e-4 if shear else 0.0
    T = T0 * (1 - Cs * Ueq**2 / GRAVITY * numpy.sin(lat) ** 2) + 0.0 * z
    p = Peq * numpy.exp(-(Ueq**2) / (2 * RD * T0) * numpy.sin(lat) ** 2 - GRAVITY * z / (RD * T))
    u = Ueq * numpy.cos(lat) * (2 * T0 / T * Cs * z + T / T0) ** 0.5
    u1, u2 = wind2contra_2d(t, u, numpy.zeros_like(u), X, Y, lat, z)   # w = 0: no vertical contribution
    rho = p / (RD * T)
    theta = T * (P0 / p) ** (RD / CPD)
    return _state(t, rho, u1, u2, numpy.zeros_like(rho), theta)


def initial_state(t: CubedSphere3DTile) -> numpy.ndarray:
    """The conserved state of the tile's test case (init/initialize.py:57-129 for the supported cases)."""
    if t.case_number == 31:
        return dcmip_gravity_wave(t)
    if t.case_number in (21, 22):
        return dcmip_schar_waves(t, shear=t.case_number == 22)
    raise ValueError(f"no initial state for case {t.case_number} (supported: 21, 22, 31)")
