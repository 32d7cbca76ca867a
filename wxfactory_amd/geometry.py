"""Setup-time geometry and metric of the equiangular gnomonic cubed sphere (shallow-water / 2-D case).

Restates, as NumPy setup code (runs once; its outputs are the static inputs of the RHS kernels):
  * Gauss-Legendre solution points                 reference geometry/quadrature.py:11-72, geometry.py:18-41
  * tile coordinates, halo-padded interface layout geometry/cubed_sphere_2d.py:88-200, 318-338, 472-562
  * rotated-grid panel centres (lon_p, lat_p, angle_p)  cubed_sphere_2d.py:216-268
  * the 2-D metric: sqrt(g), contravariant metric, the eight Christoffel symbols rhs_sw.py reads
                                                   geometry/metric2d.py:17-167
Pinned in tests/test_geometry.py against the metric arrays of the reference-generated shallow-water
fixtures (grid rotated by phi0 = pi/4, every panel).  The 3-D topography-following metric
(geometry/metric3d.py) is not restated yet.

With X = tan x1, Y = tan x2, delta^2 = 1 + X^2 + Y^2, a = earth radius, the equiangular gnomonic metric is
  sqrt(g) = a^2 (1+X^2)(1+Y^2) / delta^3,   H^11 = delta^2 / (a^2 (1+X^2)),   H^22 = delta^2 / (a^2 (1+Y^2)),
  H^12 = delta^2 X Y / (a^2 (1+X^2)(1+Y^2)),
then everything is rescaled to the reference element (each coordinate in [-1, 1] per element).
"""
import math
from typing import Dict, Tuple

import numpy

EARTH_RADIUS = 6371220.0   # cubed_sphere_2d.py:274
ROTATION_SPEED = 7.29212e-5  # cubed_sphere_2d.py:275


def gauss_legendre(n: int) -> Tuple[numpy.ndarray, numpy.ndarray]:
    x, w = numpy.polynomial.legendre.leggauss(n)
    return x, w


def panel_centre(panel: int, lambda0: float, phi0: float, alpha0: float) -> Tuple[float, float, float]:
    """(lon_p, lat_p, angle_p) of a panel for a grid whose panel 0 is centred at (lambda0, phi0) and
    turned by alpha0 (cubed_sphere_2d.py:216-268)."""
    c1, c2, c3 = math.cos(lambda0), math.cos(phi0), math.cos(alpha0)
    s1, s2, s3 = math.sin(lambda0), math.sin(phi0), math.sin(alpha0)
    if panel == 0:
        return lambda0, phi0, alpha0
    if panel == 1:
        return (math.atan2(s1 * s2 * s3 + c1 * c3, c1 * s2 * s3 - s1 * c3), -math.asin(c2 * s3), math.atan2(s2, c2 * c3))
    if panel == 2:
        return math.atan2(-s1, -c1), -phi0, -math.atan2(s3, c3)
    if panel == 3:
        return (math.atan2(-s1 * s2 * s3 - c1 * c3, -c1 * s2 * s3 + s1 * c3), math.asin(c2 * s3), -math.atan2(s2, c2 * c3))
    polar = abs(phi0) < 1e-13 and abs(alpha0) < 1e-13
    if panel == 4:
        if polar:
            return 0.0, math.pi / 2.0, -lambda0
        return (math.atan2(-s1 * s2 * c3 + c1 * s3, -c1 * s2 * c3 - s1 * s3), math.asin(c2 * c3), math.atan2(c2 * s3, -s2))
    if panel == 5:
        if polar:
            return 0.0, -math.pi / 2.0, lambda0
        return (math.atan2(s1 * s2 * c3 - c1 * s3, c1 * s2 * c3 + s1 * s3), -math.asin(c2 * c3), math.atan2(c2 * s3, s2))
    raise ValueError(f"Invalid panel number {panel}")


class CubedSphereTile2D:
    """One tile (k x k per panel) of the 2-D cubed-sphere grid in the element-blocked layout."""

    def __init__(self, n: int, H: int, panel: int, row: int = 0, col: int = 0, k: int = 1, lambda0: float = 0.0,
                 phi0: float = 0.0, alpha0: float = 0.0):
        self.n, self.H, self.panel = n, H, panel
        pts, self.glweights = gauss_legendre(n)
        self.solution_points = pts
        width = (math.pi / 2) / k
        x1_lo = -math.pi / 4 + col * width
        x2_lo = -math.pi / 4 + row * width
        self.delta_x1 = self.delta_x2 = width / H
        ref = self.delta_x1 / 2.0 * (1.0 + pts)  # solution points of one element, from its lower edge
        self.x1 = numpy.repeat(x1_lo + self.delta_x1 * numpy.arange(H), n) + numpy.tile(ref, H)
        self.x2 = numpy.repeat(x2_lo + self.delta_x2 * numpy.arange(H), n) + numpy.tile(ref, H)
        self.x1_itf = numpy.linspace(x1_lo, x1_lo + width, H + 1)
        self.x2_itf = numpy.linspace(x2_lo, x2_lo + width, H + 1)
        self.lon_p, self.lat_p, self.angle_p = panel_centre(panel, lambda0, phi0, alpha0)
        self.earth_radius, self.rotation_speed = EARTH_RADIUS, ROTATION_SPEED

        tx1, tx2 = numpy.tan(self.x1), numpy.tan(self.x2)
        self.boundary_sn = tx1.copy()  # X along the south/north edges
        self.boundary_we = tx2.copy()  # Y along the west/east edges
        # solution points, element-blocked (H, H, n^2), point p = jl*n + il
        self.X = self._blocked(numpy.broadcast_to(tx1[None, :], (H * n, H * n)))
        self.Y = self._blocked(numpy.broadcast_to(tx2[:, None], (H * n, H * n)))
        # west/east faces: X = tan(x1_itf[face]), Y = tan(x2 of the line); padded (H, H+2, 2n)
        self.X_itf_i = self._pad_i(numpy.broadcast_to(numpy.tan(self.x1_itf)[None, :, None], (H, H + 1, n)))
        self.Y_itf_i = self._pad_i(numpy.broadcast_to(tx2.reshape(H, 1, n), (H, H + 1, n)))
        # south/north faces: padded (H+2, H, 2n)
        self.X_itf_j = self._pad_j(numpy.broadcast_to(tx1.reshape(1, H, n), (H + 1, H, n)))
        self.Y_itf_j = self._pad_j(numpy.broadcast_to(numpy.tan(self.x2_itf)[:, None, None], (H + 1, H, n)))

    def _blocked(self, a):
        n, H = self.n, self.H
        return numpy.ascontiguousarray(a.reshape(H, n, H, n).transpose(0, 2, 1, 3).reshape(H, H, n * n))

    def _pad_i(self, face):
        """(H, H+1 faces, n) -> (H, H+2, 2n): plus slot of index f and minus slot of index f+1 hold face f;
        the two outermost slots are zero (cubed_sphere_2d.py:519-536)."""
        n, H = self.n, self.H
        out = numpy.zeros((H, H + 2, 2 * n))
        out[:, :-1, n:] = face
        out[:, 1:, :n] = face
        return out

    def _pad_j(self, face):
        n, H = self.n, self.H
        out = numpy.zeros((H + 2, H, 2 * n))
        out[:-1, :, n:] = face
        out[1:, :, :n] = face
        return out


def metric2d(g: CubedSphereTile2D) -> Dict[str, numpy.ndarray]:
    """The arrays wx_sw_metric needs (minus topography), named as in include/wxhip.h (metric2d.py:17-167)."""
    a2 = g.earth_radius**2
    sx = 4.0 / (g.delta_x1**2)
    sxy = 4.0 / (g.delta_x1 * g.delta_x2)
    sy = 4.0 / (g.delta_x2**2)
    jac = g.delta_x1 * g.delta_x2 / 8.0

    def sqrt_g(X, Y):
        d2 = 1.0 + X**2 + Y**2
        return a2 * (1.0 + X**2) * (1.0 + Y**2) / (d2 * numpy.sqrt(d2)) * jac

    def h11(X, Y):
        return (1.0 + X**2 + Y**2) / (a2 * (1.0 + X**2)) * sx

    def h12(X, Y):
        return (1.0 + X**2 + Y**2) * X * Y / (a2 * (1.0 + X**2) * (1.0 + Y**2)) * sxy

    def h22(X, Y):
        return (1.0 + X**2 + Y**2) / (a2 * (1.0 + Y**2)) * sy

    X, Y = g.X, g.Y
    d2 = 1.0 + X**2 + Y**2
    gridrot = (math.sin(g.lat_p) - X * math.cos(g.lat_p) * math.sin(g.angle_p)
               + Y * math.cos(g.lat_p) * math.cos(g.angle_p))
    om = g.rotation_speed
    m = {
        "sqrtG": sqrt_g(X, Y), "H_contra_11": h11(X, Y), "H_contra_12": h12(X, Y), "H_contra_21": h12(X, Y),
        "H_contra_22": h22(X, Y),
        "christoffel_1_01": om * X * Y / d2 * gridrot,
        "christoffel_1_02": -om * (1.0 + Y**2) / d2 * gridrot,
        "christoffel_2_01": om * (1.0 + X**2) / d2 * gridrot,
        "christoffel_2_02": -om * X * Y / d2 * gridrot,
        "christoffel_1_11": 2 * X * Y**2 / d2 * (0.5 * g.delta_x1),
        "christoffel_1_12": -(Y + Y**3) / d2 * (0.5 * g.delta_x1),
        "christoffel_2_12": -X * (1.0 + X**2) / d2 * (0.5 * g.delta_x2),
        "christoffel_2_22": 2.0 * X**2 * Y / d2 * (0.5 * g.delta_x2),
        "sqrtG_itf_i": sqrt_g(g.X_itf_i, g.Y_itf_i), "sqrtG_itf_j": sqrt_g(g.X_itf_j, g.Y_itf_j),
        "H_contra_11_itf_i": h11(g.X_itf_i, g.Y_itf_i), "H_contra_21_itf_i": h12(g.X_itf_i, g.Y_itf_i),
        "H_contra_12_itf_j": h12(g.X_itf_j, g.Y_itf_j), "H_contra_22_itf_j": h22(g.X_itf_j, g.Y_itf_j),
        "boundary_sn": g.boundary_sn, "boundary_we": g.boundary_we,
    }
    return {k: numpy.ascontiguousarray(v, dtype=numpy.float64) for k, v in m.items()}


def metric2d_torch(g: CubedSphereTile2D, device) -> Dict[str, "object"]:
    import torch

    return {k: torch.from_numpy(v).to(device) for k, v in metric2d(g).items()}
