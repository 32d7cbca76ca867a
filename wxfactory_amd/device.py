"""HipDevice: the MI355X counterpart of the reference's Device / CudaDevice.

Mirrors reference wx_factory/device/device.py:16-68 (Device ABC) and :125-220 (CudaDevice):
attributes `comm`, `xp`, `xalg`, `pde`; methods `synchronize`, `array`, `pinned`, `to_host`,
`timestamp`, `elapsed`, `has_128_bits_float`.  Differences that follow from the platform:
  * `xp` is torch (ROCm) - device memory, streams and collectives are torch's job here;
  * `pde` is a namespace over libwxhip.so exposing the reference's compiled-module functions under
    their own names (pde/interface.cpp:282-302), taking torch tensors where the reference takes
    CuPy arrays; work is enqueued on torch's current HIP stream, never the default stream;
  * one GPU per rank, device = rank % visible devices (CudaDevice picks the same way, :162-168).
Raises ValueError when no GPU is visible, which is what Simulation._make_device catches to fall
back to the CPU device (simulation/simulation.py:197-203).
"""
from time import time
from typing import List

import torch

from . import _lib

_DT = {torch.float64: _lib.WX_F64, torch.complex128: _lib.WX_C128}


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _same(*ts):
    t0 = ts[0]
    for t in ts:
        if t.dtype != t0.dtype or t.device != t0.device or not t.is_contiguous() or not t.is_cuda:
            raise TypeError("pde kernels need contiguous GPU tensors of one dtype (float64 or complex128)")
    if t0.dtype not in _DT:
        raise TypeError(f"unsupported dtype {t0.dtype}: float64 or complex128 (interface.cu:187-210 is silent here)")
    return _DT[t0.dtype]


class HipPde:
    """The four names of the reference's `pde_cpp` / `pde_cuda` modules over the C ABI."""

    def __init__(self):
        self.lib = _lib.load()

    def pointwise_eulercartesian_2d(self, q, flux_x1, flux_x3, num_elem_x1, num_elem_x3, num_solpts_tot):
        dt = _same(q, flux_x1, flux_x3)
        _lib.check(self.lib.wx_pointwise_eulercartesian_2d(q.data_ptr(), flux_x1.data_ptr(), flux_x3.data_ptr(),
                                                           num_elem_x1, num_elem_x3, num_solpts_tot, dt, _stream(q)),
                   "pointwise_eulercartesian_2d")

    def riemann_eulercartesian_ausm_2d(self, q_itf_x1, q_itf_x3, flux_itf_x1, flux_itf_x3, num_elem_x1, num_elem_x3,
                                       num_solpts):
        dt = _same(q_itf_x1, q_itf_x3, flux_itf_x1, flux_itf_x3)
        _lib.check(self.lib.wx_riemann_eulercartesian_ausm_2d(q_itf_x1.data_ptr(), q_itf_x3.data_ptr(),
                                                              flux_itf_x1.data_ptr(), flux_itf_x3.data_ptr(),
                                                              num_elem_x1, num_elem_x3, num_solpts, dt, _stream(q_itf_x1)),
                   "riemann_eulercartesian_ausm_2d")

    def forcing_euler_cubesphere_3d(self, q, pressure, sqrt_g, h, christoffel, forcing, num_elem_x1, num_elem_x2,
                                    num_elem_x3, num_solpts, verbose=0):
        dt = _same(q, pressure, forcing)
        for m in (sqrt_g, h, christoffel):
            if m.dtype != torch.float64 or not m.is_contiguous() or m.device != q.device:
                raise TypeError("metric arrays must be contiguous float64 on the state's device")
        _lib.check(self.lib.wx_forcing_euler_cubesphere_3d(q.data_ptr(), pressure.data_ptr(), sqrt_g.data_ptr(),
                                                           h.data_ptr(), christoffel.data_ptr(), forcing.data_ptr(),
                                                           num_elem_x1, num_elem_x2, num_elem_x3, num_solpts, dt,
                                                           _stream(q)), "forcing_euler_cubesphere_3d")

    def pointwise_euler_cubedsphere_3d(self, *args, **kwargs):
        """A no-op in the reference too: its kernel call is commented out (interface.cpp:119)."""
        return None


class HipDevice:
    def __init__(self, comm=None, rank: int = 0):
        self.lib = _lib.load()
        n = self.lib.wx_device_count()
        if n <= 0 or not torch.cuda.is_available():
            raise ValueError("No HIP device visible")  # -> "Switching to CPU" in Simulation._make_device
        self.comm = comm
        self.index = rank % n
        self.device = torch.device("cuda", self.index)
        torch.cuda.set_device(self.device)
        self.xp = torch
        self.xalg = torch.linalg
        self.pde = HipPde()
        self.main_stream = torch.cuda.current_stream(self.device)
        self.copy_stream = torch.cuda.Stream(self.device)  # CudaDevice.copy_stream (:174-175)

    def synchronize(self, copy_stream: bool = False, **kwargs):
        (self.copy_stream if copy_stream else torch.cuda.current_stream(self.device)).synchronize()

    def array(self, a, *args, **kwargs):
        return torch.as_tensor(a).to(self.device, non_blocking=False).contiguous()

    def pinned(self, *args, dtype=torch.float64, **kwargs):
        return torch.empty(*args, dtype=dtype, pin_memory=True)

    def to_host(self, val, **kwargs):
        return val.cpu().numpy() if isinstance(val, torch.Tensor) else val

    def timestamp(self, **kwargs):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def elapsed(self, timestamps: List) -> List[float]:
        """Seconds between consecutive timestamps + total (device.py:210-220); waits for the last event."""
        timestamps[-1].synchronize()
        out = [timestamps[i].elapsed_time(timestamps[i + 1]) * 1e-3 for i in range(len(timestamps) - 1)]
        out.append(timestamps[0].elapsed_time(timestamps[-1]) * 1e-3)
        return out

    def has_128_bits_float(self) -> bool:
        return False


def host_timestamp():
    return time()
