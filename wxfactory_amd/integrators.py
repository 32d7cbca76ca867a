"""Time steppers that call the RHS kernels with the state resident on the GPU.

Mirror the reference's callers of the path:
  Tvdrk3  integrators/tvdrk3.py:12-19   SSP-RK3, three RHS evaluations per step
  Ros2    integrators/ros2.py:24-81     Rosenbrock-2: (I - dt/2 J) dQ via matvec_rat + FGMRES
The state is one torch tensor holding this rank's panels stacked (see PanelRhs)."""
from time import time
from typing import Callable

import torch

from .matvec import MatvecOpRat
from .solvers import fgmres


class Tvdrk3:
    def __init__(self, rhs: Callable):
        self.rhs = rhs

    def step(self, Q: torch.Tensor, dt: float) -> torch.Tensor:
        rhs = self.rhs
        Q1 = torch.add(Q, rhs(Q), alpha=dt)
        Q2 = 0.75 * Q + 0.25 * Q1
        Q2.add_(rhs(Q1), alpha=0.25 * dt)
        Q = (1.0 / 3.0) * Q + (2.0 / 3.0) * Q2
        Q.add_(rhs(Q2), alpha=(2.0 / 3.0) * dt)
        return Q


class Ros2:
    def __init__(self, rhs_handle: Callable, tol: float = 1e-7, gmres_restart: int = 20, verbose: int = 0):
        self.rhs_handle, self.tol, self.gmres_restart, self.verbose = rhs_handle, tol, gmres_restart, verbose
        self.solver_info = None
        self.failure_flag = 0

    def step(self, Q: torch.Tensor, dt: float) -> torch.Tensor:
        rhs = self.rhs_handle(Q)
        Q_flat = Q.flatten()
        A = MatvecOpRat(dt, Q, rhs, self.rhs_handle)
        b = A(Q_flat) + rhs.flatten() * dt
        t0 = time()
        Qnew, norm_r, norm_b, num_iter, flag, residuals = fgmres(
            A, b, x0=Q_flat, tol=self.tol, restart=self.gmres_restart, maxiter=20000 // self.gmres_restart,
            verbose=self.verbose)
        self.solver_info = dict(flag=flag, time=time() - t0, iterations=num_iter, residuals=residuals,
                                rel_residual=norm_r / norm_b)
        self.failure_flag = flag
        return Qnew.reshape(Q.shape)
