"""Time steppers that call the RHS kernels with the state resident on the GPU.

Mirror the reference's callers of the path:
  Tvdrk3  integrators/tvdrk3.py:12-19   SSP-RK3, three RHS evaluations per step
  Ros2    integrators/ros2.py:24-81     Rosenbrock-2: (I - dt/2 J) dQ via matvec_rat + FGMRES
The state is one torch tensor holding this rank's panels stacked (see PanelRhs)."""
from time import time
from typing import Callable

import torch

from . import reduce as _reduce
from .matvec import ComplexStepOperator, MatvecOpRat
from .solvers import fgmres


class Tvdrk3:
    """SSP-RK3.  With a PanelRhs whose plans offer the fused stage update (rhs_axpy: the stage's
    linear combination is formed in the RHS kernel's store), a step is three RHS evaluations plus two
    extra reads of Q and nothing else; `fused=False` is the reference's literal sequence."""

    def __init__(self, rhs: Callable, fused: bool = True, pipeline: bool = True, final_filter=None, nan_flag=None):
        """final_filter: a filters.ExpFilter3D (or its n x n matrix) to apply to the new state inside the LAST
        stage's kernel (pipelined path only), with nan_flag (filters.NanFlag) raised by the same kernel: the body
        of Simulation.step (simulation.py:147-155) in three launches per panel."""
        self.rhs = rhs
        self.final_filter, self.nan_flag = final_filter, nan_flag
        self.fused = fused and bool(getattr(rhs, "supports_axpy", False))
        # stage pipeline: each stage's kernel also extrapolates its output to the faces (no separate pass)
        self.pipeline = self.fused and pipeline and bool(getattr(rhs, "supports_pipeline", False))
        self.fused_filter = False
        if final_filter is not None and self.pipeline and hasattr(rhs, "set_exp_filter"):
            rhs.set_exp_filter(getattr(final_filter, "matrix", final_filter))
            self.fused_filter = True

    def step(self, Q: torch.Tensor, dt: float) -> torch.Tensor:
        rhs = self.rhs
        small = bool(getattr(rhs, "_small_tiles", lambda: False)()) and not self.fused_filter
        # (small tiles are launch-bound: the batched stage update, two launches per stage for all tiles, beats the
        #  per-tile pipelined stages)
        if self.pipeline and not small and isinstance(Q, torch.Tensor) and Q.is_contiguous() and rhs.panels:
            Q1 = rhs.stage(Q, None, 0.0, 1.0, dt)
            Q2 = rhs.stage(Q1, Q, 0.75, 0.25, 0.25 * dt)
            if self.fused_filter:
                return rhs.stage(Q2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt, filtered=True, nan_flag=self.nan_flag)
            return rhs.stage(Q2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt)
        if self.fused:
            Q1 = rhs.axpy(Q, None, 0.0, 1.0, dt)
            Q2 = rhs.axpy(Q1, Q, 0.75, 0.25, 0.25 * dt)
            return rhs.axpy(Q2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt)
        Q1 = torch.add(Q, rhs(Q), alpha=dt)
        Q2 = 0.75 * Q + 0.25 * Q1
        Q2.add_(rhs(Q1), alpha=0.25 * dt)
        Q = (1.0 / 3.0) * Q + (2.0 / 3.0) * Q2
        Q.add_(rhs(Q2), alpha=(2.0 / 3.0) * dt)
        return Q


class Ros2:
    def __init__(self, rhs_handle: Callable, tol: float = 1e-7, gmres_restart: int = 20, verbose: int = 0,
                 ortho: str = "igs", group=None):
        self.rhs_handle, self.tol, self.gmres_restart, self.verbose = rhs_handle, tol, gmres_restart, verbose
        # who the solver's reductions run over (reduce.py): by default what the RHS object exchanges its halos on - the
        # library's communicator (backend "rccl") or its torch.distributed group
        self.group = group if group is not None else getattr(rhs_handle, "reduce_group", None)
        self.ortho = ortho   # fgmres orthogonalisation: "igs" = the reference's one-synchronisation variant, "cgs"
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
            verbose=self.verbose, ortho=self.ortho, group=self.group)
        self.solver_info = dict(flag=flag, time=time() - t0, iterations=num_iter, residuals=residuals,
                                rel_residual=norm_r / norm_b, **(getattr(fgmres, "last_stats", None) or {}))
        self.failure_flag = flag
        return Qnew.reshape(Q.shape)


class Epi:
    """Exponential propagation iterative methods, orders 2-6 (integrators/epi.py:28-360), with the
    phi-function evaluation by KIOPS (solvers/kiops.py) or PMEX (solvers/pmex.py, the schema's default) and the JVP by the RHS kernels
    (matvec_fun: complex step by default, `jacobian_method` of config-format.json)."""

    _A = {
        2: [[]],
        3: [[2 / 3]],
        4: [[-3 / 10, 3 / 40], [32 / 5, -11 / 10]],
        5: [[-4 / 5, 2 / 5, -4 / 45], [12, -9 / 2, 8 / 9], [3, 0, -1 / 3]],
        6: [[-49 / 60, 351 / 560, -359 / 1260, 367 / 6720], [92 / 7, -99 / 14, 176 / 63, -1 / 2],
            [485 / 21, -151 / 14, 23 / 9, -31 / 168]],
    }

    def __init__(self, order: int, rhs: Callable, tol: float = 1e-7, jacobian_method: str = "complex",
                 init_substeps: int = 1, init_method=None, exponential_solver: str = "kiops", group=None):
        from collections import deque

        # who the solver's reductions run over (reduce.py): by default what the RHS object exchanges its halos on
        self.group = group if group is not None else getattr(rhs, "reduce_group", None)

        if exponential_solver not in ("kiops", "pmex"):   # (the two the shipped configurations name, epi.py:314-348)
            raise ValueError(f"Unrecognized exponential solver {exponential_solver}")
        # A[k][i]: weight of the i-th previous state's remainder in phi-row first_row + k
        self.A, self.first_row, self.n_prev, self.max_phi = self._tables(order)
        self.rhs, self.tol, self.jacobian_method = rhs, tol, jacobian_method
        self.exponential_solver = exponential_solver
        # how a second sub-step of the phi solver restarts its augmented components: "reference" = as the reference
        # computes them (its results), "phipm" = exact; they differ from order 3 on, see solvers._restart_tail
        self.restart_powers = "reference"
        # Replay whole KIOPS passes as HIP graphs (one GPU, launch-bound sizes).  Off by default: measured at the size of
        # config/dcmip31.ini the adaptive basis size m changes almost every step, so passes are re-captured (8 ms each)
        # more often than replayed, and with a Krylov vector built from ONE host call (wx_euler3d_batch_kiops_vector) the
        # eager pass is GPU-bound already (tools/kiopsgraph.py).
        self.graph_passes = False
        self._force_split = False   # tests: the several-rank code paths of the phi solvers on one rank
        self._static = None
        self._ws = None
        self.krylov_size = 1
        self.previous_Q, self.previous_rhs = deque(), deque()
        self.dt = 0.0
        self.init_method = init_method if (init_method or self.n_prev == 0) else Epi(
            2, rhs, tol, jacobian_method, exponential_solver=exponential_solver)
        self.init_substeps = init_substeps
        self.solver_info = None

    def _tables(self, order: int):
        if order not in self._A:
            raise ValueError(f"Unsupported order {order} for EPI method")
        A = self._A[order]
        k = len(A) - (1 if order == 2 else 0)
        return A, 2, len(A[0]), k + 1

    def step(self, Q: torch.Tensor, dt: float) -> torch.Tensor:
        from .matvec import matvec_fun

        if self.dt and abs(self.dt - dt) > 1e-10:
            self.previous_Q.clear()
            self.previous_rhs.clear()
        self.dt = dt
        release = getattr(self.rhs, "jvp_release", None)
        if release is not None:
            release()   # the previous step's linearisation state is history (all ranks pass here together)
        if len(self.previous_Q) < self.n_prev:
            self.previous_Q.appendleft(Q)
            self.previous_rhs.appendleft(self.rhs(Q))
            for _ in range(self.init_substeps):
                Q = self.init_method.step(Q, dt / self.init_substeps)
            self.solver_info = getattr(self.init_method, "solver_info", None)
            return Q
        rhs = self.rhs(Q)
        vec = torch.zeros((self.max_phi + 1, rhs.numel()), dtype=Q.dtype, device=Q.device)
        vec[1] = rhs.flatten()
        for i in range(self.n_prev):
            JdQ = matvec_fun((self.previous_Q[i] - Q).flatten(), 1.0, Q, rhs, self.rhs, self.jacobian_method)
            r = (self.previous_rhs[i] - rhs).flatten() - JdQ
            for k, row in enumerate(self.A, start=self.first_row):
                vec[k] += row[i] * r
        return self._advance(Q, rhs, self._phi_sum(Q, rhs, vec, dt), dt)

    def _phi_sum(self, Q, rhs, vec, dt):
        """sum_k phi_k(dt J) vec[k] with the configured exponential solver; fills `solver_info`."""
        import math

        if self.exponential_solver == "pmex":
            from .solvers import pmex

            phiv, stats = pmex([1.0], ComplexStepOperator(dt, Q, rhs, self.rhs, self.jacobian_method), vec, tol=self.tol,
                               mmax=64, task1=False, restart_powers=self.restart_powers, group=self.group,
                               _force_split=self._force_split)
            self.solver_info = dict(substeps=stats[0], rejected=stats[1], iterations=stats[2], exps=stats[3],
                                    error=stats[4], krylov_size=stats[5], own_norms=stats[6])
            return phiv
        from .solvers import kiops

        ws, token, Qm, Rm = None, None, Q, rhs
        if Q.is_cuda and Q.dtype == torch.float64:
            from .solvers import KiopsWorkspace

            if self._ws is None:
                self._ws = KiopsWorkspace()   # basis, Hessenberg columns and scratch survive from step to step
            ws = self._ws
            if self.graph_passes and bool(getattr(self.rhs, "_small_tiles", lambda: False)()) \
                    and (getattr(self.rhs, "world", 1) == 1 or _reduce.capturable(self.group)):
                # replay whole Krylov passes as HIP graphs: the matvec then has to read the linearisation state from
                # fixed addresses (static copies of Q and R(Q)); a pass is re-captured whenever (j0, m) is new
                if self._static is None or self._static[0].shape != Q.shape:
                    self._static = (torch.empty_like(Q), torch.empty_like(rhs))
                Qm, Rm = self._static
                Qm.copy_(Q)
                Rm.copy_(rhs)
                token = (Qm.data_ptr(), Rm.data_ptr(), float(dt), self.jacobian_method)
        phiv, stats = kiops([1], ComplexStepOperator(dt, Qm, Rm, self.rhs, self.jacobian_method), vec,
                            tol=self.tol, m_init=self.krylov_size, mmin=16, mmax=64, task1=False, workspace=ws,
                            graph_token=token, restart_powers=self.restart_powers, group=self.group,
                            _force_split=self._force_split)
        self.krylov_size = math.floor(0.7 * stats[5] + 0.3 * self.krylov_size)
        self.solver_info = dict(substeps=stats[0], rejected=stats[1], iterations=stats[2], exps=stats[3],
                                error=stats[4], krylov_size=stats[5])
        return phiv

    def _advance(self, Q, rhs, phiv, dt):
        if self.n_prev > 0:
            self.previous_Q.pop()
            self.previous_Q.appendleft(Q)
            self.previous_rhs.pop()
            self.previous_rhs.appendleft(rhs)
        return Q + phiv.reshape(Q.shape) * dt


class EpiStiff(Epi):
    """(OUTSIDE the coverage contract, SURVEY.md section 8 / section 2 #14: config/dcmip20.ini is not a BASELINE configuration; kept because it
    exercises the fused kernels' second code path through the exponential solvers, no effort goes here.)  Stiffness-resilient EPI of order >= 2 (integrators/epi_stiff.py:14-132; `time_integrator = epi_stiff<order>`,
    config/dcmip20.ini): the remainders of the order - 2 previous states enter from phi_3 on, with the weights of
    integrators/integrator.py:135-146 for the nodes 1, 2, ..., order - 2.  Start-up by EPI2 steps; simulation.py:336-340
    builds it with init_substeps = 10."""

    def _tables(self, order: int):
        import math
        from itertools import combinations

        if order < 2:
            raise ValueError("Unsupported order for EPI method")
        c = [float(i) for i in range(1, order - 1)]
        m = len(c)
        A = [[0.0] * m for _ in range(m)]
        for i in range(m):
            others = c[:i] + c[i + 1:]
            denom = c[i] ** 2 * math.prod(c[i] - cl for cl in others)
            for k in range(m):
                esym = sum(math.prod(v) for v in combinations(others, m - k - 1))   # elementary symmetric polynomial
                A[k][i] = (-1) ** (m - k + 1) * math.factorial(k + 2) * esym / denom
        return A, 3, m, (order if order > 2 else 1)


class StepLoop:
    """The device-side body of `Simulation.step` (simulation/simulation.py:147-155):
    Q = integrator.step(Q, dt); Q = operators.apply_filters(Q, ...); NaN check (ValueError("NaN") on all ranks).

    `filt` is a filters.ExpFilter3D (or None), `nan_flag` a filters.NanFlag (or None).  When the filter owns
    the flag it raises it while filtering, so the state is not read a second time.  The flag is fetched
    (one host sync + one tiny all-reduce) every `check_every` steps; the reference does so every step."""

    def __init__(self, stepper, filt=None, nan_flag=None, check_every: int = 1):
        self.stepper, self.filt, self.nan_flag = stepper, filt, nan_flag
        self.check_every = max(1, int(check_every))
        self.step_id = 0
        if filt is not None and nan_flag is not None and filt.nan_flag is None:
            filt.nan_flag = nan_flag
        # a pipelined SSP-RK3 can filter (and NaN-check) inside its last stage's kernel
        self.fused = False
        small = bool(getattr(getattr(stepper, "rhs", None), "_small_tiles", lambda: False)())
        # (small tiles are launch-bound: batched stages + one stacked filter launch beat the per-tile fused kernels)
        if filt is not None and isinstance(stepper, Tvdrk3) and stepper.pipeline and not stepper.fused_filter and not small \
                and hasattr(stepper.rhs, "set_exp_filter") and getattr(filt, "matrix", None) is not None:
            stepper.rhs.set_exp_filter(filt.matrix)
            stepper.final_filter, stepper.nan_flag, stepper.fused_filter = filt, nan_flag, True
        if isinstance(stepper, Tvdrk3) and stepper.fused_filter:
            self.fused = True

    def step(self, Q: torch.Tensor, dt: float) -> torch.Tensor:
        Q = self.stepper.step(Q, dt)
        if self.filt is not None and not self.fused:
            Q = self.filt(Q, out=Q)  # the stepper returned fresh storage: filter it in place
            inv = getattr(getattr(self.stepper, "rhs", None), "invalidate_faces", None)
            if inv is not None:
                inv()  # the stage pipeline's prepared faces belong to the unfiltered state
        self.step_id += 1
        if self.nan_flag is not None:
            scanned = self.fused or (self.filt is not None and self.filt.nan_flag is self.nan_flag)
            if not scanned:
                self.nan_flag.check(Q)
            if self.step_id % self.check_every == 0:
                self.nan_flag.raise_if_set()
        return Q

    def run(self, Q: torch.Tensor, dt: float, nsteps: int) -> torch.Tensor:
        for _ in range(nsteps):
            Q = self.step(Q, dt)
        return Q
