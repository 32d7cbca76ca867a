#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE, build-container only (needs /root/reference).  The reference
is imported in place through ``bootstrap.py``; this script only *drives* it (it
builds the reference's own objects, calls ``rhs.full(Q)`` and reads the arrays it
leaves behind) and stores inputs + expected outputs as compressed ``.npz``.
No reference source text is stored - fixtures are data.

    python oracle/refharness/gen_golden.py [case ...]

Cases are listed in CASES below.  Every fixture holds, per stored panel ``pN/``:
inputs (Q, operators' 1-D pieces, every metric array the path reads, the halo
faces received from the 4 neighbour panels) and outputs (per-phase
intermediates, final R), for real Q and for a complex-step perturbed Q.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from bootstrap import bootstrap, REF  # noqa: E402

MPI = bootstrap(6)

import numpy  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")

SCHEMA_TEXT = open(os.path.join(REF, "config", "config-format.json")).read()


def _config(ini, overrides):
    from common import Configuration, ConfigurationSchema

    cfg = Configuration(open(os.path.join(REF, "config", ini)).read(), ConfigurationSchema(SCHEMA_TEXT))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    if "num_elements_horizontal" in overrides:
        cfg.num_elements_horizontal_total = overrides["num_elements_horizontal"]
    if "num_solpts" in overrides:
        cfg.initial_num_solpts = overrides["num_solpts"]
    cfg.output_freq = 0
    cfg.save_state_freq = 0
    cfg.stat_freq = 0
    cfg.output_dir = "/tmp/wx_golden_out"
    return cfg


def _ops_1d(ops, geom):
    return {
        "ops/solution_points": numpy.asarray(geom.solutionPoints, dtype=float),
        "ops/glweights": numpy.asarray(geom.glweights, dtype=float),
        "ops/extrap_neg": numpy.asarray(ops.extrap_west, dtype=float),
        "ops/extrap_pos": numpy.asarray(ops.extrap_east, dtype=float),
        "ops/diff_solpt": numpy.asarray(ops.diff_solpt, dtype=float),
        "ops/correction": numpy.asarray(ops.correction, dtype=float),
        "ops/highfilter": numpy.asarray(ops.highfilter, dtype=float),
    }


# ---------------------------------------------------------------------------------------------
# 3-D Euler on the cubed sphere (rhs_dfr.py + pde_euler_cubesphere.py + fluxes.py)
# ---------------------------------------------------------------------------------------------
EULER_PHASE_ATTRS = [
    "q_itf_x1", "q_itf_x2", "q_itf_x3",
    "q_itf_s", "q_itf_n", "q_itf_w", "q_itf_e",
    "f_x1", "f_x2", "f_x3", "pressure", "log_p",
    "wflux_adv_x1", "wflux_adv_x2", "wflux_adv_x3",
    "wflux_pres_x1", "wflux_pres_x2", "wflux_pres_x3",
    "f_itf_x1", "f_itf_x2", "f_itf_x3",
    "pressure_itf_x1", "pressure_itf_x2", "pressure_itf_x3",
    "wflux_adv_itf_x1", "wflux_adv_itf_x2", "wflux_adv_itf_x3",
    "wflux_pres_itf_x1", "wflux_pres_itf_x2", "wflux_pres_itf_x3",
    "forcing",
]
# Small: kept for every stored panel.  Phases: only when ``phases`` is requested.
EULER_LIGHT_ATTRS = ["q_itf_s", "q_itf_n", "q_itf_w", "q_itf_e"]

EULER_METRIC_ATTRS = [
    "sqrtG_new", "inv_sqrtG_new", "h_contra_new", "christoffel", "inv_dzdeta_new",
    "sqrtG_itf_i_new", "sqrtG_itf_j_new", "sqrtG_itf_k_new",
    "h_contra_itf_i_new", "h_contra_itf_j_new", "h_contra_itf_k_new",
]


def _damping_fields(geom, metric, case_number, shape):
    """Static Rayleigh-damping fields of cases 21/22, obtained by probing the reference's
    dcmip_schar_damping with unit states: forcing_i += coef*rho*(u_i - uref_i)."""
    from init.dcmip import dcmip_schar_damping

    shear = case_number == 22
    ones = numpy.ones(shape)
    zeros = numpy.zeros(shape)
    f0 = numpy.zeros((5,) + shape)
    dcmip_schar_damping(f0, ones, zeros, zeros, zeros, metric, geom, shear=shear, new_layout=True)
    f1 = numpy.zeros((5,) + shape)
    dcmip_schar_damping(f1, ones, ones, ones, ones, metric, geom, shear=shear, new_layout=True)
    coef = f1[1] - f0[1]  # = coef * 1
    uref = numpy.zeros((3,) + shape)
    nz = coef != 0.0
    for i in range(3):
        uref[i][nz] = -f0[1 + i][nz] / coef[nz]
    return coef, uref


def euler_case(name, ini, overrides, metric_panels, phase_panels, perturb=0.0, seed=1234, monolith=False):
    cfg_probe = _config(ini, overrides)
    n = cfg_probe.num_solpts
    print(f"[{name}] {ini} n={n} H={cfg_probe.num_elements_horizontal} V={cfg_probe.num_elements_vertical}",
          flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle

        cfg = _config(ini, overrides)
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(
            cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
            cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev,
        )
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)

        rng = numpy.random.default_rng(seed + rank)
        if perturb > 0.0:
            # O(1)-RHS state: seeded relative perturbation of the balanced initial condition
            Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
            umag = numpy.abs(Q[1]).max() + 1e-12 * numpy.abs(Q[0]).max()
            Q[3] = Q[3] + Q[0] * perturb * 1e-5 * rng.uniform(-1.0, 1.0, Q[0].shape)

        out = {}
        out["Q"] = Q.copy()
        R = rhs.full(Q)
        out["R"] = R.copy()
        if monolith:
            # the reference's one-function RHS (rhs/rhs_euler.py:158-517, dead code there) on the
            # same state.  Its import of ProcessTopology from wx_mpi is stale at this commit: give the module the
            # name it asks for (SURVEY 8c's workaround; no reference file is touched).
            import process_topology
            import wx_mpi

            if not hasattr(wx_mpi, "ProcessTopology"):
                wx_mpi.ProcessTopology = process_topology.ProcessTopology
            from rhs.rhs_euler import RhsEuler

            mono = RhsEuler(Q.shape, geom, ops, metric, pt, cfg.num_solpts, cfg.num_elements_horizontal,
                            cfg.num_elements_vertical, cfg.case_number)
            out["R_mono"] = numpy.array(mono(Q.copy()), copy=True)
        r = rhs.full
        want = set(EULER_LIGHT_ATTRS)
        if rank in phase_panels:
            want |= set(EULER_PHASE_ATTRS)
        for a in sorted(want):
            out["phase/" + a] = numpy.array(getattr(r, a), copy=True)
        if rank in phase_panels:
            out["phase/df1_dx1"] = r.df1_dx1.copy()  # after corrections (phase 7)
            out["phase/df2_dx2"] = r.df2_dx2.copy()
            out["phase/df3_dx3"] = r.df3_dx3.copy()
            out["phase/w_df1_dx1"] = r.w_df1_dx1.copy()
            out["phase/w_df2_dx2"] = r.w_df2_dx2.copy()
            out["phase/w_df3_dx3"] = numpy.array(r.w_df3_dx3, copy=True)

        # complex-step perturbed evaluation (matvec_fun semantics: Q + i*eps*v)
        v = rng.uniform(-1.0, 1.0, Q.shape) * numpy.abs(Q).max(axis=(1, 2, 3, 4), keepdims=True) * 1e-3
        eps = numpy.sqrt(numpy.finfo(float).eps)
        Qc = Q + 1j * eps * v
        Rc = rhs.full(Qc)
        if rank in metric_panels:  # complex inputs/outputs only where they can be checked
            out["V"] = v
            out["Rc"] = Rc.copy()
        for a in EULER_LIGHT_ATTRS:
            out["cphase/" + a] = numpy.array(getattr(r, a), copy=True)

        if rank in metric_panels:
            for a in EULER_METRIC_ATTRS:
                out["metric/" + a] = numpy.array(getattr(metric, a), copy=True)
            if cfg.case_number in (21, 22):
                coef, uref = _damping_fields(geom, metric, cfg.case_number, Q.shape[1:])
                out["metric/damp_coef"] = coef
                out["metric/damp_uref"] = uref
        out["geom/boundary_sn_new"] = numpy.array(geom.boundary_sn_new, copy=True)
        out["geom/boundary_we_new"] = numpy.array(geom.boundary_we_new, copy=True)
        if rank == 0:
            out.update(_ops_1d(ops, geom))
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/eps"] = numpy.float64(eps)
            # Kronecker-ordering pin: dense operators applied to a seeded element
            u = numpy.random.default_rng(7).uniform(-1, 1, n**3)
            out["kron/u"] = u
            for opn in ("derivative_x", "derivative_y", "derivative_z", "extrap_x", "extrap_y", "extrap_z",
                        "highfilter_k"):
                out["kron/" + opn] = u @ getattr(ops, opn)
            f = numpy.random.default_rng(8).uniform(-1, 1, 2 * n**2)
            out["kron/f"] = f
            for opn in ("correction_WE", "correction_SN", "correction_DU"):
                out["kron/" + opn] = f @ getattr(ops, opn)
        return out

    t0 = time.time()
    MPI.reset_world(6)
    res, err = MPI.run_ranks(work, 6)
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            if k.startswith(("ops/", "meta/", "kron/")):
                flat[k] = v
            else:
                flat[f"p{p}/{k}"] = v
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)  "
          f"max|R| per var panel0 = {numpy.abs(res[0]['R']).max(axis=(1,2,3,4))}", flush=True)



# ---------------------------------------------------------------------------------------------
# Shallow water on the cubed sphere (rhs_sw.py)
# ---------------------------------------------------------------------------------------------
SW_METRIC_ATTRS = [
    "sqrtG", "inv_sqrtG", "H_contra_11", "H_contra_12", "H_contra_21", "H_contra_22",
    "christoffel_1_01", "christoffel_1_02", "christoffel_1_11", "christoffel_1_12",
    "christoffel_2_01", "christoffel_2_02", "christoffel_2_12", "christoffel_2_22",
    "sqrtG_itf_i", "sqrtG_itf_j",
    "H_contra_11_itf_i", "H_contra_21_itf_i", "H_contra_12_itf_j", "H_contra_22_itf_j",
]
SW_TOPO_ATTRS = ["hsurf", "dzdx1", "dzdx2", "hsurf_itf_i", "hsurf_itf_j"]


class _Recorder:
    """Wraps a ProcessTopology's exchange starters to keep what wait() delivered (harness-side
    instrumentation; the reference object itself is untouched)."""

    def __init__(self, pt):
        self.pt = pt
        self.vectors = None
        self.scalars = None
        self._sv, self._ss = pt.start_exchange_vectors, pt.start_exchange_scalars
        pt.start_exchange_vectors = self._vec
        pt.start_exchange_scalars = self._sca

    def _wrap(self, req, slot):
        rec = self
        orig_wait = req.wait

        def wait():
            out = orig_wait()
            setattr(rec, slot, [numpy.array(o, copy=True) for o in out])
            return out

        req.wait = wait
        return req

    def _vec(self, *a, **k):
        return self._wrap(self._sv(*a, **k), "vectors")

    def _sca(self, *a, **k):
        return self._wrap(self._ss(*a, **k), "scalars")


def sw_case(name, ini, overrides, perturb=0.0, seed=4321, n_ranks=6, rk3=None, epi=None):
    cfg_probe = _config(ini, overrides)
    print(f"[{name}] {ini} n={cfg_probe.num_solpts} H={cfg_probe.num_elements_horizontal} on {n_ranks} ranks", flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere2D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle

        cfg = _config(ini, overrides)
        import math
        per_line = int(math.isqrt(n_ranks // 6))
        cfg.num_elements_horizontal = cfg.num_elements_horizontal_total // per_line
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere2D(cfg.num_elements_horizontal, cfg.num_solpts, cfg.lambda0, cfg.phi0, cfg.alpha0,
                             pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)
        rec = _Recorder(pt)
        rng = numpy.random.default_rng(seed + rank)
        if perturb > 0.0:
            Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
            Q[1:] += perturb * 1e-6 * Q[0] * rng.uniform(-1.0, 1.0, Q[1:].shape)
        out = {"Q": Q.copy()}
        with numpy.errstate(all="ignore"):
            out["R"] = rhs.full(Q).copy()
        # halo faces as delivered: [edge][var h, hu1, hu2] each (H, n)
        out["halo"] = numpy.stack([numpy.stack([rec.scalars[e], rec.vectors[e][0], rec.vectors[e][1]])
                                   for e in range(4)])
        v = rng.uniform(-1.0, 1.0, Q.shape) * numpy.abs(Q).max(axis=(1, 2, 3), keepdims=True) * 1e-3
        eps = numpy.sqrt(numpy.finfo(float).eps)
        with numpy.errstate(all="ignore"):
            Rc = rhs.full(Q + 1j * eps * v)
        out["V"] = v
        out["Rc"] = Rc.copy()
        out["chalo"] = numpy.stack([numpy.stack([rec.scalars[e], rec.vectors[e][0], rec.vectors[e][1]])
                                    for e in range(4)])
        if rk3 is not None:   # the explicit time loop of BASELINE configs 2 / 3: Tvdrk3.step + apply_filters (simulation.py:147-155)
            from integrators import Tvdrk3

            nsteps, dt_rk = rk3
            stepper = Tvdrk3(cfg, rhs.full, device=dev)
            Qs = Q.copy()
            for i in range(nsteps):
                Qs = stepper.step(Qs, dt_rk)
                Qs = ops.apply_filters(Qs, geom, metric, dt_rk)
                if i == 0:
                    out["rk3_1"] = Qs.copy()
            out["rk3_n"] = Qs.copy()
        if epi is not None:   # config/case6.ini as shipped: Epi (order 3) with the ini's exponential solver (pmex), complex step
            import integrators.epi as epi_mod

            order, nsteps, dt_epi = epi
            cfg.verbose_solver = 0
            solver_stats = []
            real = getattr(epi_mod, cfg.exponential_solver)

            def logging_solver(*a, **k):
                phiv, stats = real(*a, **k)
                solver_stats.append([float(x) for x in stats])
                return phiv, stats

            if rank == 0:
                setattr(epi_mod, cfg.exponential_solver, logging_solver)   # (module attribute: every rank's calls pass here)
            MPI.COMM_WORLD.Barrier()
            try:
                stepper = epi_mod.Epi(cfg, order, rhs.full, device=dev)
                Qs = Q.copy()
                for i in range(nsteps):
                    Qs = stepper.step(Qs, dt_epi)
                    Qs = ops.apply_filters(Qs, geom, metric, dt_epi)
                    out[f"epi_{i + 1}"] = numpy.array(Qs, copy=True)
            finally:
                MPI.COMM_WORLD.Barrier()
                if rank == 0:
                    setattr(epi_mod, cfg.exponential_solver, real)
            if rank == 0:
                out["meta/epi_order"], out["meta/epi_steps"] = numpy.int64(order), numpy.int64(nsteps)
                out["meta/epi_dt"], out["meta/epi_tol"] = numpy.float64(dt_epi), numpy.float64(cfg.tolerance)
                out["meta/epi_solver"] = numpy.array(cfg.exponential_solver)
                # one row per call of the solver, in order, all ranks' calls interleaved: rank 0's are rows 0, 6, 12, ...
                out["meta/epi_solver_stats_all"] = numpy.array(solver_stats)
        for a in SW_METRIC_ATTRS:
            out["metric/" + a] = numpy.array(getattr(metric, a), copy=True)
        if topo is not None:
            for a in SW_TOPO_ATTRS:
                out["topo/" + a] = numpy.array(getattr(topo, a), copy=True)
        out["geom/boundary_sn"] = numpy.array(geom.boundary_sn, copy=True)
        out["geom/boundary_we"] = numpy.array(geom.boundary_we, copy=True)
        out["tile/panel_row_col"] = numpy.array([pt.my_panel, pt.my_row, pt.my_col], dtype=numpy.int64)
        if rank == 0:
            out.update(_ops_1d(ops, geom))
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/k"] = numpy.int64(per_line)
            out["meta/eps"] = numpy.float64(eps)
            if rk3 is not None:
                out["meta/rk3_steps"], out["meta/rk3_dt"] = numpy.int64(rk3[0]), numpy.float64(rk3[1])
            out["meta/grid_rotation"] = numpy.array([cfg.lambda0, cfg.phi0, cfg.alpha0], dtype=float)
            n = cfg.num_solpts
            u = numpy.random.default_rng(7).uniform(-1, 1, n * n)
            out["kron/u"] = u
            for opn in ("derivative_x", "derivative_y", "extrap_x", "extrap_y"):
                out["kron/" + opn] = u @ getattr(ops, opn)
            f = numpy.random.default_rng(8).uniform(-1, 1, 2 * n)
            out["kron/f"] = f
            for opn in ("correction_WE", "correction_SN"):
                out["kron/" + opn] = f @ getattr(ops, opn)
        return out

    if n_ranks == 6:
        _run6(name, work)
        return
    t0 = time.time()
    MPI.reset_world(n_ranks)
    res, err = MPI.run_ranks(work, n_ranks)
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            flat[k if k.startswith(("ops/", "meta/", "kron/")) else f"p{p}/{k}"] = v
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)", flush=True)
    MPI.reset_world(6)


# ---------------------------------------------------------------------------------------------
# 2-D Cartesian Euler (rhs_dfr.py:8-45 + pde_euler_cartesian.py + the reference's native pde_cpp)
# ---------------------------------------------------------------------------------------------
def cart2d_case(name, ini, overrides, seed=99, rk3=None, epi=None, wind=0.5):
    MPI.reset_world(1)

    def work(rank):
        from device import CpuDevice
        from geometry import Cartesian2D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle

        cfg = _config(ini, overrides)
        dev = CpuDevice(MPI.COMM_WORLD)
        geom = Cartesian2D((cfg.x0, cfg.x1), (cfg.z0, cfg.z1), cfg.num_elements_horizontal,
                           cfg.num_elements_vertical, cfg.num_solpts, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rng = numpy.random.default_rng(seed)
        Q = Q * (1.0 + 1e-3 * rng.uniform(-1, 1, Q.shape))
        Q[1] += wind * Q[0] * rng.uniform(-1, 1, Q[0].shape)   # some wind so that both AUSM branches are hit
        Q[2] += wind * Q[0] * rng.uniform(-1, 1, Q[0].shape)
        rhs = RhsBundle(geom, ops, metric, topo, None, cfg, Q.shape, False)
        out = {"Q": Q.copy(), "R": rhs.full(Q).copy()}
        r = rhs.full
        for a in ("q_itf_x1", "q_itf_x3", "f_x1", "f_x3", "f_itf_x1", "f_itf_x3"):
            out["phase/" + a] = numpy.array(getattr(r, a), copy=True)
        v = rng.uniform(-1, 1, Q.shape) * numpy.abs(Q).max(axis=(1, 2, 3), keepdims=True) * 1e-3
        eps = numpy.sqrt(numpy.finfo(float).eps)
        out["V"] = v
        out["Rc"] = rhs.full(Q + 1j * eps * v).copy()
        if rk3 is not None:   # BASELINE config 1's explicit time loop: Tvdrk3.step + apply_filters (simulation.py:147-155)
            from integrators import Tvdrk3

            nsteps, dt_rk = rk3
            stepper = Tvdrk3(cfg, rhs.full, device=dev)
            Qs = Q.copy()
            for i in range(nsteps):
                Qs = stepper.step(Qs, dt_rk)
                Qs = ops.apply_filters(Qs, geom, metric, dt_rk)
                if i == 0:
                    out["rk3_1"] = Qs.copy()
            out["rk3_n"] = Qs.copy()
            out["meta/rk3_steps"], out["meta/rk3_dt"] = numpy.int64(nsteps), numpy.float64(dt_rk)
        if epi is not None:   # the shipped bubble configurations' own integrator: Epi + the ini's exponential solver, complex step
            import integrators.epi as epi_mod

            order, nsteps, dt_epi, tol = epi
            cfg.verbose_solver = 0
            cfg.tolerance = tol
            solver_stats = []
            real = getattr(epi_mod, cfg.exponential_solver)

            def logging_solver(*a, **k):
                phiv, stats = real(*a, **k)
                solver_stats.append([float(x) for x in stats])
                return phiv, stats

            setattr(epi_mod, cfg.exponential_solver, logging_solver)
            try:
                stepper = epi_mod.Epi(cfg, order, rhs.full, device=dev)
                Qs = Q.copy()
                for i in range(nsteps):
                    Qs = stepper.step(Qs, dt_epi)
                    Qs = ops.apply_filters(Qs, geom, metric, dt_epi)
                    out[f"epi_{i + 1}"] = numpy.array(Qs, copy=True)
            finally:
                setattr(epi_mod, cfg.exponential_solver, real)
            out["meta/epi_order"], out["meta/epi_steps"] = numpy.int64(order), numpy.int64(nsteps)
            out["meta/epi_dt"], out["meta/epi_tol"] = numpy.float64(dt_epi), numpy.float64(tol)
            out["meta/epi_solver"] = numpy.array(cfg.exponential_solver)
            out["meta/epi_solver_stats"] = numpy.array(solver_stats)
        out.update(_ops_1d(ops, geom))
        out["meta/n"] = numpy.int64(cfg.num_solpts)
        out["meta/nx"] = numpy.int64(cfg.num_elements_horizontal)
        out["meta/nz"] = numpy.int64(cfg.num_elements_vertical)
        out["meta/dx1"] = numpy.float64(geom.Δx1)
        out["meta/dx3"] = numpy.float64(geom.Δx3)
        out["meta/eps"] = numpy.float64(eps)
        return out

    t0 = time.time()
    print(f"[{name}] {ini}", flush=True)
    res, err = MPI.run_ranks(work, 1)
    if err[0]:
        print(err[0])
        raise SystemExit(1)
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **res[0])
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)", flush=True)
    MPI.reset_world(6)


def _run6(name, work):
    t0 = time.time()
    MPI.reset_world(6)
    res, err = MPI.run_ranks(work, 6)
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            if k.startswith(("ops/", "meta/", "kron/")):
                flat[k] = v
            else:
                flat[f"p{p}/{k}"] = v
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    probe = res[0]["R"] if "R" in res[0] else res[0]["Q"]
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)  "
          f"max|{'R' if 'R' in res[0] else 'Q'}| panel0 = {numpy.abs(probe).max(axis=tuple(range(1, probe.ndim)))}", flush=True)



# ---------------------------------------------------------------------------------------------
# Callers of the path: SSP-RK3 step, matvec_fun / matvec_rat JVPs, Ros2 + FGMRES step
# ---------------------------------------------------------------------------------------------
def callers_case(name, ini, overrides, dt_rk=2.0, dt_jvp=30.0, perturb=0.01, seed=777):
    print(f"[{name}] {ini}", flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle
        from integrators import Tvdrk3, Ros2
        from solvers.matvec import matvec_fun, matvec_rat

        cfg = _config(ini, overrides)
        cfg.verbose_solver = 0
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)
        rng = numpy.random.default_rng(seed + rank)
        Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
        v = rng.uniform(-1.0, 1.0, Q.shape) * numpy.abs(Q).max(axis=(1, 2, 3, 4), keepdims=True) * 1e-3
        out = {"Q": Q.copy(), "V": v.copy()}
        R = rhs.full(Q)
        out["R"] = R.copy()
        out["jvp_complex"] = matvec_fun(v.flatten(), dt_jvp, Q, R, rhs.full, "complex").reshape(Q.shape)
        out["jvp_fd"] = matvec_fun(v.flatten(), dt_jvp, Q, R, rhs.full, "fd").reshape(Q.shape)
        out["rat"] = matvec_rat(v.flatten(), dt_jvp, Q, R, rhs.full).reshape(Q.shape)
        rk = Tvdrk3(cfg, rhs.full, device=dev)
        out["rk3"] = rk.step(Q.copy(), dt_rk).copy()
        cfg.tolerance = 1e-9
        cfg.gmres_restart = 30
        cfg.linear_solver = "fgmres"
        ros = Ros2(cfg, rhs.full, device=dev)
        out["ros2"] = ros.step(Q.copy(), dt_jvp).copy()
        # EPI2 + KIOPS with the complex-step JVP (config/dcmip31.ini's own integrator)
        from integrators import Epi
        from solvers.kiops import kiops
        cfg.tolerance = 1e-7
        cfg.exponential_solver = "kiops"
        cfg.jacobian_method = "complex"
        epi = Epi(cfg, 2, rhs.full, device=dev)
        out["epi2"] = epi.step(Q.copy(), dt_jvp).copy()
        vec = numpy.zeros((2, R.size))
        vec[1, :] = R.flatten()
        phiv, stats = kiops([1], lambda x: matvec_fun(x, dt_jvp, Q, R, rhs.full, "complex"), vec, tol=1e-7,
                            m_init=1, mmin=16, mmax=64, task1=False, device=dev)
        out["kiops_phiv"] = numpy.asarray(phiv).reshape(Q.shape)
        out["kiops_stats"] = numpy.array([float(x) for x in stats])
        for a in EULER_METRIC_ATTRS:
            out["metric/" + a] = numpy.array(getattr(metric, a), copy=True)
        out["geom/boundary_sn_new"] = numpy.array(geom.boundary_sn_new, copy=True)
        out["geom/boundary_we_new"] = numpy.array(geom.boundary_we_new, copy=True)
        if rank == 0:
            out.update(_ops_1d(ops, geom))
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/dt_rk"] = numpy.float64(dt_rk)
            out["meta/dt_jvp"] = numpy.float64(dt_jvp)
            out["meta/eps"] = numpy.float64(numpy.sqrt(numpy.finfo(float).eps))
        return out

    _run6(name, work)



def pmex_case(name, base, ini, overrides, dt_jvp=30.0, perturb=0.01, seed=777):
    """solvers/pmex.py (the schema's default `exponential_solver`) on the state of the callers fixture `base` (same
    configuration, seed and perturbation, so that file's Q, R and metric are this one's inputs: only pmex's outputs are
    stored here, with max|Q| per panel as the consistency check): phi_1(dt J) R exactly as Epi calls it
    (integrators/epi.py:314-315), a three-row call with two output times (the augmented block with p = 2 and the
    intermediate outputs of solvers/pmex.py:318-333), and the EPI2 step with exponential_solver = pmex."""
    print(f"[{name}] {ini} (inputs of {base})", flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle
        from integrators import Epi
        from solvers.matvec import matvec_fun
        from solvers.pmex import pmex

        cfg = _config(ini, overrides)
        cfg.verbose_solver = 0
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)
        rng = numpy.random.default_rng(seed + rank)
        Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
        v = rng.uniform(-1.0, 1.0, Q.shape) * numpy.abs(Q).max(axis=(1, 2, 3, 4), keepdims=True) * 1e-3
        R = rhs.full(Q)
        out = {"Q_absmax": numpy.abs(Q).max(axis=(1, 2, 3, 4)), "R_absmax": numpy.abs(R).max(axis=(1, 2, 3, 4))}
        A = lambda x: matvec_fun(x, dt_jvp, Q, R, rhs.full, "complex")
        vec = numpy.zeros((2, R.size))
        vec[1, :] = R.flatten()
        phiv, stats = pmex([1.0], A, vec, tol=1e-7, mmax=64, task1=False, device=dev)
        out["pmex_phiv"] = numpy.asarray(phiv).reshape(Q.shape)
        out["pmex_stats"] = numpy.array([float(x) for x in stats])
        vec3 = numpy.zeros((3, R.size))
        vec3[0, :] = v.flatten()
        vec3[1, :] = R.flatten()
        vec3[2, :] = 0.01 * A(v.flatten())
        w3, stats3 = pmex([0.25, 0.6, 1.0], A, vec3, tol=1e-9, m_init=6, mmin=6, mmax=40, task1=True, device=dev)
        out["pmex3_w"] = numpy.asarray(w3).reshape((3,) + Q.shape)
        out["pmex3_stats"] = numpy.array([float(x) for x in stats3])
        cfg.tolerance = 1e-7
        cfg.exponential_solver = "pmex"
        cfg.jacobian_method = "complex"
        epi = Epi(cfg, 2, rhs.full, device=dev)
        out["epi2_pmex"] = epi.step(Q.copy(), dt_jvp).copy()
        if rank == 0:
            out["meta/dt_jvp"] = numpy.float64(dt_jvp)
            out["meta/base"] = numpy.array(base)
        return out

    t0 = time.time()
    MPI.reset_world(6)
    res, err = MPI.run_ranks(work, 6)
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            flat[k if k.startswith("meta/") else f"p{p}/{k}"] = v
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)  pmex stats "
          f"{res[0]['pmex_stats'].tolist()}  three-row {res[0]['pmex3_stats'].tolist()}", flush=True)



def solvers_dense_case(name):
    """solvers/kiops.py and solvers/pmex.py on seeded dense operators (one rank): what a CPU-only test can hold the
    host logic of both to - the projector, the norm estimate, both controllers, intermediate output times, task1,
    a negative time, a happy breakdown and the reference's own unit-test problem (64 rows of a 64 x 64 matrix as `u`,
    identity operator: tests/unit/solvers/test_pmex.py, test_kiops_pmex_tolerance_cpu.py).  Inputs and outputs only."""
    from device import CpuDevice
    from solvers.kiops import kiops
    from solvers.pmex import pmex

    MPI.reset_world(1)
    rng = numpy.random.default_rng(20260401)

    def stiff(n, scale, spread):
        return scale * (-numpy.diag(rng.uniform(0.05, spread, n)) + 0.4 * rng.standard_normal((n, n)) / numpy.sqrt(n))

    n = 160
    problems = {}
    problems["phi1"] = dict(A=stiff(n, 1.0, 6.0), u=numpy.vstack((numpy.zeros(n), rng.standard_normal(n))), tau=[1.0],
                            kiops=dict(tol=1e-7, m_init=1, mmin=16, mmax=64, task1=False), pmex=dict(tol=1e-7, mmax=64, task1=False))
    problems["phi3_outputs"] = dict(A=stiff(n, 1.0, 4.0), u=rng.standard_normal((4, n)), tau=[0.3, 0.7, 1.0],
                                    kiops=dict(tol=1e-10, m_init=6, mmin=6, mmax=40, task1=True),
                                    pmex=dict(tol=1e-10, m_init=6, mmin=6, mmax=40, task1=True))
    problems["long_interval"] = dict(A=stiff(n, 12.0, 8.0), u=rng.standard_normal((3, n)), tau=[2.5],
                                     kiops=dict(tol=1e-8, m_init=10, mmin=10, mmax=24, task1=False),
                                     pmex=dict(tol=1e-8, m_init=10, mmin=10, mmax=24, task1=False))
    problems["backwards"] = dict(A=stiff(n, 1.0, 3.0), u=rng.standard_normal((2, n)), tau=[-1.0],
                                 kiops=dict(tol=1e-9, m_init=8, mmin=8, mmax=48, task1=False),
                                 pmex=dict(tol=1e-9, m_init=8, mmin=8, mmax=48, task1=False))
    basis = numpy.linalg.qr(rng.standard_normal((n, 5)))[0]                   # a 5-dimensional invariant subspace
    inv = basis @ (-numpy.diag([0.5, 1.0, 1.5, 2.0, 2.5])) @ basis.T
    # (a one-row `u` cannot be given to the reference: its p = 0 branch stacks a row of the wrong length, kiops.py:83)
    problems["invariant_subspace"] = dict(A=inv, u=numpy.vstack((basis @ rng.standard_normal(5), numpy.zeros(n))), tau=[1.0],
                                          kiops=dict(tol=1e-9, m_init=12, mmin=12, mmax=30, task1=False),
                                          pmex=dict(tol=1e-9, m_init=12, mmin=12, mmax=30, task1=False))
    problems["unit_test_identity"] = dict(A=numpy.eye(64), u=rng.uniform(-1000.0, 1000.0, (64, 64)), tau=[1.0],
                                          kiops=dict(tol=1e-7), pmex=dict(tol=1e-7))

    def work(rank):
        dev = CpuDevice(MPI.COMM_WORLD)
        out = {}
        for key, pr in problems.items():
            A = pr["A"]
            for solver, fn in (("kiops", kiops), ("pmex", pmex)):
                w, stats = fn(list(pr["tau"]), lambda v: A @ v, pr["u"].copy(), device=dev, **pr[solver])
                out[f"{key}/{solver}_w"] = numpy.array(w, copy=True)
                out[f"{key}/{solver}_stats"] = numpy.array([float(x) for x in stats])
        return out

    res, err = MPI.run_ranks(work, 1)
    if err[0]:
        print(err[0])
        raise SystemExit(1)
    flat = dict(res[0])
    import json
    for key, pr in problems.items():
        flat[f"{key}/A"], flat[f"{key}/u"], flat[f"{key}/tau"] = pr["A"], pr["u"], numpy.array(pr["tau"])
        flat[f"{key}/kiops_args"] = numpy.array(json.dumps(pr["kiops"]))
        flat[f"{key}/pmex_args"] = numpy.array(json.dumps(pr["pmex"]))
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    for key in problems:
        print(f"   {key}: kiops {flat[key + '/kiops_stats'].tolist()}  pmex {flat[key + '/pmex_stats'].tolist()}")
    print(f"[{name}] -> {path}  {os.path.getsize(path)/1e6:.2f} MB", flush=True)
    MPI.reset_world(6)


def epi_case(name, ini, overrides, orders=(3, 4, 5, 6), extra_steps=2, dt=30.0, perturb=0.01, seed=999, stiff=False,
             solver="kiops"):
    """Multistep EPI integrators (integrators/epi.py:28-141): for each order, n_prev start-up steps (EPI2) plus
    `extra_steps` regular steps from the same perturbed state; pins the coefficient tables, the phi-vector
    assembly from previous states and KIOPS with several phi functions."""
    print(f"[{name}] {ini}", flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle
        from integrators import Epi

        cfg = _config(ini, overrides)
        cfg.verbose_solver = 0
        cfg.tolerance = 1e-7
        cfg.exponential_solver = solver
        cfg.jacobian_method = "complex"
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)
        rng = numpy.random.default_rng(seed + rank)
        Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
        out = {"Q": Q.copy()}
        for order in orders:
            if stiff:   # integrators/epi_stiff.py as simulation.py:336-340 builds it (config/dcmip20.ini: epi_stiff3)
                from integrators import EpiStiff

                epi = EpiStiff(cfg, order, rhs.full, init_substeps=10, device=dev)
            else:
                epi = Epi(cfg, order, rhs.full, device=dev)
            Qn = Q.copy()
            for _ in range(epi.n_prev + extra_steps):
                Qn = epi.step(Qn, dt)
            out[f"epi{order}"] = numpy.array(Qn, copy=True)
            if rank == 0:
                out[f"meta/steps_epi{order}"] = numpy.int64(epi.n_prev + extra_steps)
        for a in EULER_METRIC_ATTRS:
            out["metric/" + a] = numpy.array(getattr(metric, a), copy=True)
        out["geom/boundary_sn_new"] = numpy.array(geom.boundary_sn_new, copy=True)
        out["geom/boundary_we_new"] = numpy.array(geom.boundary_we_new, copy=True)
        if rank == 0:
            out.update(_ops_1d(ops, geom))
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/dt"] = numpy.float64(dt)
            out["meta/eps"] = numpy.float64(numpy.sqrt(numpy.finfo(float).eps))
        return out

    _run6(name, work)


def euler_tiles_case(name, ini, overrides, n_ranks, metric_tiles, perturb=0.01, seed=4242):
    """3-D Euler on 6 k^2 tiles (k^2 ranks per panel): pins the tile topology, the identity rotation on
    interior tile edges and the panel-edge tables for tiles (process_topology.py:69-256)."""
    print(f"[{name}] {ini} on {n_ranks} ranks", flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle

        cfg = _config(ini, overrides)
        import math
        per_line = int(math.isqrt(n_ranks // 6))
        cfg.num_elements_horizontal = cfg.num_elements_horizontal_total // per_line
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)
        rng = numpy.random.default_rng(seed + rank)
        Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
        Q[3] = Q[3] + Q[0] * perturb * 1e-5 * rng.uniform(-1.0, 1.0, Q[0].shape)
        out = {"Q": Q.copy(), "R": rhs.full(Q).copy()}
        r = rhs.full
        for a in EULER_LIGHT_ATTRS:
            out["phase/" + a] = numpy.array(getattr(r, a), copy=True)
        out["topo/neighbors"] = numpy.array(pt.sources, dtype=numpy.int64)      # S, N, W, E ranks
        out["topo/flip"] = numpy.array(pt.flip, dtype=numpy.int64)
        out["topo/panel_row_col"] = numpy.array([pt.my_panel, pt.my_row, pt.my_col], dtype=numpy.int64)
        if rank in metric_tiles:
            for a in EULER_METRIC_ATTRS:
                out["metric/" + a] = numpy.array(getattr(metric, a), copy=True)
        out["geom/boundary_sn_new"] = numpy.array(geom.boundary_sn_new, copy=True)
        out["geom/boundary_we_new"] = numpy.array(geom.boundary_we_new, copy=True)
        if rank == 0:
            out.update(_ops_1d(ops, geom))
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/k"] = numpy.int64(per_line)
            out["meta/eps"] = numpy.float64(numpy.sqrt(numpy.finfo(float).eps))
        return out

    t0 = time.time()
    MPI.reset_world(n_ranks)
    res, err = MPI.run_ranks(work, n_ranks)
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            if k.startswith(("ops/", "meta/")):
                flat[k] = v
            else:
                flat[f"p{p}/{k}"] = v
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)", flush=True)
    MPI.reset_world(6)


def metric_case(name, ini, overrides, n_ranks, direct=False):
    """Only the static metric of a 3-D run: every tile, any grid rotation, with topography
    (cases 21/22 through init_state_vars) or, with direct=True, Metric3DTopo.build_metric on the smooth sphere
    for a case number whose planet rotates (exercises the rotation Christoffel symbols)."""
    print(f"[{name}] {ini} on {n_ranks} ranks", flush=True)

    def work(rank):
        import math
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators, Metric3DTopo
        from init.init_state_vars import init_state_vars

        cfg = _config(ini, overrides)
        per_line = int(math.isqrt(n_ranks // 6))
        cfg.num_elements_horizontal = cfg.num_elements_horizontal_total // per_line
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        if direct:
            metric = Metric3DTopo(geom, ops)
            metric.build_metric()
        else:
            Q, topo, metric = init_state_vars(geom, ops, cfg)
        out = {}
        for a in EULER_METRIC_ATTRS:
            out["metric/" + a] = numpy.array(getattr(metric, a), copy=True)
        if cfg.case_number in (21, 22) and not direct:
            coef, uref = _damping_fields(geom, metric, cfg.case_number, Q.shape[1:])
            out["metric/damp_coef"] = coef
            out["metric/damp_uref"] = uref
        out["geom/boundary_sn_new"] = numpy.array(geom.boundary_sn_new, copy=True)
        out["geom/boundary_we_new"] = numpy.array(geom.boundary_we_new, copy=True)
        out["topo/panel_row_col"] = numpy.array([pt.my_panel, pt.my_row, pt.my_col], dtype=numpy.int64)
        if rank == 0:
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/k"] = numpy.int64(per_line)
            out["meta/ztop"] = numpy.float64(cfg.ztop)
            out["meta/rotation"] = numpy.array([cfg.lambda0, cfg.phi0, cfg.alpha0], dtype=numpy.float64)
            out["meta/deep"] = numpy.int64(cfg.depth_approx.lower() == "deep")
        return out

    t0 = time.time()
    MPI.reset_world(n_ranks)
    res, err = MPI.run_ranks(work, n_ranks)
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            flat[k if k.startswith("meta/") else f"p{p}/{k}"] = v
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)", flush=True)
    MPI.reset_world(6)


def state_file_case(name):
    """Bytes written by the reference's save_state (output/state.py:9-16) for a seeded global state."""
    import types
    from common import ConfigurationSchema
    from output.state import save_state
    from device import CpuDevice

    MPI.reset_world(1)
    schema = ConfigurationSchema(SCHEMA_TEXT)
    rng = numpy.random.default_rng(2024)
    state = rng.uniform(-1, 1, (6, 3, 2, 2, 9))
    cfg_text = "[General]\nequations = shallow_water\n\n[Grid]\ngrid_type = cubed_sphere\n"
    param = types.SimpleNamespace(state_version=schema.version, config_content=cfg_text)
    path = "/tmp/wx_golden_state.npy"

    def work(rank):
        save_state(state, param, path, device=CpuDevice(MPI.COMM_WORLD))
        return True

    res, err = MPI.run_ranks(work, 1)
    if err[0]:
        print(err[0])
        raise SystemExit(1)
    raw = numpy.frombuffer(open(path, "rb").read(), dtype=numpy.uint8)
    out = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(out, file_bytes=raw, state=state, state_version=numpy.array(str(schema.version)),
                           config_text=numpy.array(cfg_text))
    print(f"[{name}] -> {out} ({raw.size} bytes in the reference-written file, version {schema.version!r})", flush=True)
    MPI.reset_world(6)


# ---------------------------------------------------------------------------------------------
# Per-step filter of the explicit loop: operators.apply_filters = 3-D exponential modal filter
# ---------------------------------------------------------------------------------------------
def filters_case(name, ini, overrides, perturb=0.02, seed=31337):
    print(f"[{name}] {ini}", flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars

        cfg = _config(ini, overrides)
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rng = numpy.random.default_rng(seed + rank)
        Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
        assert ops.expfilter_apply
        out = {"Q": Q.copy(), "R": ops.apply_filters(Q.copy(), geom, metric, cfg.dt).copy(),
               "metric/sqrtG_new": numpy.array(metric.sqrtG_new, copy=True),
               "metric/inv_sqrtG_new": numpy.array(metric.inv_sqrtG_new, copy=True)}
        if rank == 0:
            out["ops/expfilter"] = numpy.array(ops.expfilter, copy=True)
            out["ops/solution_points"] = numpy.array(geom.solutionPoints, copy=True)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/expfilter_strength"] = numpy.float64(cfg.expfilter_strength)
            out["meta/expfilter_order"] = numpy.int64(cfg.expfilter_order)
            out["meta/expfilter_cutoff"] = numpy.float64(cfg.expfilter_cutoff)
        return out

    _run6(name, work)


# ---------------------------------------------------------------------------------------------
# The body of Simulation.step repeated (simulation.py:147-155): integrator step, apply_filters, NaN check
# ---------------------------------------------------------------------------------------------
def steploop_case(name, ini, overrides, nsteps=5, dt=None, perturb=0.01, seed=2718):
    print(f"[{name}] {ini}", flush=True)

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle
        from integrators import Tvdrk3

        cfg = _config(ini, overrides)
        step = float(dt if dt is not None else cfg.dt)
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)
        rng = numpy.random.default_rng(seed + rank)
        Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
        out = {"Q": Q.copy()}
        stepper = Tvdrk3(cfg, rhs.full, device=dev)
        assert ops.expfilter_apply
        for i in range(nsteps):
            Q = stepper.step(Q, step)
            Q = ops.apply_filters(Q, geom, metric, step)
            assert not numpy.any(numpy.isnan(Q))
            if i == 0:
                out["Q1"] = Q.copy()
        out["Qn"] = Q.copy()
        if rank == 0:
            out["ops/expfilter"] = numpy.array(ops.expfilter, copy=True)
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/ztop"] = numpy.float64(cfg.ztop)
            out["meta/dt"] = numpy.float64(step)
            out["meta/nsteps"] = numpy.int64(nsteps)
        return out

    t0 = time.time()
    MPI.reset_world(6)
    res, err = MPI.run_ranks(work, 6)
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            flat[k if k.startswith(("meta/", "ops/")) else f"p{p}/{k}"] = v
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)", flush=True)


# ---------------------------------------------------------------------------------------------
# BASELINE config 5's own combination: config/dcmip21.ini (Schaer mountain: topography + Rayleigh sponge),
# EPI2 + KIOPS with the complex-step JVP, followed by the per-step exponential filter - the body of
# Simulation.step (simulation.py:147-155) with integrators/epi.py:81-360 as the stepper.
# ---------------------------------------------------------------------------------------------
def config5_case(name, ini, overrides, nsteps=2, dt=None, perturb=0.01, seed=5150):
    print(f"[{name}] {ini}", flush=True)
    import integrators.epi as epi_mod
    import threading

    stats_log = {}
    lock = threading.Lock()
    real_kiops = epi_mod.kiops

    def logging_kiops(*a, **k):
        phiv, stats = real_kiops(*a, **k)
        with lock:
            stats_log.setdefault(MPI.COMM_WORLD.rank, []).append([float(x) for x in stats])
        return phiv, stats

    def work(rank):
        from device import CpuDevice
        from process_topology import ProcessTopology
        from geometry import CubedSphere3D, DFROperators
        from init.init_state_vars import init_state_vars
        from rhs.rhs_selector import RhsBundle
        from integrators import Epi

        cfg = _config(ini, overrides)
        cfg.verbose_solver = 0
        cfg.exponential_solver = "kiops"   # BASELINE config 5 (the schema's default is pmex)
        assert cfg.time_integrator == "epi2" and cfg.jacobian_method == "complex"
        step = float(dt if dt is not None else cfg.dt)
        comm = MPI.COMM_WORLD
        dev = CpuDevice(comm)
        pt = ProcessTopology(dev, comm=comm)
        geom = CubedSphere3D(cfg.num_elements_horizontal, cfg.num_elements_vertical, cfg.num_solpts,
                             cfg.lambda0, cfg.phi0, cfg.alpha0, cfg.ztop, pt, cfg, dev)
        ops = DFROperators(geom, cfg, dev)
        Q, topo, metric = init_state_vars(geom, ops, cfg)
        rhs = RhsBundle(geom, ops, metric, topo, pt, cfg, Q.shape, False)
        rng = numpy.random.default_rng(seed + rank)
        Q = Q * (1.0 + perturb * rng.uniform(-1.0, 1.0, Q.shape))
        out = {"Q": Q.copy()}
        stepper = Epi(cfg, 2, rhs.full, device=dev)
        assert ops.expfilter_apply
        for i in range(nsteps):
            Q = stepper.step(Q, step)
            out[f"Q{i + 1}_unfiltered"] = Q.copy()
            Q = ops.apply_filters(Q, geom, metric, step)
            assert not numpy.any(numpy.isnan(Q))
            out[f"Q{i + 1}"] = Q.copy()
        if rank == 0:
            out["ops/expfilter"] = numpy.array(ops.expfilter, copy=True)
            out["meta/case_number"] = numpy.int64(cfg.case_number)
            out["meta/n"] = numpy.int64(cfg.num_solpts)
            out["meta/H"] = numpy.int64(cfg.num_elements_horizontal)
            out["meta/V"] = numpy.int64(cfg.num_elements_vertical)
            out["meta/ztop"] = numpy.float64(cfg.ztop)
            out["meta/dt"] = numpy.float64(step)
            out["meta/nsteps"] = numpy.int64(nsteps)
            out["meta/tolerance"] = numpy.float64(cfg.tolerance)
            out["meta/expfilter_strength"] = numpy.float64(cfg.expfilter_strength)
            out["meta/expfilter_order"] = numpy.int64(cfg.expfilter_order)
            out["meta/expfilter_cutoff"] = numpy.float64(cfg.expfilter_cutoff)
        return out

    t0 = time.time()
    epi_mod.kiops = logging_kiops
    try:
        MPI.reset_world(6)
        res, err = MPI.run_ranks(work, 6)
    finally:
        epi_mod.kiops = real_kiops
    for e in err:
        if e:
            print(e)
            raise SystemExit(1)
    flat = {}
    for p, d in enumerate(res):
        for k, v in d.items():
            flat[k if k.startswith(("meta/", "ops/")) else f"p{p}/{k}"] = v
    st = numpy.array(stats_log[0])
    for r in range(1, 6):
        assert numpy.array_equal(st, numpy.array(stats_log[r])), "KIOPS statistics differ between ranks"
    flat["meta/kiops_stats"] = st   # one row per step: the reference's `stats` tuple
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **flat)
    print(f"   -> {path}  {os.path.getsize(path)/1e6:.2f} MB  ({time.time()-t0:.1f}s)  kiops stats {st.tolist()}", flush=True)


CASES = {
    # balanced gravity-wave state, small n: all panels carry metrics + phases (exchange coverage)
    "euler3d_c31_n3_h4_v2": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=3, num_elements_horizontal=4, num_elements_vertical=2),
        metric_panels=range(6), phase_panels=(0, 4)),
    # same with an O(1) RHS (1 % seeded perturbation): tolerance is meaningful without cancellation
    "euler3d_c31p_n3_h4_v2": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=3, num_elements_horizontal=4, num_elements_vertical=2),
        metric_panels=range(6), phase_panels=(), perturb=0.01),
    # the benchmark order p=7
    "euler3d_c31p_n8_h2_v2": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2),
        metric_panels=(0, 4), phase_panels=(), perturb=0.01),
    "euler3d_c31_n8_h2_v2": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2),
        metric_panels=(0, 4), phase_panels=()),
    # one vertical element (E7 is reported at V in {1, 8}): both vertical faces of every element are walls
    "euler3d_c31p_n5_h2_v1": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=5, num_elements_horizontal=2, num_elements_vertical=1),
        metric_panels=(1, 5), phase_panels=(), perturb=0.01),
    # a11: the monolithic rhs_euler.py evaluated beside the DFR path (perturbed gravity wave; Schaer mountain + sponge)
    "euler3d_mono_c31p_n4_h2_v2": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=4, num_elements_horizontal=2, num_elements_vertical=2),
        metric_panels=(0, 4), phase_panels=(), perturb=0.01, monolith=True),
    "euler3d_mono_c21_n3_h2_v3": lambda nm: euler_case(
        nm, "dcmip21.ini", dict(num_solpts=3, num_elements_horizontal=2, num_elements_vertical=3),
        metric_panels=(1, 5), phase_panels=(), monolith=True),
    # topography + Rayleigh damping (Schaer mountain), even n
    "euler3d_c21_n4_h3_v4": lambda nm: euler_case(
        nm, "dcmip21.ini", dict(num_solpts=4, num_elements_horizontal=3, num_elements_vertical=4),
        metric_panels=(0, 3, 5), phase_panels=()),
    # the same with an O(1) perturbation (round 4): topography + sponge are neither balanced nor symmetric, so the TANGENT
    # Im R(Q + i eps v) / eps - sponge tangent rho / tau (u - u_ref) included - is pinned tightly (1e-10), as on c31p;
    # n = 4 (vector pipe) and the benchmark order n = 8 (matrix cores), panels that carry the Schaer mountain (0, 1)
    "euler3d_c21p_n4_h3_v4": lambda nm: euler_case(
        nm, "dcmip21.ini", dict(num_solpts=4, num_elements_horizontal=3, num_elements_vertical=4),
        metric_panels=(0, 3, 5), phase_panels=(), perturb=0.01, seed=2121),
    "euler3d_c21p_n8_h2_v2": lambda nm: euler_case(
        nm, "dcmip21.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2),
        metric_panels=(0, 1, 4), phase_panels=(), perturb=0.01, seed=2128),
    # shallow water: Rossby-Haurwitz wave (case 6), p=4 as BASELINE config 2; mountain (case 5,
    # topography); steady zonal flow (case 2) at the benchmark order p=7
    "sw_c6_n5_h4": lambda nm: sw_case(nm, "case6.ini", dict(num_solpts=5, num_elements_horizontal=4)),
    "sw_c5_n4_h3": lambda nm: sw_case(nm, "case5.ini", dict(num_solpts=4, num_elements_horizontal=3)),
    "sw_c2p_n8_h3": lambda nm: sw_case(nm, "case2.ini", dict(num_solpts=8, num_elements_horizontal=3), perturb=0.01),
    # 24 ranks = 2x2 tiles per panel, mountain case (topography), grid as shipped
    "sw_tiles24_c5_n4_h2": lambda nm: sw_case(nm, "case5.ini", dict(num_solpts=4, num_elements_horizontal=4), perturb=0.01,
                                              n_ranks=24),
    # callers: one SSP-RK3 step, JVPs (complex step / finite difference), Rosenbrock operator, Ros2 step
    "callers_euler3d_n3_h3_v2": lambda nm: callers_case(
        nm, "dcmip31.ini", dict(num_solpts=3, num_elements_horizontal=3, num_elements_vertical=2)),
    # multistep exponential integrators (orders 3 to 6) over their start-up and two regular steps
    "epi_multistep_n3_h2_v2": lambda nm: epi_case(
        nm, "dcmip31.ini", dict(num_solpts=3, num_elements_horizontal=2, num_elements_vertical=2)),
    # stiffness-resilient EPI (config/dcmip20.ini: epi_stiff3), with the start-up simulation.py gives it (10 EPI2 sub-steps)
    "epi_stiff_n3_h2_v2": lambda nm: epi_case(
        nm, "dcmip31.ini", dict(num_solpts=3, num_elements_horizontal=2, num_elements_vertical=2), orders=(3, 4), stiff=True,
        solver="pmex"),
    "state_file_v": state_file_case,
    # five SSP-RK3 steps + exponential filter of the Schaer-mountain case (topography, sponge): the time loop
    "steploop_c21_n4_h2_v3": lambda nm: steploop_case(
        nm, "dcmip21_rk3.ini", dict(num_solpts=4, num_elements_horizontal=2, num_elements_vertical=3), nsteps=5, dt=0.05),
    # static metric only: Schaer mountain on a rotated grid over 24 tiles (every kind of tile edge carries
    # a slope); deep atmosphere on a rotating Earth-size planet (rotation Christoffel symbols)
    "metric3d_c21_rot_tiles24_n3_h2_v3": lambda nm: metric_case(
        nm, "dcmip21.ini", dict(num_solpts=3, num_elements_horizontal=4, num_elements_vertical=3,
                                lambda0=-0.3, phi0=0.6, alpha0=-0.4), 24),
    "metric3d_c77_deep_rot_n4_h2_v2": lambda nm: metric_case(
        nm, "dcmip31.ini", dict(num_solpts=4, num_elements_horizontal=2, num_elements_vertical=2, case_number=77,
                                depth_approx="deep", lambda0=-0.2, phi0=0.3, alpha0=-0.1), 6, direct=True),
    # exponential modal filter applied after every step (dcmip21.ini: strength 0.1, order 4, cutoff 0.5;
    # dcmip21_rk3.ini: strength 1e-3), even and odd n
    "filters_c21_n4_h3_v4": lambda nm: filters_case(
        nm, "dcmip21.ini", dict(num_solpts=4, num_elements_horizontal=3, num_elements_vertical=4)),
    "filters_c21_n5_h2_v2": lambda nm: filters_case(
        nm, "dcmip21_rk3.ini", dict(num_solpts=5, num_elements_horizontal=2, num_elements_vertical=2)),
    # 24 ranks = 2x2 tiles per panel (the reference's own 6 k^2 decomposition)
    "euler3d_tiles24_n3_h2_v2": lambda nm: euler_tiles_case(
        nm, "dcmip31.ini", dict(num_solpts=3, num_elements_horizontal=4, num_elements_vertical=2), 24,
        metric_tiles=(0, 3, 6, 9, 13, 18, 23)),
    # 2-D Cartesian Euler: the plumbing reference (config/gaussian_bubble.ini, smaller grid)
    "cart2d_bubble_n5": lambda nm: cart2d_case(nm, "gaussian_bubble.ini",
                                               dict(num_solpts=5, num_elements_horizontal=7, num_elements_vertical=9)),
    "cart2d_bubble_n4": lambda nm: cart2d_case(nm, "gaussian_bubble.ini",
                                               dict(num_solpts=4, num_elements_horizontal=5, num_elements_vertical=6)),
    # --- round 3: the explicit time loops of BASELINE configs 1-3 (SSP-RK3 on the 2-D Cartesian bubble and on the
    # shallow-water sphere: config/case6.ini at p = 4 = config 2, and at p = 7 = the order of config 3, whose own
    # galewsky.ini cannot initialise in the reference at this commit - BASELINE.md section 2)
    "sw_rk3_c6_n5_h4": lambda nm: sw_case(nm, "case6.ini", dict(num_solpts=5, num_elements_horizontal=4), rk3=(5, 60.0)),
    "sw_rk3_c6_n8_h3": lambda nm: sw_case(nm, "case6.ini", dict(num_solpts=8, num_elements_horizontal=3), rk3=(5, 40.0)),
    "cart2d_rk3_bubble_n4": lambda nm: cart2d_case(nm, "gaussian_bubble.ini",
                                                  dict(num_solpts=4, num_elements_horizontal=5, num_elements_vertical=6),
                                                  rk3=(5, 0.01)),
    # config/gaussian_bubble.ini's own integrator (epi2, the default exponential solver pmex, complex step) at its own dt
    "cart2d_epi2_pmex_bubble_n4": lambda nm: cart2d_case(nm, "gaussian_bubble.ini",
                                                        dict(num_solpts=4, num_elements_horizontal=5, num_elements_vertical=6),
                                                        epi=(2, 3, 5.0, 1e-9), wind=0.01),
    # --- round 3: the remaining templated orders straight from the reference (n = 2 is what config/dcmip31.ini ships
    # with; 6 was pinned through the oracle on synthetic tiles only)
    "euler3d_c31p_n2_h4_v3": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=2, num_elements_horizontal=4, num_elements_vertical=3),
        metric_panels=(2, 4), phase_panels=(), perturb=0.01),
    "euler3d_c31p_n6_h2_v2": lambda nm: euler_case(
        nm, "dcmip31.ini", dict(num_solpts=6, num_elements_horizontal=2, num_elements_vertical=2),
        metric_panels=(1, 5), phase_panels=(), perturb=0.01),
    # (n = 7 cannot be produced: the reference's own DFROperators raises on it - its skew-centrosymmetry check of the
    # differentiation matrix, geometry/operators.py:140-141, fails at that order - so n = 7 stays pinned through the
    # oracle on synthetic tiles, tests/test_synthetic_gpu.py)
    # --- round 3: the benchmark order n = 8 for every caller (the matrix-core instantiations of the JVP, stage and
    # filter kernels), and BASELINE config 5's own combination (dcmip21 + EPI2 + KIOPS + filter)
    "callers_euler3d_n8_h2_v2": lambda nm: callers_case(
        nm, "dcmip31.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2), dt_rk=0.5),
    "filters_c21_n8_h2_v2": lambda nm: filters_case(
        nm, "dcmip21.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2)),
    "steploop_c21_n8_h2_v2": lambda nm: steploop_case(
        nm, "dcmip21_rk3.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2), nsteps=3, dt=0.02),
    "config5_c21_n4_h2_v3": lambda nm: config5_case(
        nm, "dcmip21.ini", dict(num_solpts=4, num_elements_horizontal=2, num_elements_vertical=3)),
    # (n = 8: dt = 10 instead of the ini's 25.  At 25 the adaptive controller sits on a threshold - a relative
    # perturbation of 1e-11 of the Jacobian-vector products, i.e. any other summation order, changes its path from
    # 3 substeps / 192 vectors to 4 / 256 - so exact statistics would pin rounding, not the algorithm; at 10 the path
    # (2 substeps, 6 rejections, 128 vectors, basis at mmax) is the same under perturbations of 1e-9)
    "config5_c21_n8_h2_v2": lambda nm: config5_case(
        nm, "dcmip21.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2), nsteps=2, dt=10.0),
    "solvers_dense": solvers_dense_case,
    # config/case6.ini's own integrator (epi3 + pmex, complex-step JVP, dt = 1800 s) on the shallow-water sphere
    "sw_epi3_pmex_c6_n5_h4": lambda nm: sw_case(nm, "case6.ini", dict(num_solpts=5, num_elements_horizontal=4), epi=(3, 4, 1800.0)),
    # pmex, the schema's default exponential solver (case6.ini, density_current.ini), on the callers fixtures' states
    "pmex_euler3d_n3_h3_v2": lambda nm: pmex_case(
        nm, "callers_euler3d_n3_h3_v2", "dcmip31.ini", dict(num_solpts=3, num_elements_horizontal=3, num_elements_vertical=2)),
    "pmex_euler3d_n8_h2_v2": lambda nm: pmex_case(
        nm, "callers_euler3d_n8_h2_v2", "dcmip31.ini", dict(num_solpts=8, num_elements_horizontal=2, num_elements_vertical=2)),
}


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    names = sys.argv[1:] or list(CASES)
    for nm in names:
        CASES[nm](nm)
