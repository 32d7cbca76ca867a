"""The secondary measurements of bench.py (`extra` of the line; N = 1 only, outside the timed region)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

from .roofline import ALGO_BYTES_PER_POINT, HBM_PEAK_GBS, copy_ceiling, mfma_block, pmc_traffic  # noqa: F401


def k2_full_metric(dev, seed, n=8, H=60, V=8, reps=20):
    """The fused RHS kernel on one E7 panel whose 27 Christoffel fields are all non-zero (SURVEY 8d's synthetic
    metric = a rotating planet): the full 384 B/point configuration, where nothing is skipped at plan time."""
    from wxfactory_amd import _lib, synthetic
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), synthetic.euler3d_metric(n, H, V, 0, dev, seed))
    q = synthetic.euler3d_state(n, H, V, 0, dev, seed)
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=dev)
    sp = [send[e] for e in range(4)]
    out = torch.empty_like(q)
    ts = []
    for it in range(reps + 3):
        plan.extrap_pack(q, sp)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        plan.rhs(q, sp, out, _lib.WX_REGION_ALL)
        b.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append(a.elapsed_time(b) * 1e-3)
    tk = sum(ts) / len(ts)
    bpp = plan.bytes_per_point
    gbs = bpp * V * H * H * n**3 / tk / 1e9
    return {"workload": "one E7 panel, seeded synthetic metric (SURVEY 8d)", "launch_ms": round(tk * 1e3, 4),
            "algorithmic_bytes_per_point": bpp, "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(0, n, H, V, bpp)[0], "traffic_source": pmc_traffic(0, n, H, V, bpp)[1]}


def k2_case21(dev, n=8, H=60, V=8, reps=20, panel=0):
    """The fused RHS kernel on one E7 panel of DCMIP 2-1 (BASELINE config 5's physics): the
    Schaer mountain - a metric that depends on the level, so the general kernel and not the column form - and the Rayleigh
    sponge above 20 km (pde_euler_cubesphere.py:285-288, init/dcmip.py:676-757): four more fields per point, 344 B/point.  Panel 0
    holds the mountain.  Parity at this size: tests/test_fullsize_gpu.py (case 21)."""
    from wxfactory_amd import _lib, synthetic
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    t = CubedSphere3DTile(n, H, V, panel, 30000.0, 21, topo=topography_for_case(21, planet_for_case(21)[0]))
    plan = Euler3DPlan(n, H, V, 21, panel, synthetic.dfr_ops(n), metric3d_torch(t, dev))
    q = torch.from_numpy(initial_state(t)).to(dev)
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=dev)
    sp = [send[e] for e in range(4)]
    out = torch.empty_like(q)
    ts = []
    for it in range(reps + 3):
        plan.extrap_pack(q, sp)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        plan.rhs(q, sp, out, _lib.WX_REGION_ALL)
        b.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append(a.elapsed_time(b) * 1e-3)
    tk = sum(ts) / len(ts)
    bpp = plan.bytes_per_point
    pts = V * H * H * n**3
    gbs = bpp * pts / tk / 1e9
    res = {"workload": "one E7 panel of DCMIP 2-1 (case 21): Schaer mountain + Rayleigh sponge, own geometry and initial state",
           "kernel": "euler_rhs_kernel<8,double>", "launch_ms": round(tk * 1e3, 4), "algorithmic_bytes_per_point": bpp,
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": bpp * pts, "traffic": None}}
    del plan, out, send
    return res


def ini_size_extras(dev, seed):
    """The 3-D Euler configs of BASELINE.json at the sizes their .ini files ship with (config/dcmip31.ini: n = 2,
    12 x 12 x 3 elements per panel; config/dcmip21.ini: n = 3, 3 x 3 x 4): launch-bound, evaluated with one launch
    per phase for all six panels (wx_euler3d_batch_*)."""
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.matvec import matvec_fun, matvec_rat
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd import synthetic

    out = {}
    for label, case, n, H, V, ztop, dt_ini in (("dcmip31.ini", 31, 2, 12, 3, 10000.0, 30.0),
                                               ("dcmip21.ini", 21, 3, 3, 4, 30000.0, 25.0)):
        topo = topography_for_case(case, planet_for_case(case)[0])
        plans, q = {}, []
        for p in range(6):
            t = CubedSphere3DTile(n, H, V, p, ztop, case, topo=topo)
            plans[p] = Euler3DPlan(n, H, V, case, p, synthetic.dfr_ops(n), metric3d_torch(t, dev))
            q.append(torch.from_numpy(initial_state(t)).to(dev))
        Q = torch.stack(q)
        rhs = RhsEuler3D(plans)
        R = rhs(Q)
        v = (torch.rand(Q.shape, device=dev, dtype=Q.dtype) - 0.5).flatten()

        def clock(fn, reps=300):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / reps * 1e6, 1)

        out[label] = {"n": n, "elements_per_panel": [H, H, V], "dof": Q.numel(), "rhs_us": clock(lambda: rhs(Q)),
                      "matvec_complex_us": clock(lambda: matvec_fun(v, 1.0, Q, R, rhs, "complex")),
                      "matvec_rat_us": clock(lambda: matvec_rat(v, 1.0, Q, R, rhs))}
        # the step both files configure: time_integrator = epi2 (KIOPS + complex-step JVP), tolerance 1e-7, their dt
        from wxfactory_amd.integrators import Epi

        epi, Qs, ts = Epi(2, rhs, tol=1e-7), Q, []
        for i in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Qs = epi.step(Qs, dt_ini)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[label].update(epi2_dt_s=dt_ini, epi2_step_ms=round(sorted(ts[3:])[2] * 1e3, 2),
                          epi2_krylov_vectors=int(epi.solver_info["iterations"]))
        # ... and with the schema's default exponential solver (config-format.json: exponential_solver = pmex)
        epx, Qp, tp = Epi(2, rhs, tol=1e-7, exponential_solver="pmex"), Q, []
        for i in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Qp = epx.step(Qp, dt_ini)
            torch.cuda.synchronize()
            tp.append(time.perf_counter() - t0)
        out[label].update(epi2_pmex_step_ms=round(sorted(tp[2:])[2] * 1e3, 2),
                          epi2_pmex_krylov_vectors=int(epx.solver_info["iterations"]))
        # BASELINE config 4's integrator at this size: Rosenbrock-2 + FGMRES (integrators/ros2.py:24-81), the Gram-Schmidt step on
        # the device and one read-back per pass of Krylov vectors (include/wxhip.h: wx_fgmres_vector)
        from wxfactory_amd.integrators import Ros2

        ros, Qr, tr = Ros2(rhs, tol=1e-7, gmres_restart=30), Q, []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Qr = ros.step(Qr, dt_ini)
            torch.cuda.synchronize()
            tr.append(time.perf_counter() - t0)
        si = ros.solver_info
        med = sorted(tr[1:])[1]
        out[label].update(ros2_fgmres_step_ms=round(med * 1e3, 2), ros2_fgmres_iterations=int(si["iterations"]),
                          ros2_fgmres_us_per_iteration=round(med / max(1, si["iterations"]) * 1e6, 1),
                          ros2_fgmres_device_passes=si.get("device_passes"), ros2_fgmres_wasted_vectors=si.get("wasted_vectors"),
                          one_kernel_form=bool(plans[0].one_kernel))
    return out


def extras(dev, seed):
    """Secondary, non-headline numbers on the same GPU: the shallow-water S7 workload of BASELINE.json's
    galewsky line (n=8, 60x60 elements/panel, 6 panels), whole-sphere R(Q)."""
    from wxfactory_amd import synthetic
    from wxfactory_amd.geometry import CubedSphereTile2D, metric2d_torch
    from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan

    n, H = 8, 60
    ops = synthetic.dfr_ops(n)
    # the true equiangular-gnomonic metric (wxfactory_amd/geometry.py, pinned against the reference's) and the Galewsky
    # jet + bump of BASELINE.json's galewsky line (own implementation, wxfactory_amd/initial_sw.py: the reference's cannot run)
    from wxfactory_amd.initial_sw import galewsky, galewsky_h0

    tiles = [CubedSphereTile2D(n, H, p) for p in range(6)]
    plans = {p: SwPlan(n, H, p, ops, metric2d_torch(tiles[p], dev)) for p in range(6)}
    h0 = galewsky_h0(tiles[0].earth_radius, tiles[0].rotation_speed)
    Q = torch.stack([torch.from_numpy(galewsky(t, True, h0)).to(dev) for t in tiles])
    rhs = RhsShallowWater(plans)
    # (an evaluation is 50 us: 100 of them are over before the chip has left its idle clocks - 50 untimed ones first, then 400)
    for _ in range(50):
        rhs(Q)
    torch.cuda.synchronize()
    reps = 400
    t0 = time.perf_counter()
    for _ in range(reps):
        rhs(Q)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / reps
    dof = 3 * 6 * H * H * n * n
    return {"euler_k2_all_27_christoffel": k2_full_metric(dev, seed), "euler_k2_case21": k2_case21(dev), "sw_s7": {"workload": "shallow water, n=8, H=60, 6 panels on 1 GPU (4147200 DOF = 3 x 1382400 points), whole-sphere R(Q) of the Galewsky jet + bump",
                      "us_per_eval": te * 1e6, "dof_updates_per_s": dof / te,
                      "algorithmic_GBps": 156.0 * 6 * H * H * n * n / te / 1e9,
                      "form": ("direct, ONE launch (tile-edge lines pulled from the neighbour tiles' nodal values, no interface buffer)"
                               if rhs._batches[torch.float64].pulls else "direct (ring pack + one launch, no interface buffer)")
                              if rhs._use_direct(torch.float64) else "two kernels (extrapolation + RHS)",
                      "roofline": {"bound": "hbm", "kernels": "sw_rhs_direct_batch_kernel (one R(Q) = one launch)",
                                   "algorithmic_bytes_per_point": 156.0, "achieved": round(156.0 * 6 * H * H * n * n / te / 1e9, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(156.0 * 6 * H * H * n * n / te / 1e9 / HBM_PEAK_GBS, 4),
                                   "profile": "profiles/r06_v3_swbench_kernel_stats.csv, profiles/r06_pmc_sw_summary.json"},
                      "note": "all six panels in ONE launch (wx_sw_batch_rhs_direct, wx_sw_batch_direct_pulls = 1): own face states "
                              "from LDS, the neighbours' from the neighbour elements' nodal values - across panel edges too (sum, "
                              "rotation and flip of the neighbour panel's line formed in place); 5.5 MB of state per panel"}}


def rhs_benchmark_matrix(dev, seed):
    """The reference's own RHS benchmark matrix (tests/rhs_benchmark/run.sh:67-71): 3-D Euler, DCMIP 3-1, 6 ranks,
    (num_solpts, horizontal, vertical elements per panel) = (2,30,30) (3,20,20) (4,15,15) (5,12,12) (6,10,10), i.e.
    60^3 points per panel at every order; here all six panels on one GPU, whole-sphere R(Q) and the complex-step
    matvec the benchmark's epi2 + KIOPS integrator calls."""
    from wxfactory_amd import synthetic
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    rows = []
    for n, H, V in ((2, 30, 30), (3, 20, 20), (4, 15, 15), (5, 12, 12), (6, 10, 10)):
        plans, q = {}, []
        for p in range(6):
            t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
            plans[p] = Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metric3d_torch(t, dev))
            q.append(torch.from_numpy(initial_state(t)).to(dev))
        Q = torch.stack(q)
        rhs = RhsEuler3D(plans)
        R = rhs(Q)
        v = (torch.rand(Q.shape, device=dev, dtype=Q.dtype) - 0.5).flatten()

        def clock(fn, reps=50):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        te, tj = clock(lambda: rhs(Q)), clock(lambda: matvec_fun(v, 1.0, Q, R, rhs, "complex"))
        pts = 6 * V * H * H * n**3
        bpp = plans[0].bytes_per_point   # this ORDER's compulsory bytes (the interface metric is 192 / n B/point)
        gbs = bpp * pts / te / 1e9
        rows.append({"num_solpts": n, "elements": [H, H, V], "dof": 5 * pts, "rhs_ms": round(te * 1e3, 4),
                     "dof_updates_per_s": 5 * pts / te, "matvec_complex_ms": round(tj * 1e3, 4),
                     "form": "one kernel (bricks, csrc/euler3d_brick.h)" if plans[0].one_kernel else "two kernels",
                     "algorithmic_GBps": round(gbs, 1),
                     "roofline": {"bound": "hbm", "of": "whole R(Q): every launch of the evaluation", "achieved": round(gbs, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                  "algorithmic_bytes_per_point": bpp}})
        del rhs, plans, Q, R, v, q
        torch.cuda.empty_cache()
    return rows


def column_metric_extras(plans, mine, state, out_general, edge_doubles, dev, k, general_s, steps):
    """The same whole-sphere R(Q) with the plans' opt-in column form of the metric (Euler3DPlan(column_metric="auto"):
    on the benchmark's shallow atmosphere without topography every metric array is the same on all levels, to rounding; the
    fused kernel then reads one (n x n) slab per column and field instead of V n of them).  NOT the headline: the
    headline kernels take the metric arrays as the reference hands them over, whatever they hold."""
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    col = {}
    for t in mine:
        p = plans[t]
        col[t] = Euler3DPlan(p.n, p.H, p.V, p.case_number, p.panel, p._ops, p._metric, on_panel_edge=p.on_panel_edge,
                             column_metric="auto")
    if not all(pl.column_metric for pl in col.values()):
        return {"applies": False, "note": "the metric arrays differ between levels"}
    rhs = RhsEuler3D(col, PanelExchange(edge_doubles, dev, rank=0, world_size=1, tiles_per_side=k))
    for _ in range(3):
        o = rhs(state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        o = rhs(state)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    scale = out_general.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
    diff = float(((o - out_general).abs() / scale).max())
    # ... and the prepared complex-step matvec on the same plans (their dual twins take the slabs too)
    from wxfactory_amd.matvec import ComplexStepOperator

    op = ComplexStepOperator(1.0, state, o, rhs)
    v = (torch.rand(state.shape, device=dev, dtype=state.dtype) - 0.5).flatten()
    for _ in range(3):
        op(v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        op(v)
    torch.cuda.synchronize()
    tm = (time.perf_counter() - t0) / 10
    rhs.jvp_release()
    return {"applies": True, "ms_per_eval": round(t * 1e3, 4), "dof_updates_per_s": state.numel() / t,
            "speedup_over_headline": round(general_s / t, 3), "max_rel_difference_from_headline_result": diff,
            "roofline": column_roofline(col, mine, state, dev),
            "matvec_fun_complex_prepared_ms": round(tm * 1e3, 3),
            "note": "opt-in plan form for column-invariant geometries (include/wxhip.h: wx_euler3d_plan_set_column_metric); "
                    "the RHS and the complex-step JVP kernels, whole-tile and split launches; the stage kernels read the "
                    "full arrays"}


def column_roofline(col, mine, state, dev, reps=10):
    """The column form's fused kernel on one tile: what it is COMPELLED to move is Q, R and the interface values only (the
    metric arrives as one (n x n) slab per column and field: 1 / (V n) of the full arrays) - 80 B/point + the slabs - and
    at that traffic it is no longer bound by memory but by the vector pipe (profiles/r03_column_k2_sq_counters.json)."""
    from wxfactory_amd import _lib

    t0 = mine[0]
    pl = col[t0]
    q = state.reshape((len(mine),) + tuple(pl.shape))[0]
    send = torch.zeros((4, pl.edge_count), dtype=torch.float64, device=dev)
    sp = [send[e] for e in range(4)]
    out = torch.empty_like(q)
    ts = []
    for it in range(reps + 2):
        pl.extrap_pack(q, sp)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        pl.rhs(q, sp, out, _lib.WX_REGION_ALL)
        b.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(a.elapsed_time(b) * 1e-3)
    tk = sum(ts) / len(ts)
    pts = q.numel() // 5
    slab_bpp = (pl.bytes_per_point - 80.0) / (pl.V * pl.n)          # every metric field once per column instead of per level
    bpp = 80.0 + slab_bpp
    gbs = bpp * pts / tk / 1e9
    return {"bound": "valu", "kernel": "euler_rhs_column_kernel<8>", "launch_ms": round(tk * 1e3, 4),
            "algorithmic_bytes_per_point": round(bpp, 2), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
            "note": "vector-pipe bound at this traffic: 62 % of the issue slots, 17 % of the instructions are f64 FMAs "
                    "(profiles/r03_column_k2_sq_counters.json; with every load served from cache it still takes 0.65 ms: profiles/r05_k2_latency_ceiling.txt); the fraction of the HBM peak is reported for scale only"}


def epi2_kiops_e7_extras(dev, seed, n=8, H=60, V=2, dt=0.5, steps=3):
    """BASELINE config 5 at the benchmark's resolution: EPI2 + KIOPS (complex-step JVP, tol 1e-7) on the whole sphere at
    n = 8, 60 x 60 elements per panel, V = 2 vertical elements (a Krylov basis of 64 vectors of the V = 8 sphere does not
    fit on one GPU beside the metric), DCMIP 3-1 + 1 % perturbation: time per step and per Krylov vector, beside the bare
    prepared matvec - i.e. what the solver adds around the kernels (wx_kiops_long_*: 11 vector sweeps per Krylov vector)."""
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.integrators import Epi
    from wxfactory_amd.matvec import ComplexStepOperator
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd import synthetic

    gen = torch.Generator(device=dev).manual_seed(seed)
    plans, qs = {}, []
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
        plans[p] = Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metric3d_torch(t, dev))
        q = torch.from_numpy(initial_state(t)).to(dev)
        qs.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5)))
    Q = torch.stack(qs)
    rhs = RhsEuler3D(plans)
    R = rhs(Q)
    op = ComplexStepOperator(dt, Q, R, rhs)
    v = torch.randn(Q.numel(), generator=gen, device=dev, dtype=torch.float64)
    for _ in range(3):
        op(v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        op(v)
    torch.cuda.synchronize()
    mv = (time.perf_counter() - t0) / 10
    rhs.jvp_release()
    del op, v
    epi = Epi(2, rhs, tol=1e-7)
    Q = epi.step(Q, dt)   # first step: basis and workspace allocation
    rows = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Q = epi.step(Q, dt)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        rows.append((t, int(epi.solver_info["iterations"])))
    t, it = min(rows)
    return {"workload": f"EPI2 + KIOPS, n={n}, {H}x{H}x{V} elements/panel, 6 panels ({Q.numel()} DOF), dt={dt} s, tol 1e-7",
            "step_ms": round(t * 1e3, 1), "krylov_vectors": it, "ms_per_krylov_vector": round(t / it * 1e3, 3),
            "prepared_matvec_ms": round(mv * 1e3, 3), "solver_overhead_over_matvec": round(t / it / mv - 1.0, 3),
            "finite": bool(torch.isfinite(Q).all())}


def e7_v1_extras(dev, seed, n=8, H=60):
    """E7 is reported at V in {1, 8}; the headline is V = 8, this is the whole-sphere R(Q) at V = 1
    (one vertical element, 8 levels; 6 x 1.8 M points: the six panels go in one launch per phase)."""
    from wxfactory_amd import synthetic
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    ops = synthetic.dfr_ops(n)
    plans = {p: Euler3DPlan(n, H, 1, 31, p, ops, metric3d_torch(CubedSphere3DTile(n, H, 1, p, 10000.0, 31), dev))
             for p in range(6)}
    Q = torch.stack([synthetic.euler3d_state(n, H, 1, p, dev, seed) for p in range(6)])
    rhs = RhsEuler3D(plans)
    for _ in range(5):
        rhs(Q)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        rhs(Q)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / reps
    pts = 6 * H * H * n**3
    return {"workload": f"E7 at V=1: n={n}, {H}x{H}x1 elements/panel, 6 panels ({5*pts} DOF), whole-sphere R(Q)",
            "ms_per_eval": round(te * 1e3, 4), "dof_updates_per_s": 5 * pts / te,
            "algorithmic_GBps": round(plans[0].bytes_per_point * pts / te / 1e9, 1)}


def jvp_kernel_rooflines(rhs, Q, v, reps=10):
    """Per-kernel roofline blocks of the prepared complex-step matvec (solvers/matvec.py:56-61), kernel time from HIP
    events on the launch stream, one panel per launch: euler_jvp_kernel (reads Q, v, the static fields, cached face
    values and this product's face tangents; stores the real tangent) and the tangent extrapolation in front of it."""
    from wxfactory_amd import _lib

    eps = 1.4901161193847656e-08
    rhs.jvp_prepare(Q)
    try:
        plans, exv, ext = rhs._jvp_plans(), rhs._ex_val, rhs._ex_tan
        p = rhs.panels[0]
        shp = (len(rhs.panels),) + tuple(rhs.panel_shape)
        q0, v0 = Q.reshape(shp)[0], v.reshape(shp)[0]
        out = torch.empty_like(q0)
        t1, t2 = [], []
        for it in range(reps + 2):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            plans[p].jvp_tangent_pack(q0, v0, eps, ext.send_views(p))
            e1.record()
            plans[p].jvp_prepared(q0, v0, eps, exv.halo_views(p), ext.halo_views(p), out, 1.0 / eps, _lib.WX_REGION_ALL)
            e2.record()
            torch.cuda.synchronize()
            if it >= 2:
                t1.append(e0.elapsed_time(e1))
                t2.append(e1.elapsed_time(e2))
        pts = q0.numel() // 5
        n = plans[p].n
        static = plans[p].bytes_per_point - 80.0          # the RHS kernel's static fields of this plan
        blocks = {}
        for name, ms, bpp in (("euler_jvp_kernel", sum(t2) / len(t2), 40.0 + 40.0 + static + 40.0),
                              ("tangent_extrapolation (euler_tan_extrap_kernel)", sum(t1) / len(t1), 16.0 + 40.0 + 240.0 / n)):
            gbs = bpp * pts / (ms * 1e-3) / 1e9
            blocks[name] = {"bound": "hbm", "launch_ms": round(ms, 4), "algorithmic_bytes_per_point": bpp,
                            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        blocks["euler_jvp_kernel"]["matrix_cores"] = bool(
            plans[p].lib.wx_euler3d_uses_matrix_cores(plans[p]._h, _lib.WX_KERNEL_JVP))
        return blocks
    finally:
        rhs.jvp_release()


def caller_extras(rhs, qs, reps=5):
    """the JVP variants and one explicit step on the SAME plans / metric as the headline (N = 1)."""
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.matvec import matvec_fun, matvec_rat

    Q = torch.stack([qs[p] for p in sorted(qs)])
    g = torch.Generator(device=Q.device).manual_seed(1)
    v = (torch.rand(Q.shape, generator=g, device=Q.device, dtype=Q.dtype) - 0.5) * 1e-3 * Q.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
    R = rhs(Q)
    dt = 1.0

    def timeit(fn):
        for _ in range(2):  # lazy state (twin plans, second interface slot, allocator blocks) is built here
            fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        return sorted(times)[len(times) // 2]  # median

    rhs.jvp_prepare(Q)   # what a Krylov solve does once per linearisation state (ComplexStepOperator)
    prepared = timeit(lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "complex"))
    rhs.jvp_release()
    out = {"matvec_fun_complex_prepared_ms": prepared,
           "matvec_fun_complex_ms": timeit(lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "complex")),
           "matvec_fun_fd_ms": timeit(lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "fd")),
           "matvec_rat_ms": timeit(lambda: matvec_rat(v.flatten(), dt, Q, R, rhs))}
    stepper = Tvdrk3(rhs)
    state = {"q": Q}

    def step():
        state["q"] = stepper.step(state["q"], 1e-3)

    # 424 B/point compulsory for a complex-step / dual JVP (Q, v, Jv, 35 static fields, interface metric);
    # minus 72 B/point where the plan skips the nine identically-zero rotation Christoffel fields
    pts = Q.numel() // 5
    bpp = 424.0 - (384.0 - next(iter(rhs.plans.values())).bytes_per_point)
    gbs = bpp * pts / (out["matvec_fun_complex_prepared_ms"] * 1e-3) / 1e9
    out["matvec_roofline"] = {"bound": "hbm", "what": "whole-sphere prepared complex-step matvec (tangent extrapolation + JVP kernel, "
                              "6 panels)", "algorithmic_bytes_per_point": bpp, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    out["jvp_kernels"] = jvp_kernel_rooflines(rhs, Q, v)
    out["tvdrk3_step_ms"] = timeit(step)
    out["tvdrk3_mode"] = "pipelined" if stepper.pipeline else ("fused" if stepper.fused else "plain")
    out["note"] = ("whole sphere, same plans as the headline: complex-step JVP = fused dual-number kernels (wx_euler3d_jvp), "
                   "prepared = face values of Q cached once per Krylov solve, tangents only per product (wx_euler3d_jvp_prepare); "
                   "fd / Rosenbrock operator = shifted state formed on load + difference formed in the store; SSP-RK3 step "
                   "= 3 pipelined stages (wx_euler3d_stage)")
    return {k: (round(x, 3) if isinstance(x, float) else x) for k, x in out.items()}


