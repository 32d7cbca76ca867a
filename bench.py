#!/usr/bin/env python3
"""Headline benchmark: whole-sphere 3-D Euler RHS evaluations (E7 of SURVEY.md section 8d).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete collective R(Q) on the WHOLE cubed sphere (6 panels, n=8 (p=7),
60x60 elements per panel, V vertical elements), inputs resident in HBM, result left in HBM,
halo exchange included.  STRONG scaling: the same sphere is cut into 6 k^2 tiles (the reference's own
decomposition), k the smallest that spreads evenly over the N GPUs (k=1, whole panels, for N = 1, 2, 3, 6;
k=2 for N = 4, 8); at N=1 all six panels live on one GPU and exchange by aliasing.
value = DOF-updates/s = 5 vars * points * 6 panels * evals/s.

Prints ONE JSON line (rank 0) with `roofline` for the dominant kernel (euler_rhs_kernel,
timed live with HIP events on its launch stream) and `cpu_baseline` (the NumPy oracle on a
bounded sample of the same workload, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_POINT = 384.0  # 360 B/point fields + 24 B/point interface metric
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_quota():
    """(cores this process may use at once under its cgroup CPU bandwidth limit, where that was read) - or (None, None).
    A container sees every CPU of the host in /proc/cpuinfo and in its affinity mask and is still throttled to its
    quota: on the MI355X boxes of this pool /sys/fs/cgroup/cpu.max reads "1600000 100000" = 16 cores of a 128-core
    host, and 32 or 64 busy processes get 16 CPU-seconds per second between them (measured)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            return float(quota) / float(period), f"/sys/fs/cgroup/cpu.max = {quota} {period}"
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return quota / period, f"cpu.cfs_quota_us / cpu.cfs_period_us = {quota:.0f} / {period:.0f}"
    except (OSError, ValueError):
        pass
    return None, None


def host_cpu():
    """(model name, physical cores of the host visible to this process, logical CPUs in its affinity mask)."""
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or logical
    except Exception:
        phys = logical
    return model, min(phys, logical), logical


def cpu_baseline_run(flavour, n, H, V, reps, threads, seed, procs=6):
    """one cube panel per process, `procs` processes at once, `threads` OMP/BLAS threads each,
    each timing the CPU restatement (oracle/cpu_bench.py) on an H x H x V-element tile of the E7 workload.  The
    whole-sphere rate is all panels' DOF over the slowest worker's time per evaluation."""
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    ws = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_bench", "--flavour", flavour, "--n", str(n), "--H", str(H),
                            "--V", str(V), "--reps", str(reps), "--threads", str(threads), "--panel", str(p), "--seed",
                            str(seed)], cwd=ROOT, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
          for p in range(procs)]
    try:
        for w in ws:   # setup (synthetic metric, warm-up evaluation) finished everywhere ...
            if w.stdout.readline().strip() != "READY":
                raise RuntimeError("cpu_bench worker failed during setup")
        for w in ws:   # ... then all timed loops start together
            w.stdin.write("go\n")
            w.stdin.flush()
        res = [json.loads(w.stdout.readline()) for w in ws]
    finally:
        for w in ws:
            try:
                w.wait(timeout=30)
            except Exception:
                w.kill()
    slow = max(r["s_per_eval"] for r in res)
    pts = res[0]["dof"] // 5
    return {"dof_updates_per_s": sum(r["dof"] for r in res) / slow, "s_per_eval": round(slow, 4),
            "processes": procs, "threads_per_process": threads, "finite": all(r["finite"] for r in res),
            # what one process moves, on the SURVEY 8d byte count of the synthetic (27-Christoffel) metric: a weak port shows here
            "algorithmic_GBps_per_process": round(ALGO_BYTES_PER_POINT * pts / slow / 1e9, 2),
            "tile": f"n={n}, {H}x{H}x{V} elements per process ({res[0]['dof']} DOF)", "evals_timed": reps}


def cpu_baseline(n, V, seed, H=60):
    """The CPU path beside the GPU figure, as SURVEY.md section 8d writes it: six processes at once, one WHOLE cube panel
    each (H x H x V elements), OMP/BLAS threads = floor(cores / 6) with cores = the physical cores this process can really
    use (the host's count, capped by the container's cgroup CPU quota: cpu_quota), both flavours:
    (2) the optimised C++/OpenMP restatement - `value`; (1) the reference-style dense-Kronecker NumPy restatement (V = 1)."""
    model, phys, logical = host_cpu()
    quota, quota_src = cpu_quota()
    usable = int(min(phys, quota)) if quota else phys      # cores the six processes can really occupy together
    threads = max(1, usable // 6)
    from oracle import c_port

    c_port.load()   # (re)build the C++ port for THIS host once, before six workers would each try to
    # three repeats of the timed sample: `value` is their median, `range` their min-max (the figure swings with what else
    # the host runs: 531-866 M between boxes in round 4)
    runs = sorted((cpu_baseline_run("cpp", n, H, V, 5, threads, seed) for _ in range(3)), key=lambda r: r["dof_updates_per_s"])
    cpp = runs[1]
    dense = cpu_baseline_run("dense", n, H, 1, 3, threads, seed)
    return {"value": cpp["dof_updates_per_s"], "unit": "DOF-updates/s", "cores": 6 * threads, "kind": "port",
            "range": {"min": runs[0]["dof_updates_per_s"], "max": runs[-1]["dof_updates_per_s"], "repeats": len(runs),
                      "value_is": "median"},
            "cpu_model": model, "physical_cores": phys, "logical_cpus": logical, "processes": 6,
            "threads_per_process": threads, "usable_cores": usable,
            "cpu_quota": {"cores": quota, "source": quota_src} if quota else None,
            "sample": f"oracle/c/euler3d_port.cpp (sum-factorised C++/OpenMP, pinned by tests/test_oracle_c.py): six "
                      f"processes x {threads} thread(s) = floor({usable} usable cores / 6) ({phys} physical cores on the host"
                      + (f", cgroup CPU quota {quota:g} cores" if quota else "") + f"), each one whole {H}x{H}x{V}-element "
                      f"panel of the n={n} workload (the E7 sphere), 5 evals after a warm-up, {cpp['s_per_eval']} s/eval on the "
                      f"slowest, {cpp['algorithmic_GBps_per_process']} GB/s per process",
            "flavours": {"cpp_openmp_sum_factorised": cpp,
                         "numpy_dense_kronecker_reference_style": dict(dense, note="oracle/euler3d_dense.py: dense n^3 x n^3 "
                                                                       "operators applied with @ as the reference does "
                                                                       "(operators.py:157-183); E7 at V = 1 per SURVEY 8d")},
            "survey_time_reference": "BASELINE.md section 2: the actual reference, n=8 H=10 V=4, 1.7 M DOF-updates/s per rank "
                                     "(measured in the survey container, not on this host)"}


KERNEL_SOURCES = ("wxfactory_amd/csrc/euler3d.hip", "wxfactory_amd/csrc/euler3d_common.h", "wxfactory_amd/csrc/euler3d_extrap.h",
                  "wxfactory_amd/csrc/euler3d_rhs.h", "wxfactory_amd/csrc/euler3d_jvp.h", "wxfactory_amd/csrc/euler3d_launch.h",
                  "wxfactory_amd/csrc/wx_math.h", "wxfactory_amd/csrc/wx_mfma.h", "wxfactory_amd/csrc/wx_common.h",
                  "wxfactory_amd/csrc/wx_panels.h")


def kernel_source_hash():
    """sha256 over the sources of the 3-D Euler kernels: PMC summaries carry the hash of the build they were measured on."""
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


PMC_SUMMARIES = {384.0: "r05_pmc_full_summary.json", 312.0: "r05_pmc_rotzero_summary.json"}


def pmc_traffic(region, n, H, V, bpp=ALGO_BYTES_PER_POINT):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of tools/kbench.py on one E7 panel, gfx950 corrections applied by tools/pmc_summary.py).
    Counters cannot be read from inside this process, so the figure comes from profiles/<name> - and is REFUSED
    (None, with the reason) unless that summary was measured on exactly the kernel sources this run was built from.
    Returns (bytes or None, provenance dict)."""
    name = PMC_SUMMARIES.get(float(bpp))
    if name is None or region != 0 or (n, H, V) != (8, 60, 8):
        return None, {"reason": "no PMC pass for this launch shape"}
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", name)))
        prov = {"profile": "profiles/" + name, "commit": doc.get("commit"), "source_sha256": doc.get("source_sha256")}
        if doc.get("source_sha256") != kernel_source_hash():
            prov["reason"] = "kernel sources changed since the PMC pass: traffic refused"
            return None, prov
        from wxfactory_amd import _lib

        build = _lib.load().wx_build_info().decode()
        prov["build_info"] = doc.get("build_info")
        if doc.get("build_info") != build:   # an A/B or diagnostic variant (-DWX_MFMA=0, -DWX_K2_DIAG=..) is another kernel
            prov["reason"] = f"PMC pass taken on build '{doc.get('build_info')}', this run is '{build}': traffic refused"
            return None, prov
        k2 = doc["kernels"]["wx::euler_rhs_kernel<8, double, false>"]
        prov["fetch_bytes"], prov["write_bytes"] = k2.get("fetch_bytes"), k2.get("write_bytes")
        return k2["hbm_bytes"], prov
    except (OSError, KeyError, ValueError) as e:
        return None, {"reason": f"{type(e).__name__}: {e}"}


FP64_MFMA_PEAK_TFLOPS = 78.6    # SURVEY 8d; = 512 flop per 16 issue cycles per SIMD (tools/mfma_f64_probe.hip) x 1024 SIMDs x 2.4 GHz
SQ_COUNTERS = "r05_v4_k2_sq_counters.json"


def mfma_block(plans, n, launch_s, elements):
    """The element-local contractions of the dominant kernel on the matrix cores (north star: MFMA utilisation against the
    gfx950 peak).  The flop rate is live: the three directional passes issue 520 `v_mfma_f64_4x4x4_4b_f64` per element at
    n = 8 (csrc/wx_mfma.h, mf4_dir_pass; four 4 x 4 x 4 products = 512 flop each) over the launch time measured in this run.  The pipe's busy fraction comes from the committed counter pass of the same kernel
    (tools/collect_profiles.sh: SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) - which also checks the 520."""
    from wxfactory_amd import _lib

    if n != 8 or not plans:
        return None
    pl = next(iter(plans.values()))
    if not pl.lib.wx_euler3d_uses_matrix_cores(pl._h, _lib.WX_KERNEL_RHS):
        return {"used": False}
    insts = 520.0 * elements
    blk = {"used": True, "instruction": "v_mfma_f64_4x4x4_4b_f64", "instructions_per_launch": insts, "flop_per_instruction": 512,
           "achieved": round(insts * 512 / launch_s / 1e12, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(insts * 512 / launch_s / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
           "note": "the kernel is bound by HBM (roofline.bound): the contractions are 2.6 flop per byte, the matrix pipe is mostly idle by design"}
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", SQ_COUNTERS)))["wx::euler_rhs_kernel<8, double, false>"]
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy cycles over all 1024 SIMDs
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        blk["counters"] = {"profile": "profiles/" + SQ_COUNTERS, "SQ_INSTS_MFMA_per_launch": c["SQ_INSTS_MFMA"],
                           "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"],
                           "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4),
                           "vector_instructions_per_launch": c["SQ_INSTS_VALU"],
                           "wait_fraction_of_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                           "shape": "one E7 panel (tools/kbench.py --rot-zero): the launch shape of the N = 1 line"}
    except (OSError, KeyError, ValueError) as e:
        blk["counters"] = {"reason": f"{type(e).__name__}: {e}"}
    return blk


def copy_ceiling(dev, gib=2, reps=10):
    """What this GPU sustains on a read-once / write-once stream, measured in this run (wx_stream_copy: 16 bytes per lane,
    one pass over `gib` GiB, far beyond the 256 MiB Infinity Cache): the achievable side of the 8 TB/s figure."""
    from wxfactory_amd import _lib

    lib = _lib.load()
    nbytes = gib << 30
    try:
        src = torch.empty(nbytes // 8, dtype=torch.float64, device=dev).normal_()
        dst = torch.empty_like(src)
    except RuntimeError:
        return None
    st = torch.cuda.current_stream(dev).cuda_stream
    ts = []
    for it in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.wx_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, st), "wx_stream_copy")
        b.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(a.elapsed_time(b) * 1e-3)
    ok = bool(torch.equal(src[:1024], dst[:1024]) and torch.equal(src[-1024:], dst[-1024:]))
    # the read side alone (wx_stream_read): a kernel whose traffic is mostly reads is bounded by this rate, not by the copy's
    sink = torch.zeros(lib.wx_stream_read_sink_doubles(), dtype=torch.float64, device=dev)
    tr = []
    for it in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.wx_stream_read(src.data_ptr(), nbytes, sink.data_ptr(), st), "wx_stream_read")
        b.record()
        torch.cuda.synchronize()
        if it >= 2:
            tr.append(a.elapsed_time(b) * 1e-3)
    total, want = float(sink.sum()), float(src.sum())
    read_ok = abs(total - want) <= 1e-9 * float(src.abs().sum())
    del src, dst, sink
    torch.cuda.empty_cache()
    t = sum(ts) / len(ts)
    gbs = 2.0 * nbytes / t / 1e9
    rd = nbytes / (sum(tr) / len(tr)) / 1e9
    return {"kernel": "wx_stream_copy", "bytes_read_plus_written": 2 * nbytes, "launch_ms": round(t * 1e3, 4),
            "achieved": round(gbs, 1), "unit": "GB/s", "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4), "copied_correctly": ok,
            "read_only": {"kernel": "wx_stream_read", "bytes_read": nbytes, "launch_ms": round(sum(tr) / len(tr) * 1e3, 4),
                          "achieved": round(rd, 1), "unit": "GB/s", "frac_of_peak": round(rd / HBM_PEAK_GBS, 4),
                          "sum_correct": bool(read_ok)}}


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
    return {"euler_k2_all_27_christoffel": k2_full_metric(dev, seed), "sw_s7": {"workload": "shallow water, n=8, H=60, 6 panels on 1 GPU (4147200 DOF = 3 x 1382400 points), whole-sphere R(Q) of the Galewsky jet + bump",
                      "us_per_eval": te * 1e6, "dof_updates_per_s": dof / te,
                      "algorithmic_GBps": 156.0 * 6 * H * H * n * n / te / 1e9,
                      "form": ("direct, ONE launch (tile-edge lines pulled from the neighbour tiles' nodal values, no interface buffer)"
                               if rhs._batches[torch.float64].pulls else "direct (ring pack + one launch, no interface buffer)")
                              if rhs._use_direct(torch.float64) else "two kernels (extrapolation + RHS)",
                      "roofline": {"bound": "hbm", "kernels": "sw_rhs_direct_batch_kernel (one R(Q) = one launch)",
                                   "algorithmic_bytes_per_point": 156.0, "achieved": round(156.0 * 6 * H * H * n * n / te / 1e9, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(156.0 * 6 * H * H * n * n / te / 1e9 / HBM_PEAK_GBS, 4),
                                   "profile": "profiles/r05_v4_swbench_kernel_stats.csv, profiles/r05_pmc_sw_summary.json"},
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
        rows.append({"num_solpts": n, "elements": [H, H, V], "dof": 5 * pts, "rhs_ms": round(te * 1e3, 4),
                     "dof_updates_per_s": 5 * pts / te, "matvec_complex_ms": round(tj * 1e3, 4),
                     "algorithmic_GBps": round(plans[0].bytes_per_point * pts / te / 1e9, 1)})
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


def spawn_ranks(n, argv):
    """One process per GPU on this node: python -m torch.distributed.run --nproc-per-node n bench.py <argv>."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def decomposition(world, H, tiles_per_side=0, whole_panels=False):
    """(k, owner of each of the 6 k^2 tiles): the reference's own tiling (process_topology.py:69-94) with the smallest
    k that spreads the tiles evenly over the ranks - k = 1 (whole panels) for 1, 2, 3, 6 GPUs, k = 2 for 4 and 8."""
    from wxfactory_amd.panels import CubeTopology, owner_of_tiles, tiles_per_side_for

    k = tiles_per_side or (1 if whole_panels else tiles_per_side_for(world))
    if H % k:
        raise SystemExit(f"H={H} is not divisible by {k} tiles per panel side")
    return k, owner_of_tiles(world, CubeTopology(k).ntiles)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # >= 50 calls ...
    ap.add_argument("--warmup", type=int, default=10)   # ... after >= 5 warm-ups
    ap.add_argument("--n", type=int, default=8, help="num_solpts (p = n-1)")
    ap.add_argument("--H", type=int, default=60, help="elements per panel side")
    ap.add_argument("--V", type=int, default=8, help="vertical elements")
    ap.add_argument("--seed", type=int, default=20250824)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--whole-panels", action="store_true", help="one tile per panel even when 6 does not divide N")
    ap.add_argument("--tiles-per-side", type=int, default=0, help="force k (6 k^2 tiles); default: chosen from N")
    ap.add_argument("--event-every", type=int, default=1,
                    help="record HIP events around the kernel launches of every K-th timed step (the live kernel timing of "
                         "the roofline block; 1 = every launch, so that the rocprofv3 average of the same command covers the "
                         "same launches in the same state - with K > 1 the unmarked launches overlap their tails)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (shallow-water S7) measurement")
    ap.add_argument("--loopback", action="store_true",
                    help="rehearsal on one GPU: route every edge message through the RCCL collective (1-rank group) and split "
                         "the evaluation into INTERIOR / BOUNDARY launches, as a multi-GPU run does")
    ap.add_argument("--exchange", choices=("rccl", "torch"), default="rccl",
                    help="halo exchange of the several-GPU path: 'rccl' = the library's own behind the C ABI (wx_exchange_*: "
                         "grouped ncclSend / ncclRecv on a communication stream, event fork / join; the whole evaluation of a "
                         "rank is one wx_euler3d_rhs_overlapped call), 'torch' = torch.distributed.all_to_all_single")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the several-rank program flow on ONE GPU: every rank uses device 0 (needs --exchange "
                         "torch - gloo through host copies -: RCCL refuses two ranks on one device); not a measurement")
    ap.add_argument("--metric", choices=("true", "synthetic"), default="true",
                    help="static metric fields: the cubed-sphere metric of the DCMIP 3-1 planet from wxfactory_amd.geometry3d "
                         "(default, SURVEY 8d) or SURVEY's seeded synthetic fields; values do not affect speed")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` typed as is: start one rank per GPU through torch.distributed.run as a CHILD
        # (nothing has touched the GPU yet: never re-exec a process that has), relay its output, exit with its code
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # before the first call that initialises the GPU runtime (the host driver supports dmabuf IPC only: without this RCCL's
    # peer mapping fails with hipIpcGetMemHandle: invalid argument); ranks started by an external torchrun land here too
    if world > 1 or args.loopback:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")   # a communicator that cannot connect says why ...
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # ... on stderr (RCCL's default is stdout: the ONE line's stream)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.one_device:
        if args.exchange != "torch" and world > 1:
            raise SystemExit("--one-device: RCCL refuses two ranks on one device, use --exchange torch")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # a HOST-side process group (gloo) for what happens around the measurement: the 128-byte id of the library's
        # communicator, barriers, the max over ranks, the independent route of the exchange self-check.  NO NCCL process group:
        # the halo exchange and every reduction of the data path run on the library's own communicator (wx_comm_*), and a
        # torch NCCL group would add a watchdog thread issuing HIP calls beside the captures (profiles/r05_process_group_abort.md)
        # (gloo announces its connections on STDOUT: fd 1 points at fd 2 meanwhile - this program's stdout is the ONE line)
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    from wxfactory_amd import _lib, synthetic
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.panels import CubeTopology
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    _lib.load()
    n, H, V = args.n, args.H, args.V
    ops = synthetic.dfr_ops(n)
    # the sphere is cut into 6 k^2 tiles (the reference's own decomposition, process_topology.py:69-94) with the
    # smallest k that spreads evenly over the ranks: k = 1 (whole panels) for 1, 2, 3, 6 GPUs, k = 2 for 4 and 8
    k, owner = decomposition(world, H, args.tiles_per_side, args.whole_panels)
    topo = CubeTopology(k)
    Ht = H // k
    mine = [t for t, r in enumerate(owner) if r == rank]
    plans, qs = {}, {}
    t_setup = time.perf_counter()
    for t in mine:
        if args.metric == "true":
            from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch

            panel, row, col = topo.locate(t)
            metric = metric3d_torch(CubedSphere3DTile(n, Ht, V, panel, 10000.0, 31, row=row, col=col, k=k), dev)
        else:
            metric = synthetic.euler3d_metric(n, Ht, V, t, dev, args.seed)
        plans[t] = Euler3DPlan(n, Ht, V, 31, topo.locate(t)[0], ops, metric, on_panel_edge=topo.on_panel_edge(t))
        prc = topo.locate(t)
        qs[t] = synthetic.euler3d_state(n, Ht, V, prc[0], dev, args.seed, row=prc[1], col=prc[2], k=k)   # a cut of the PANEL's state
    t_setup = time.perf_counter() - t_setup
    edge_doubles = 5 * V * Ht * n * n  # WX_EULER3D_EDGE_FIELDS
    comm = None
    exchange_report = {"backend": "none needed: one rank owns every tile, the halos alias the packed edge buffers"}

    def all_ranks(flag: bool) -> bool:
        """True when `flag` holds on every rank (one all-reduce of the process group; the decision is the same everywhere)."""
        if world == 1 or not dist.is_initialized():
            return flag
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)   # (gloo: host tensors)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def torch_exchange():
        # torch.distributed.all_to_all_single on the gloo group, staged through host copies (exchange.py): the independent
        # second route of the self-check and the fallback - slow, correct, pinned by tests/test_exchange_gloo.py
        return PanelExchange(edge_doubles, dev, rank=rank, world_size=world, tiles_per_side=k, backend="torch")

    ex = None
    dog = None
    if args.exchange == "rccl" and (world > 1 or args.loopback):
        # a watchdog over the set-up and the first evaluation of the library's exchange: a rank that never reaches the
        # collective communicator set-up, or a send without its receive, becomes a message on stderr and exit code 4
        # instead of a silent timeout
        import threading

        def hung():
            sys.stderr.write(f"bench.py rank {rank}: the RCCL exchange (communicator set-up / first grouped ncclSend + ncclRecv) "
                             "did not finish in 300 s; rerun with --exchange torch (host-staged gloo)\n")
            sys.stderr.flush()
            os._exit(4)

        dog = threading.Timer(300.0, hung)
        dog.daemon = True
        dog.start()
        # the library's own exchange (wx_comm_*, wx_exchange_*).  Its first run on several GPUs is the driver's: if the
        # communicator or the buffers cannot be set up on some rank, every rank falls back to the host-staged gloo route and
        # the line says so, instead of the whole scaling run being lost
        from wxfactory_amd.exchange import RcclComm

        why = None
        # RCCL prints its version banner on STDOUT when NCCL_DEBUG is set (at the first communicator's creation), whatever
        # NCCL_DEBUG_FILE says: fd 1 points at fd 2 for the duration of the set-up - this program's stdout is the ONE line
        sys.stdout.flush()
        keep_fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = RcclComm(rank, world, device=dev)   # (the unique id travels through the gloo group; one rank needs none)
            ex = PanelExchange(edge_doubles, dev, rank=rank, world_size=world, tiles_per_side=k, loopback=args.loopback,
                               backend="rccl", comm=comm)
        except Exception as e:   # noqa: BLE001 - reported in the line
            why = f"{type(e).__name__}: {e}"
        finally:
            sys.stdout.flush()
            os.dup2(keep_fd1, 1)
            os.close(keep_fd1)
        if all_ranks(why is None):
            exchange_report = {"backend": "rccl behind the C ABI (wx_exchange_*: grouped ncclSend / ncclRecv, event fork / join)"}
        else:
            ex, comm = None, None
            exchange_report = {"backend": "gloo all_to_all_single through host copies", "fell_back_from": "rccl behind the C ABI",
                               "reason": why or "set-up failed on another rank"}
    elif world > 1:
        exchange_report = {"backend": "gloo all_to_all_single through host copies (--exchange torch)"}
    elif args.loopback:
        raise SystemExit("--loopback rehearses the library's exchange on one GPU: it needs --exchange rccl")
    if ex is None:
        ex = torch_exchange()
    rhs = RhsEuler3D(plans, ex, overlap=not args.no_overlap)

    if getattr(ex, "_native", None) is not None:
        # self-check before anything is timed: the same state through the library's exchange and through an independent
        # route must give the same R bit for bit on every rank.  Several ranks: gloo's all_to_all_single through host copies
        # (the route tests/test_exchange_gloo.py pins against the reference's halos).  One rank in loopback mode: the
        # aliasing exchange (no message moves).
        probe = torch.stack([qs[t] for t in mine]) if mine else qs
        got = rhs(probe)
        torch.cuda.synchronize()
        if world > 1:
            rhs_t = RhsEuler3D(plans, torch_exchange(), overlap=not args.no_overlap)
            route = "gloo all_to_all_single through host copies"
        else:
            rhs_t = RhsEuler3D(plans, PanelExchange(edge_doubles, dev, rank=0, world_size=1, tiles_per_side=k))
            route = "aliasing exchange of one rank"
        want = rhs_t(probe)
        torch.cuda.synchronize()
        same = all_ranks(bool(torch.equal(got, want)) if mine else True)
        exchange_report["selfcheck"] = (f"R(Q) bit-identical to the {route} on every rank" if same else
                                        f"MISMATCH against the {route}: timed on that route instead")
        if not same:
            exchange_report["backend"] = route
            exchange_report["fell_back_from"] = "rccl behind the C ABI"
            rhs, ex = rhs_t, rhs_t.ex
        del got, want, probe
    if dog is not None:
        dog.cancel()

    # live timing of the dominant kernel: HIP events on the launch stream around every K2 launch
    ev = []
    orig_rhs = Euler3DPlan.rhs

    def timed_rhs(self, q, halo, out, region=_lib.WX_REGION_ALL):
        if not recording[0]:
            return orig_rhs(self, q, halo, out, region)
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        orig_rhs(self, q, halo, out, region)
        b.record()
        ev.append((a, b, region, 1))

    ev1 = []
    orig_pack = Euler3DPlan.extrap_pack

    def timed_pack(self, q, send):
        if not recording[0]:
            return orig_pack(self, q, send)
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        orig_pack(self, q, send)
        b.record()
        ev1.append((a, b))

    # (tiles of the 24-tile layout go through one launch per phase for all local tiles: time that launch instead)
    from wxfactory_amd.rhs_euler3d import Euler3DBatch

    orig_brhs, orig_bpack = Euler3DBatch.rhs, Euler3DBatch.extrap_pack

    def timed_brhs(self, q, out, region, *a, **kw):
        if not recording[0]:
            return orig_brhs(self, q, out, region, *a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_brhs(self, q, out, region, *a, **kw)
        e1.record()
        ev.append((e0, e1, region, len(self.panels)))

    def timed_bpack(self, q, *a, **kw):
        if not recording[0]:
            return orig_bpack(self, q, *a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_bpack(self, q, *a, **kw)
        e1.record()
        ev1.append((e0, e1))

    recording = [False]
    Euler3DPlan.rhs = timed_rhs
    Euler3DPlan.extrap_pack = timed_pack
    Euler3DBatch.rhs, Euler3DBatch.extrap_pack = timed_brhs, timed_bpack
    # the state of a rank: its tiles stacked in one tensor (what a time loop holds), so that small tiles can share launches
    state = torch.stack([qs[t] for t in mine]) if mine else qs

    def barrier():
        if world > 1:
            dist.barrier()

    # the exchange behind the C ABI: a rank's whole evaluation is ONE wx_euler3d_rhs_overlapped call (whole panels; the
    # tiles of the 24-tile layout share launches through the batch instead), which the Python-side event pairs above never
    # see - the call stamps the reference's nine-slot timing row itself (wx_exchange_set_timer), one timer per timed step
    import ctypes

    native_timers = []
    one_call = (getattr(ex, "_native", None) is not None and bool(mine) and not rhs._small_tiles() and not args.no_overlap)
    if one_call:
        lib = _lib.load()
        for _ in range(args.steps):
            h = ctypes.c_void_p()
            _lib.check(lib.wx_phase_timer_create(ctypes.byref(h)), "wx_phase_timer_create")
            native_timers.append(h)

    out = None
    for _ in range(args.warmup):
        out = rhs(state)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # HIP events around every kernel launch of every `event_every`-th timed step (default: every step).  An event pair
        # is two marker packets between kernels that would otherwise follow each other directly and overlap their tails:
        # 24 pairs per evaluation cost it 1.1 % (6.80-6.82 against 6.73-6.74 ms, profiles/r05_bench_spread.txt); kept at every
        # step because the rocprofv3 averages of the same command then describe the same launches in the same state - with
        # events on every fourth step the two differed by 4 % (profiles/r03_v12_event_sampling.txt)
        recording[0] = i % args.event_every == 0
        if one_call:
            lib.wx_exchange_set_timer(ex._native, native_timers[i] if recording[0] else None)
        out = rhs(state)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    recording[0] = False
    native_rows = []
    if one_call:
        lib.wx_exchange_set_timer(ex._native, None)
        for i, h in enumerate(native_timers):
            if i % args.event_every == 0:
                row = (ctypes.c_double * 9)()
                _lib.check(lib.wx_phase_timer_elapsed(h, row), "wx_phase_timer_elapsed")
                native_rows.append(list(row))
            lib.wx_phase_timer_destroy(h)
    if mine:
        chk = float(out.abs().amax(dim=(1, 2, 3, 4, 5)).sum())
        if not (chk == chk and chk < float("inf")):
            raise SystemExit("non-finite RHS in the benchmark")

    tmax = torch.tensor([dt], dtype=torch.float64)   # (gloo: host tensors)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ranks_seen = dist.get_world_size() if world > 1 else 1
    if ranks_seen != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} ranks")
    # What makes an N-rank line checkable against the N = 1 line: the whole-sphere R of the last timed step, per variable
    # sum, sum of magnitudes and largest magnitude (every rank's tiles, all-reduced).  The sphere, its state and its
    # metric do not depend on the decomposition (the state is a cut of the panel's, synthetic.euler3d_state), and R does
    # not to 1e-13 (tests: test_result_does_not_depend_on_the_decomposition), so `sum` must agree between any two lines to
    # 1e-13 x abs_sum, max_abs to 1e-13 relative.
    local = torch.zeros((3, 5), dtype=torch.float64, device=dev)
    if mine:
        local[0] = out.sum(dim=(0, 2, 3, 4, 5))
        local[1] = out.abs().sum(dim=(0, 2, 3, 4, 5))
        local[2] = out.abs().amax(dim=(0, 2, 3, 4, 5))
    local = local.cpu()
    if world > 1:
        sums = local[:2].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        mx = local[2].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        local = torch.cat((sums, mx[None]))
    checksum = {"of": "whole-sphere R(Q) of the last timed step, per variable (rho, rho u1, rho u2, rho w, rho theta)",
                "sum": local[0].tolist(), "abs_sum": local[1].tolist(), "max_abs": local[2].tolist(),
                "agreement": "between decompositions: |sum - sum'| <= 1e-13 abs_sum, max_abs to 1e-13 relative"}

    # per-rank phase times (outside the timed region): the reference's nine RHS timestamps (rhs/rhs.py:88-118) on
    # HIP events of the launch stream, five more evaluations
    Euler3DPlan.rhs, Euler3DPlan.extrap_pack = orig_rhs, orig_pack
    Euler3DBatch.rhs, Euler3DBatch.extrap_pack = orig_brhs, orig_bpack
    rhs.timed = True   # (the timed evaluation launches tile by tile: each phase of each tile gets its own stamps)
    rhs.clear_timings()
    for _ in range(5):
        rhs(state)
    torch.cuda.synchronize()
    rhs.retrieve_last_times()
    rhs.timed = False
    tm = rhs.timings[1:] or rhs.timings
    mean = lambda i: round(sum(t[i] for t in tm) / len(tm) * 1e3, 4) if tm else None  # noqa: E731
    mine_phase = {"rank": rank, "tiles": len(mine), "pack_ms": mean(0), "exchange_start_ms": mean(1), "interior_ms": mean(2),
                  "exchange_ms": mean(4), "boundary_ms": mean(5), "total_ms": mean(8)}
    per_rank = [mine_phase]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_phase)
    barrier()

    pts_panel = V * H * H * n**3
    evals_per_s = args.steps / dt
    dof_per_s = 5 * pts_panel * 6 * evals_per_s

    # dominant-kernel roofline (rank 0's launches)
    roof = None
    if ev or native_rows:
        by_region, tiles_in_launch = {}, {}
        for a, b, region, ntl in ev:
            by_region.setdefault(region, []).append(a.elapsed_time(b) * 1e-3)
            tiles_in_launch[region] = ntl
        for row in native_rows:   # seconds between the stamps 0 1 2 3 5 8: pack, start, INTERIOR, -, join, -, -, BOUNDARY
            by_region.setdefault(_lib.WX_REGION_INTERIOR, []).append(row[2] / len(mine))
            by_region.setdefault(_lib.WX_REGION_BOUNDARY, []).append(row[7] / len(mine))
            tiles_in_launch[_lib.WX_REGION_INTERIOR] = tiles_in_launch[_lib.WX_REGION_BOUNDARY] = 1
        w = Ht - 2 if Ht > 2 else 0
        frac_of_panel = {_lib.WX_REGION_ALL: 1.0, _lib.WX_REGION_INTERIOR: (w * w) / (Ht * Ht),
                         _lib.WX_REGION_BOUNDARY: 1.0 - (w * w) / (Ht * Ht)}
        # the dominant launch shape: ALL at N=1, INTERIOR when the exchange is overlapped
        region = max(by_region, key=lambda r: sum(by_region[r]))
        tk = sum(by_region[region]) / len(by_region[region])
        # compulsory bytes of THIS launch: SURVEY 8d's 384 B/point, minus the 72 B/point of the nine rotation
        # Christoffel fields when the plan found them identically zero (non-rotating planet) and skips them
        bpp = next(iter(plans.values())).bytes_per_point if plans else ALGO_BYTES_PER_POINT
        bytes_launch = bpp * (pts_panel / (k * k)) * frac_of_panel[region] * tiles_in_launch[region]
        achieved = bytes_launch / tk / 1e9
        traffic, traffic_src = pmc_traffic(region, n, Ht, V, bpp)
        roof = {"bound": "hbm", "kernel": "euler_rhs_kernel<8,double>", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src, "launch_ms": round(tk * 1e3, 4),
                "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_point": bpp,
                "survey_bytes_per_point": ALGO_BYTES_PER_POINT,
                "region": {0: "all", 1: "interior", 2: "boundary"}[region], "tiles_per_launch": tiles_in_launch[region]}
        if ev1:
            roof["extrap_kernel_launch_ms"] = round(sum(a.elapsed_time(b) for a, b in ev1) / len(ev1), 4)
        if native_rows:
            roof["extrap_kernel_launch_ms"] = round(sum(r[0] for r in native_rows) / len(native_rows) / len(mine) * 1e3, 4)
            roof["timing_source"] = ("wx_phase_timer stamps inside wx_euler3d_rhs_overlapped (one host call per evaluation): "
                                     "the launches of a phase follow each other on the compute stream, launch time = phase / tiles")
        # the whole sweep (extrapolation kernel + exchange + fused kernel, every local tile): compulsory bytes of one
        # R(Q) of this rank's tiles (each static field and Q read once, R written once) over the step time
        sweep_bytes = bpp * (pts_panel / (k * k)) * len(mine)
        sweep = sweep_bytes / (dt / args.steps) / 1e9
        roof["sweep"] = {"algorithmic_bytes": sweep_bytes, "achieved": round(sweep, 1), "frac": round(sweep / HBM_PEAK_GBS, 4),
                         "note": "rank 0's tiles; the interface buffer's round trip through HBM and the extrapolation "
                                 "kernel's second read of Q are not compulsory bytes"}
        roof["sweep_frac"] = roof["sweep"]["frac"]
        # the north star's target (>= 50 % of the HBM roofline on the rhs_euler sweep), judged on the bytes this plan is
        # COMPELLED to move - nothing it skips is counted
        roof["sweep"]["target"] = {"source": "BASELINE.json north_star: >= 50 % of MI355X HBM roofline on the rhs_euler sweep",
                                   "bytes_per_point": bpp, "frac": roof["sweep"]["frac"],
                                   "dof_updates_per_s_this_rank": 5 * (pts_panel / (k * k)) * len(mine) / (dt / args.steps),
                                   "met": bool(roof["sweep"]["frac"] >= 0.5)}
        roof["matrix_cores"] = mfma_block(plans, n, tk, (pts_panel / (k * k)) / n**3 * frac_of_panel[region] * tiles_in_launch[region])
        roof["ceiling"] = copy_ceiling(dev)
        if roof["ceiling"] and traffic:
            on_traffic = traffic / tk / 1e9
            ceil = roof["ceiling"]
            roof["on_measured_traffic"] = {"GBps": round(on_traffic, 1), "frac_of_copy_rate": round(on_traffic / ceil["achieved"], 4),
                                           "traffic_over_algorithmic": round(traffic / bytes_launch, 4)}
            fb, wb = traffic_src.get("fetch_bytes"), traffic_src.get("write_bytes")
            rd = ceil.get("read_only", {}).get("achieved")
            if fb and wb and rd:
                # the time this launch's own mix of reads and writes would take at the measured streaming rates: its reads at
                # the read-only rate, its writes at the rate the copy's writes are left with once its reads are priced so
                n_copy = ceil["bytes_read_plus_written"] / 2.0
                t_copy_writes = ceil["launch_ms"] * 1e-3 - n_copy / (rd * 1e9)
                if t_copy_writes > 0:
                    wr = n_copy / t_copy_writes / 1e9
                    floor_s = tiles_in_launch[region] * (fb / (rd * 1e9) + wb / (wr * 1e9))
                    roof["on_measured_traffic"].update({"read_rate_GBps": rd, "implied_write_rate_GBps": round(wr, 1),
                                                        "streaming_floor_ms": round(floor_s * 1e3, 4),
                                                        "frac_of_streaming_floor": round(floor_s / tk, 4)})

    if rank == 0:
        line = {
            "metric": "DOF-updates/s (whole-sphere 3-D Euler RHS evals, cubed sphere p=7, 60x60 elem/panel)",
            "value": dof_per_s, "unit": "DOF-updates/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rhs_evals_per_s": evals_per_s, "ranks_seen": ranks_seen, "per_rank": per_rank, "checksum": checksum,
            "rccl_version": _lib.load().wx_comm_rccl_version(),
            "hip_runtime_version": _lib.load().wx_hip_runtime_version(),   # what the process BOUND (inside torch: the wheel's)
            "process_group": "gloo (host side only: id bootstrap, barriers, max over ranks); no NCCL process group" if world > 1
                             else "none",
            **({"rehearsal": "--one-device: every rank on GPU 0, halos through gloo and host copies - program flow only, NOT a "
                             "measurement"} if args.one_device else {}),
            "config": {"workload": f"E7: 3-D Euler RHS, n={n} (p={n-1}), H={H}x{H} elem/panel, V={V}, 6 panels "
                                   f"({5*pts_panel*6} DOF), halo exchange included",
                       "n": n, "H": H, "V": V, "tiles": topo.ntiles, "tile_H": Ht, "tiles_per_gpu": len(mine),
                       "parallelism": f"tile-dd{min(world, topo.ntiles)}",
                       "overlap": not args.no_overlap,
                       "exchange": exchange_report,
                       "metric": "geometry3d: equiangular cubed sphere, DCMIP 3-1 planet (R/125), ztop 10 km"
                                 if args.metric == "true" else "seeded synthetic fields (SURVEY 8d)",
                       "metric_setup_s": round(t_setup, 1)},
            "roofline": roof,
        }
        if args.gpus == 1 and not args.no_extras:
            line["extra"] = extras(dev, args.seed)
            line["extra"]["euler_callers"] = caller_extras(rhs, qs)
            line["extra"]["euler_ini_sizes"] = ini_size_extras(dev, args.seed)
            if args.metric == "true" and not args.loopback:
                line["extra"]["euler_column_metric"] = column_metric_extras(plans, mine, state, out, edge_doubles, dev, k,
                                                                            dt / args.steps, args.steps)
            del rhs, qs, plans, out
            torch.cuda.empty_cache()
            line["extra"]["euler_e7_v1"] = e7_v1_extras(dev, args.seed)
            line["extra"]["epi2_kiops_e7"] = epi2_kiops_e7_extras(dev, args.seed)
            line["extra"]["rhs_benchmark_matrix"] = rhs_benchmark_matrix(dev, args.seed)
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, V, args.seed, H)
        print(json.dumps(line), flush=True)
    # teardown in dependency order: nothing in flight, then the library's exchange and communicator, then the process group
    torch.cuda.synchronize()
    barrier()
    try:
        ex.close()
    except Exception:   # noqa: BLE001
        pass
    if comm is not None:
        comm.close()   # (closes every exchange made on it first: twins, value / tangent sets, the stage pipeline's)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
