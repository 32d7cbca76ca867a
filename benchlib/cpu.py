"""The CPU path timed beside the GPU figure (bench.py: cpu_baseline) - the only part of the benchmark that runs anything under oracle/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_POINT = 384.0  # 360 B/point fields + 24 B/point interface metric


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


