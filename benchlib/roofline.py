"""Roofline bookkeeping of bench.py: compulsory bytes, the committed counter passes (hash-checked), the matrix-core block, the
streaming ceilings measured in the same run."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

ALGO_BYTES_PER_POINT = 384.0  # 360 B/point fields + 24 B/point interface metric
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec

KERNEL_SOURCES = ("wxfactory_amd/csrc/euler3d.hip", "wxfactory_amd/csrc/euler3d_common.h", "wxfactory_amd/csrc/euler3d_extrap.h",
                  "wxfactory_amd/csrc/euler3d_rhs.h", "wxfactory_amd/csrc/euler3d_brick.h", "wxfactory_amd/csrc/euler3d_brick_jvp.h", "wxfactory_amd/csrc/euler3d_jvp.h", "wxfactory_amd/csrc/euler3d_launch.h",
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


PMC_SUMMARIES = {384.0: "r06_pmc_full_summary.json", 312.0: "r06_pmc_rotzero_summary.json"}


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
SQ_COUNTERS = "r06_v3_k2_sq_counters.json"   # (falls back to round 5's pass of the same kernel until this round's is installed)
SQ_COUNTERS_FALLBACK = "r05_v4_k2_sq_counters.json"


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
        name = SQ_COUNTERS if os.path.exists(os.path.join(ROOT, "profiles", SQ_COUNTERS)) else SQ_COUNTERS_FALLBACK
        c = json.load(open(os.path.join(ROOT, "profiles", name)))["wx::euler_rhs_kernel<8, double, false>"]
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy cycles over all 1024 SIMDs
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        blk["counters"] = {"profile": "profiles/" + name, "SQ_INSTS_MFMA_per_launch": c["SQ_INSTS_MFMA"],
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


