#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950
FETCH_SIZE reports one half of the bytes of a coalesced streaming read (calibrated here on
euler_extrap_kernel, whose read volume is exactly 5 fields x points x 8 B); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "wx::" in r["Kernel_Name"]:
                out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


if __name__ == "__main__":
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fb = 2.0 * fetch.get(k, 0.0) * 1024.0   # gfx950: FETCH_SIZE counts 64 B per 128 B request
        wb = write.get(k, 0.0) * 1024.0
        res[k] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                  "raw_FETCH_SIZE_KiB": fetch.get(k), "raw_WRITE_SIZE_KiB": write.get(k)}
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from benchlib.roofline import kernel_source_hash  # the hash bench.py checks before it quotes these bytes

    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = os.environ.get("WX_COMMIT")  # the GPU box has no .git: pass the commit in
    from wxfactory_amd import _lib   # the library the profiled program loaded (WXHIP_LIB selects a variant)

    json.dump({"unit": "bytes per launch (mean over launches)", "correction": "FETCH_SIZE x2 (gfx950), KiB -> B",
               "commit": commit, "source_sha256": kernel_source_hash(), "build_info": _lib.load().wx_build_info().decode(),
               "kernels": res}, open(sys.argv[3], "w"), indent=1)
    for k, v in res.items():
        print(f"{k:50s} fetch {v['fetch_bytes']/1e9:7.3f} GB  write {v['write_bytes']/1e9:7.3f} GB  total {v['hbm_bytes']/1e9:7.3f} GB")
