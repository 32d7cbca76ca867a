"""Build libwxhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m wxfactory_amd.build [--force]

The library has no Python / torch dependency: it is the C-ABI drop-in boundary
declared in include/wxhip.h.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libwxhip.so")
SOURCES = ["wx_api.hip", "euler3d.hip", "sw2d.hip", "cart2d.hip", "filters.hip", "krylov.hip", "exchange.hip"]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, "..", "include", "wxhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), out: str = LIB) -> str:
    """defines/out: build an experiment variant (e.g. defines=["WX_MFMA=0"], out=".../libwxhip_valu.so");
    pick it at run time with the WXHIP_LIB environment variable."""
    if out == LIB and not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    tag = os.path.basename(out).replace(".so", "")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objs = []
    procs = []
    # an object is rebuilt when its own source, any header, or the switches (defines / extra flags) changed
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(PKG, "..", "include", "wxhip.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    switches = " ".join(list(defines) + os.environ.get("WX_HIPCC_EXTRA", "").split())
    for src in SOURCES:
        obj = os.path.join(LIBDIR, tag + "_" + src.replace(".hip", ".o"))
        objs.append(obj)
        stamp = obj + ".switches"
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == switches
                 and os.path.getmtime(obj) > max(newest_header, os.path.getmtime(os.path.join(CSRC, src))))
        if fresh:
            continue
        for stale in (stamp, obj):   # (a failed compile must not leave the old object beside a stamp that matches)
            if os.path.exists(stale):
                os.remove(stale)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
        cmd += ["-D" + d for d in defines]
        cmd += os.environ.get("WX_HIPCC_EXTRA", "").split()   # development: extra compiler flags for an experiment variant
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, stamp, subprocess.Popen(cmd)))
    failed = []
    for cmd, stamp, p in procs:
        if p.wait() != 0:
            failed.append(" ".join(cmd))
        else:
            open(stamp, "w").write(switches)   # the stamp follows a successful compile only
    if failed:
        raise RuntimeError("hipcc failed: " + "; ".join(failed))
    # RCCL (the halo exchange, csrc/exchange.hip): librccl.so.1 - inside a torch process the copy torch has loaded already
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs + ["-L/opt/rocm/lib", "-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build(force="--force" in sys.argv, verbose=True, defines=defs,
                out=os.path.join(LIBDIR, outs[0]) if outs else LIB))
