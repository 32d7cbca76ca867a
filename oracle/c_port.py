"""ctypes front end of the C++/OpenMP CPU restatement (oracle/c/euler3d_port.cpp).

TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Same call shape as oracle.euler3d.Euler3DOracle (extrapolate / pack_edges / rhs), float64 and complex128; pack_edges is the
NumPy oracle's.  Built by `make -C oracle port` (or build() below, which bench.py and __graft_entry__.build() call).
"""
import ctypes
import os
import subprocess

import numpy

from .euler3d import Euler3DOracle

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "c", "libwxoracle.so")
_P = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile into a private file and rename it into place: several processes may find the library stale at once
    (bench.py starts six workers), and none of them must ever load a half-written file."""
    src = os.path.join(HERE, "c", "euler3d_port.cpp")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        tmp = f"{LIB}.{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O3", "-march=native", "-fno-math-errno", "-fopenmp", "-shared", "-std=c++17", "-fPIC", src, "-o", tmp, "-lmvec", "-lm"])
        os.replace(tmp, LIB)
    return LIB


_lib = None


def _host_stamp() -> str:
    """What decides whether a -march=native build can be reused here: the CPU's instruction-set flags (the model-name
    string is shared by different micro-architectures on virtual machines) and the source text."""
    import hashlib

    flags = ""
    if os.path.exists("/proc/cpuinfo"):
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = " ".join(sorted(line.split(":", 1)[1].split()))
                break
    src = open(os.path.join(HERE, "c", "euler3d_port.cpp"), "rb").read()
    return hashlib.sha256(flags.encode() + b"\0" + src).hexdigest()


def load():
    global _lib
    if _lib is None:
        # -march=native code does not travel between hosts: rebuild when the library was built elsewhere or from
        # another source text (the stamp is written here only, so a library made by `make port` is rebuilt once)
        stamp = LIB + ".host"
        want = _host_stamp()
        if not os.path.exists(LIB) or not os.path.exists(stamp) or open(stamp).read() != want:
            build(force=True)
            with open(f"{stamp}.{os.getpid()}.tmp", "w") as f:
                f.write(want)
            os.replace(f"{stamp}.{os.getpid()}.tmp", stamp)
        _lib = ctypes.CDLL(LIB)
        for fn in ("wxo_euler3d_extrapolate", "wxo_euler3d_rhs", "wxo_euler3d_extrapolate_c", "wxo_euler3d_rhs_c",
                   "wxo_euler3d_extrapolate_generic", "wxo_euler3d_rhs_generic"):
            getattr(_lib, fn).restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(_P) if a is not None else None   # (complex128 arrays: interleaved re, im doubles)


def _c(a, dtype=numpy.float64):
    return numpy.ascontiguousarray(a, dtype=dtype)


class Euler3DPortC(Euler3DOracle):
    """The NumPy oracle's interface over the C++ kernels (extrapolate, rhs); threads = OpenMP threads per call.
    float64 or complex128 states (the complex instantiation follows NumPy's abs / maximum rules, so that
    Im R(Q + i eps v) / eps is the reference's complex-step JVP)."""

    def __init__(self, *args, threads: int = 0, generic: bool = False, **kw):
        """generic: float64 calls take the scalar, type-generic instantiation (what complex128 states always run) instead
        of the vectorised float64 path - the cross-check of the two (tests/test_oracle_c.py)."""
        super().__init__(*args, **kw)
        self.threads = int(threads)
        self.generic = bool(generic)
        self.lib = load()
        self._mc = None   # contiguous copies of the metric, made on the first rhs() (extrapolation needs none)
        self._ops = [_c(x) for x in (self.em, self.ep, self.D, self.C, self.HF)]

    def extrapolate(self, q):
        n, H, V = self.n, self.H, self.V
        dt = numpy.complex128 if numpy.iscomplexobj(q) else numpy.float64
        q = _c(q, dt)
        itf = [numpy.empty((5, V, H, H, 2 * n * n), dtype=dt) for _ in range(3)]
        fn = self.lib.wxo_euler3d_extrapolate_c if dt is numpy.complex128 else (
            self.lib.wxo_euler3d_extrapolate_generic if self.generic else self.lib.wxo_euler3d_extrapolate)
        rc = fn(n, H, V, _p(self._ops[0]), _p(self._ops[1]), _p(q), _p(itf[0]), _p(itf[1]), _p(itf[2]), self.threads)
        assert rc == 0
        return itf

    def rhs(self, q, halo, itf=None, want=None):
        n, H, V = self.n, self.H, self.V
        dt = numpy.complex128 if numpy.iscomplexobj(q) else numpy.float64
        q = _c(q, dt)
        if itf is None:
            itf = self.extrapolate(q)
        itf = [_c(x, dt) for x in itf]
        halo = [_c(x, dt) for x in halo]
        if self._mc is None:
            m = self.m
            self._mc = {k: _c(m[k]) for k in ("sqrtG_new", "h_contra_new", "christoffel", "inv_dzdeta_new", "sqrtG_itf_i_new",
                                                "sqrtG_itf_j_new", "sqrtG_itf_k_new", "h_contra_itf_i_new",
                                                "h_contra_itf_j_new", "h_contra_itf_k_new")}
            self._damp = (_c(m["damp_coef"]), _c(m["damp_uref"])) if self.case_number in (21, 22) else (None, None)
        mc = self._mc
        out = numpy.empty_like(q)
        fn = self.lib.wxo_euler3d_rhs_c if dt is numpy.complex128 else (
            self.lib.wxo_euler3d_rhs_generic if self.generic else self.lib.wxo_euler3d_rhs)
        rc = fn(
            n, H, V, self.case_number, _p(self._ops[2]), _p(self._ops[3]), _p(self._ops[4]), _p(q), _p(itf[0]), _p(itf[1]),
            _p(itf[2]), _p(halo[0]), _p(halo[1]), _p(halo[2]), _p(halo[3]), _p(mc["sqrtG_new"]), _p(mc["h_contra_new"]),
            _p(mc["christoffel"]), _p(mc["inv_dzdeta_new"]), _p(mc["sqrtG_itf_i_new"]), _p(mc["sqrtG_itf_j_new"]),
            _p(mc["sqrtG_itf_k_new"]), _p(mc["h_contra_itf_i_new"]), _p(mc["h_contra_itf_j_new"]),
            _p(mc["h_contra_itf_k_new"]), _p(self._damp[0]), _p(self._damp[1]), _p(out), self.threads)
        assert rc == 0
        return out
