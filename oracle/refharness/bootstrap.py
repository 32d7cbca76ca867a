"""Import harness for the *reference* WxFactory (golden-vector generation only).

TEST INFRASTRUCTURE, build-container only.  Follows SURVEY.md Appendix A: the
reference tree is imported in place from /root/reference/wx_factory (never
copied, never written to); obstacles of this image are worked around without
touching a reference file:

* no mpi4py            -> threaded stand-in ``oracle/refharness/mpi4py``
* Python 3.10          -> ``typing.Self`` alias
* compiler.compile_kernels crashes on this CPU string / would write into the
  tree -> a three-function module whose ``load_module`` returns the reference's
  own ``pde_cpp`` extension, compiled by ``oracle/Makefile`` straight from
  /root/reference/wx_factory/pde/interface.cpp into ``oracle/_ref/``.
"""
import importlib
import os
import subprocess
import sys
import types
import typing

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.dirname(HERE)
REF = os.environ.get("WX_REFERENCE", "/root/reference")
REF_PKG = os.path.join(REF, "wx_factory")
REF_OUT = os.path.join(ORACLE, "_ref")


def bootstrap(n_ranks: int = 6):
    if not os.path.isdir(REF_PKG):
        raise RuntimeError(f"reference tree not found at {REF_PKG} (this harness only runs in the build container)")
    sys.dont_write_bytecode = True  # keep /root/reference pristine
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ["FAKE_MPI_SIZE"] = str(n_ranks)
    if not hasattr(typing, "Self"):
        typing.Self = typing.TypeVar("Self")

    subprocess.check_call(["make", "-s", "-C", ORACLE, "ref"])

    for p in (REF_PKG, REF_OUT, HERE):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)

    pkg = types.ModuleType("compiler")
    pkg.__path__ = []
    ck = types.ModuleType("compiler.compile_kernels")
    ck.compile = lambda *a, **k: None
    ck.load_module = lambda name, kind: importlib.import_module("pde_cpp")
    ck.clean = lambda *a, **k: None
    pkg.compile_kernels = ck
    sys.modules["compiler"] = pkg
    sys.modules["compiler.compile_kernels"] = ck

    from mpi4py import MPI  # the stand-in

    if MPI.COMM_WORLD.size != n_ranks:
        MPI.reset_world(n_ranks)
    return MPI
