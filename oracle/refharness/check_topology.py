"""Pin the GENERATOR of the exchange fixtures: the reference's own ProcessTopologyTest (13 cases: scalar and vector
exchanges of 1-d to 4-d fields in several shapes, tolerance 1e-15; /root/reference/tests/unit/common/test_process_topology.py:
78-545) run on six ranks of the threaded mpi4py stand-in that gen_golden.py runs the reference under.  If the stand-in's
Create_dist_graph_adjacent / Ineighbor_alltoall / Split delivered anything else than MPI does, these cases - which compare
what a rank receives with what its neighbour's convert_contra / flip tables say it must receive - fail.

TEST INFRASTRUCTURE, build container only (the reference cannot travel).  Prints one line per case and "ALL 13 PASSED"."""
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bootstrap import REF, bootstrap  # noqa: E402

CASES = ["vector2d_1d_shape1d", "vector2d_1d_shape2d", "vector2d_2d_shape1d", "vector2d_2d_shape3d", "vector3d_1d_shape1d",
         "vector3d_1d_shape2d", "vector3d_3d_shape1d", "vector3d_4d_shape3d", "scalar_1d_shape1d", "scalar_1d_shape2d",
         "scalar_1d_shape3d", "scalar_2d_shape1d", "scalar_2d_shape2d"]


def main():
    MPI = bootstrap(6)
    sys.path.insert(0, REF)          # the reference's `tests` package (tests/unit/mpi_test.py)
    import importlib

    mod = importlib.import_module("tests.unit.common.test_process_topology")
    failures = []
    for case in CASES:
        errs = [None] * 6

        def body(r, case=case, errs=errs):
            try:
                t = mod.ProcessTopologyTest(case)
                t.setUp()
                getattr(t, case)()
            except BaseException:   # noqa: BLE001 - unittest's AssertionError, SkipTest, anything
                errs[r] = traceback.format_exc()

        MPI.run_ranks(body, 6)
        bad = [(r, e) for r, e in enumerate(errs) if e]
        print(f"{case}: {'ok on 6 ranks' if not bad else 'FAILED on ranks ' + str([r for r, _ in bad])}", flush=True)
        failures += [(case, r, e) for r, e in bad]
    for case, r, e in failures:
        print(f"--- {case}, rank {r}\n{e}")
    if failures:
        return 1
    print("ALL 13 PASSED")
    return 0


if __name__ == "__main__":
    sys.exit(main())
