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

The parts: benchlib/launch.py (rank start-up), headline.py (the timed region and the line), roofline.py, extras.py, cpu.py.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def decomposition(world, H, tiles_per_side=0, whole_panels=False):
    """(k, owner of each of the 6 k^2 tiles) - benchlib.headline.decomposition."""
    from benchlib.headline import decomposition as d

    return d(world, H, tiles_per_side, whole_panels)


def main():
    if "WORLD_SIZE" not in os.environ and _gpus(sys.argv[1:]) > 1:
        # `python bench.py --gpus N` typed as is: one rank per GPU, started as CHILDREN of this process, which has imported
        # neither torch nor anything else that opens the GPU (never re-exec a process that has; benchlib/launch.py on why the
        # ranks are not started through torch.distributed.run); relay the first failure's exit code
        from benchlib.launch import spawn_ranks

        sys.exit(spawn_ranks(_gpus(sys.argv[1:]), sys.argv[1:], os.path.abspath(__file__)))
    from benchlib.headline import main as headline

    headline()


if __name__ == "__main__":
    main()
