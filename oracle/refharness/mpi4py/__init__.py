from . import MPI  # noqa: F401
