"""wxfactory_amd - MI355X-native RHS / JVP engine for WxFactory's cubed-sphere DFR discretisation.

Host side (Python, mirrors the reference's rhs / device / process_topology interfaces for this
one path) over the C ABI of libwxhip.so (include/wxhip.h), which holds the hand-written HIP
kernels.  torch is plumbing only: device memory, streams, torch.distributed.
"""
from . import _lib  # noqa: F401
from .panels import NEIGHBOR, landing_edge, panels_of_rank, owner_of_panels  # noqa: F401

__all__ = ["_lib", "NEIGHBOR", "landing_edge", "panels_of_rank", "owner_of_panels"]
__version__ = "0.1.0"
