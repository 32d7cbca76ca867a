"""State checkpoint wire format and the global (rank-count independent) layout.

Byte-compatible with reference wx_factory/output/state.py:9-33: a NumPy `.npy` block of the GLOBAL state
`(6, nvar, [V,] H_tot, H_tot, n^2 | n^3)`, then one line with the schema version, then the
configuration text.  With one tile per panel the global array is the panels stacked in panel order
(what process_topology.py:444-469 `gather_cube` produces on 6 ranks), so files written here restart a
WxFactory run at any rank count and vice versa (reference tests/unit/restart/test_restart.py:107-151).
"""
from typing import List, Optional, Tuple

import numpy
import torch
import torch.distributed as dist

from .panels import panels_of_rank


def save_state(state, state_version, config_content: str, output_file_name: str) -> None:
    """output/state.py:9-16 - `state` is the global array (numpy or torch)."""
    a = state.detach().cpu().numpy() if isinstance(state, torch.Tensor) else numpy.asarray(state)
    with open(output_file_name, "wb+") as f:
        numpy.save(f, a)
        f.write(bytes(f"{state_version}\n", "utf-8"))
        f.write(bytes(config_content, "utf-8"))


def load_state(input_file_name: str) -> Tuple[numpy.ndarray, str, str]:
    """output/state.py:19-33 - returns (global state, state_version, config text); parsing the config
    is the caller's business (the reference builds a Configuration from it)."""
    with open(input_file_name, "rb") as f:
        state = numpy.load(f)
        version = str(f.readline(), "utf-8").strip()
        # (the reference joins lines that still carry their newline, which doubles every line break;
        #  configparser does not care - here the text comes back as it was written)
        config = "".join(str(line, "utf-8") for line in f.readlines()).strip()
    return state, version, config


def gather_cube(local: torch.Tensor, rank: int = 0, world_size: int = 1, group=None) -> Optional[torch.Tensor]:
    """Panels owned by this rank, stacked (P_local, ...) -> global (6, ...) on rank 0, None elsewhere."""
    if world_size == 1:
        return local
    mine = panels_of_rank(rank, world_size)
    pieces: List = [None] * world_size
    dist.all_gather_object(pieces, (mine, local.detach().cpu().numpy() if len(mine) else None), group=group)
    if rank != 0:
        return None
    out = [None] * 6
    for panels, arr in pieces:
        for i, p in enumerate(panels):
            out[p] = arr[i]
    return torch.from_numpy(numpy.stack(out))


def distribute_cube(global_state, rank: int = 0, world_size: int = 1, device="cpu", group=None) -> torch.Tensor:
    """Global (6, ...) array on rank 0 -> this rank's panels stacked (process_topology.py:471-539 for one
    tile per panel)."""
    if world_size > 1:
        box = [global_state if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        global_state = box[0]
    g = global_state.detach().cpu().numpy() if isinstance(global_state, torch.Tensor) else numpy.asarray(global_state)
    if g.shape[0] != 6:
        raise ValueError(f"This is not a cube: leading dimension {g.shape[0]} != 6")
    mine = panels_of_rank(rank, world_size)
    return torch.from_numpy(numpy.ascontiguousarray(g[mine])).to(device)
