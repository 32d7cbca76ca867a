"""State checkpoint wire format and the global (rank-count independent) layout.

Byte-compatible with reference wx_factory/output/state.py:9-33: a NumPy `.npy` block of the GLOBAL state
`(6, nvar, [V,] H_tot, H_tot, n^2 | n^3)`, then one line with the schema version, then the
configuration text.  The global array is assembled from the 6 k^2 tiles of the run (k x k per panel,
process_topology.py:444-539: `gather_cube` / `distribute_cube`), so files written here restart a WxFactory
run at any rank count and vice versa (reference tests/unit/restart/test_restart.py:107-151).
gather_cube / distribute_cube move TENSORS (torch.distributed gather / scatter to and from rank 0, RCCL
on GPUs, gloo on CPU): only rank 0 ever holds the global state, as in the reference.
"""
from typing import Optional, Tuple

import numpy
import torch
import torch.distributed as dist

from .panels import CubeTopology, owner_of_tiles


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


def _tile_slices(topo: CubeTopology, t: int, Ht: int):
    """Index of tile t inside the global array (6, ..., H_tot, H_tot, points): panel, rows (axis -3), cols (axis -2)."""
    p, r, c = topo.locate(t)
    return (p, Ellipsis, slice(r * Ht, (r + 1) * Ht), slice(c * Ht, (c + 1) * Ht), slice(None))


def _layout(world_size: int, tiles_per_side: int):
    topo = CubeTopology(tiles_per_side)
    owner = owner_of_tiles(world_size, topo.ntiles)
    per_rank = [[t for t in range(topo.ntiles) if owner[t] == r] for r in range(world_size)]
    return topo, per_rank, max(len(x) for x in per_rank)


def gather_cube(local: torch.Tensor, rank: int = 0, world_size: int = 1, group=None,
                tiles_per_side: int = 1) -> Optional[torch.Tensor]:
    """Tiles owned by this rank, stacked in tile order (T_local, nvar, [V,] Ht, Ht, points) -> the global array
    (6, nvar, [V,] k Ht, k Ht, points) on rank 0 (on `local`'s device), None elsewhere.  Collective; a rank that owns
    no tile passes an empty stack (shape (0, ...))."""
    topo, per_rank, width = _layout(world_size, tiles_per_side)
    k = tiles_per_side
    if local.shape[0] != len(per_rank[rank]):
        raise ValueError(f"rank {rank} owns {len(per_rank[rank])} tile(s), got a stack of {local.shape[0]}")
    tile_shape = tuple(local.shape[1:])
    send = local.new_zeros((width,) + tile_shape)
    send[: local.shape[0]] = local
    if world_size > 1:
        recv = [torch.empty_like(send) for _ in range(world_size)] if rank == 0 else None
        dist.gather(send.contiguous(), recv, dst=0, group=group)
    else:
        recv = [send]
    if rank != 0:
        return None
    Ht = tile_shape[-2]
    if tile_shape[-3] != Ht:
        raise ValueError(f"tiles must be square in the horizontal, got {tile_shape}")
    out = local.new_empty((6,) + tile_shape[:-3] + (k * Ht, k * Ht, tile_shape[-1]))
    for r, tiles in enumerate(per_rank):
        for i, t in enumerate(tiles):
            out[_tile_slices(topo, t, Ht)] = recv[r][i]
    return out


def distribute_cube(global_state, rank: int = 0, world_size: int = 1, device="cpu", group=None,
                    tiles_per_side: int = 1, tile_shape=None, dtype=torch.float64) -> torch.Tensor:
    """Global (6, ...) array on rank 0 -> this rank's tiles stacked in tile order on `device`
    (process_topology.py:471-539).  Collective.  Ranks other than 0 pass None; the tile shape and dtype come from rank 0
    (the `tile_shape` / `dtype` arguments are accepted for compatibility and ignored over several ranks)."""
    topo, per_rank, width = _layout(world_size, tiles_per_side)
    k = tiles_per_side
    scatter = None
    # Rank 0 validates and tells everybody the outcome (and the tile shape / dtype) BEFORE the scatter, so that a bad
    # array raises on all ranks together instead of leaving the others waiting in the collective.
    err, g = None, None
    if rank == 0:
        g = global_state if isinstance(global_state, torch.Tensor) else torch.from_numpy(numpy.ascontiguousarray(global_state))
        if g.shape[0] != 6 or g.shape[-2] != g.shape[-3]:
            err = f"This is not a cube with square panels: {tuple(g.shape)}"
        elif g.shape[-2] % k:
            ok = [6 * i * i for i in range(1, g.shape[-2] + 1) if g.shape[-2] % i == 0]
            err = (f"shape {tuple(g.shape)} cannot be cut into {k} x {k} tiles per panel; "
                   f"acceptable numbers of tiles are {ok}")
        else:
            Ht = g.shape[-2] // k
            tile_shape, dtype = tuple(g.shape[1:-3]) + (Ht, Ht, g.shape[-1]), g.dtype
    if world_size > 1:
        head = [err, tuple(tile_shape) if tile_shape is not None else None, dtype]
        dist.broadcast_object_list(head, src=0, group=group)
        err, tile_shape, dtype = head
    if err is not None:
        raise ValueError(err)
    if rank == 0:
        Ht = tile_shape[-2]
        g = g.to(device)
        scatter = []
        for tiles in per_rank:   # one padded buffer per destination rank (this rank's own is the result)
            buf = g.new_zeros((width,) + tuple(tile_shape))
            for i, t in enumerate(tiles):
                buf[i] = g[_tile_slices(topo, t, Ht)]
            scatter.append(buf)
        del g
    if world_size == 1:
        return scatter[0][: len(per_rank[0])].contiguous()
    mine = torch.empty((width,) + tuple(tile_shape), dtype=dtype, device=device)
    dist.scatter(mine, scatter, src=0, group=group)
    return mine[: len(per_rank[rank])].contiguous()
