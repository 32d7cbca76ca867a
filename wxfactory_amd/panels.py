"""Cubed-sphere panel graph (host logic of the halo exchange).

Mirrors reference wx_factory/process_topology.py:105-113 (all_neighbors) for the one-tile-
per-panel decomposition, and the delivery rule of MPI_Neighbor_alltoall on the dist-graph
communicator (process_topology.py:259-261): what panel p sends through edge e arrives in
the recv slot e' of q = NEIGHBOR[p][e] where NEIGHBOR[q][e'] == p.
Edge order everywhere: SOUTH, NORTH, WEST, EAST = 0, 1, 2, 3.
"""
SOUTH, NORTH, WEST, EAST = 0, 1, 2, 3
EDGE_NAMES = ("s", "n", "w", "e")

NEIGHBOR = (
    (5, 4, 3, 1),
    (5, 4, 0, 2),
    (5, 4, 1, 3),
    (5, 4, 2, 0),
    (0, 2, 3, 1),
    (2, 0, 3, 1),
)


def landing_edge(panel: int, edge: int) -> int:
    return NEIGHBOR[NEIGHBOR[panel][edge]].index(panel)


def owner_of_panels(world_size: int):
    """Rank that owns each of the 6 panels: panel p -> rank p % min(world_size, 6).
    Ranks >= 6 own nothing (an 8-GPU node leaves two GPUs idle, as 6 MPI ranks would)."""
    active = min(world_size, 6)
    return [p % active for p in range(6)]


def panels_of_rank(rank: int, world_size: int):
    return [p for p, r in enumerate(owner_of_panels(world_size)) if r == rank]
