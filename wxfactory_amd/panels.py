"""Cubed-sphere tile graph (host logic of the halo exchange).

Mirrors reference wx_factory/process_topology.py:69-256 for 6 k^2 tiles (k x k per panel, k = 1: one tile
per panel): tile numbering (:30-47, 89-94), the panel neighbour table (:105-113), the tile that sits
across each panel edge (`edge_coords`, :118-125), which edges flip (:127-134), and the delivery rule of
MPI_Neighbor_alltoall on the dist-graph communicator (:259-261): what tile t sends through edge e
arrives in the recv slot e' of t' = neighbor(t, e) where neighbor(t', e') == t.  Interior tile edges
carry no rotation and no flip (:219-228).  Edge order everywhere: SOUTH, NORTH, WEST, EAST = 0, 1, 2, 3.
"""
from functools import lru_cache
from typing import List, Tuple

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

FLIP = (
    (False, False, False, False),
    (True, False, False, False),
    (True, True, False, False),
    (False, True, False, False),
    (False, True, True, False),
    (True, False, False, True),
)


def landing_edge(panel: int, edge: int) -> int:
    return NEIGHBOR[NEIGHBOR[panel][edge]].index(panel)


class CubeTopology:
    """6 k^2 tiles; tile id = panel*k^2 + row*k + col (row along x2, col along x1)."""

    def __init__(self, k: int = 1):
        if k < 1:
            raise ValueError("k >= 1")
        self.k = k
        self.ntiles = 6 * k * k

    def locate(self, t: int) -> Tuple[int, int, int]:
        k2 = self.k * self.k
        return t // k2, (t % k2) // self.k, t % self.k

    def tile(self, panel: int, row: int, col: int) -> int:
        k = self.k
        return panel * k * k + (row % k) * k + (col % k)

    def on_panel_edge(self, t: int) -> List[bool]:
        _, r, c = self.locate(t)
        k = self.k
        return [r == 0, r == k - 1, c == 0, c == k - 1]

    def _across(self, panel: int, r: int, c: int, e: int) -> Tuple[int, int]:
        """(row, col) on the neighbouring panel of the tile across panel edge e (process_topology.py:118-125);
        negative indices count from the far side."""
        table = (
            ((-1, c), (0, c), (r, -1), (r, 0)),
            ((-c - 1, -1), (c, -1), (r, -1), (r, 0)),
            ((0, -c - 1), (-1, -c - 1), (r, -1), (r, 0)),
            ((c, 0), (-c - 1, 0), (r, -1), (r, 0)),
            ((-1, c), (-1, -c - 1), (-1, -r - 1), (-1, r)),
            ((0, -c - 1), (0, c), (0, r), (0, -r - 1)),
        )
        return table[panel][e]

    @lru_cache(maxsize=None)
    def neighbor(self, t: int, e: int) -> int:
        p, r, c = self.locate(t)
        if self.on_panel_edge(t)[e]:
            rr, cc = self._across(p, r, c, e)
            return self.tile(NEIGHBOR[p][e], rr, cc)
        dr, dc = ((-1, 0), (1, 0), (0, -1), (0, 1))[e]
        return self.tile(p, r + dr, c + dc)

    @lru_cache(maxsize=None)
    def landing(self, t: int, e: int) -> int:
        """Edge of neighbor(t, e) through which it sees t."""
        q = self.neighbor(t, e)
        hits = [e2 for e2 in range(4) if self.neighbor(q, e2) == t]
        if len(hits) != 1:
            raise RuntimeError(f"ambiguous delivery between tiles {t} and {q}")  # cannot happen for k >= 1
        return hits[0]

    def flips(self, t: int) -> List[bool]:
        p = self.locate(t)[0]
        on = self.on_panel_edge(t)
        return [bool(FLIP[p][e]) and on[e] for e in range(4)]


def owner_of_tiles(world_size: int, ntiles: int = 6) -> List[int]:
    """Rank that owns each tile: contiguous, equal-sized runs over the first min(world, ntiles) ranks when
    that divides evenly, round-robin otherwise.  Ranks beyond the tile count own nothing."""
    active = min(world_size, ntiles)
    if ntiles % active == 0:
        per = ntiles // active
        return [t // per for t in range(ntiles)]
    return [t % active for t in range(ntiles)]


def tiles_of_rank(rank: int, world_size: int, ntiles: int = 6) -> List[int]:
    return [t for t, r in enumerate(owner_of_tiles(world_size, ntiles)) if r == rank]


def tiles_per_side_for(world_size: int) -> int:
    """Smallest k whose 6 k^2 tiles spread evenly over the ranks (1 for 1, 2, 3, 6 GPUs; 2 for 4, 8, 12, 24)."""
    for k in (1, 2, 3, 4, 5, 6):
        if (6 * k * k) % world_size == 0:
            return k
    return 1


# one tile per panel (k = 1): the names the rest of the package grew up with
def owner_of_panels(world_size: int) -> List[int]:
    return owner_of_tiles(world_size, 6)


def panels_of_rank(rank: int, world_size: int) -> List[int]:
    return tiles_of_rank(rank, world_size, 6)
