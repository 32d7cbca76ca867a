"""Cubed-sphere panel graph, edge flips and vector rotations (CPU oracle).

TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (wxfactory_amd/) never does.

Restates, for the 6-rank (one panel per rank) decomposition:
  * the neighbour table            reference wx_factory/process_topology.py:105-113
  * the per-edge flip flags        reference wx_factory/process_topology.py:126-134
  * the contravariant rotations    reference wx_factory/process_topology.py:137-175
  * MPI neighbour-alltoall routing reference wx_factory/process_topology.py:259-261, 318, 384
as data tables (coefficients) instead of lambdas.
"""
import numpy

SOUTH, NORTH, WEST, EAST = 0, 1, 2, 3

# NEIGHBOR[p][edge] = panel on the other side of `edge` of panel p (order S, N, W, E)
NEIGHBOR = (
    (5, 4, 3, 1),
    (5, 4, 0, 2),
    (5, 4, 1, 3),
    (5, 4, 2, 0),
    (0, 2, 3, 1),
    (2, 0, 3, 1),
)

# FLIP[p][edge]: reverse the along-edge ordering before sending
FLIP = (
    (False, False, False, False),
    (True, False, False, False),
    (True, True, False, False),
    (False, True, False, False),
    (False, True, True, False),
    (True, False, False, True),
)

# Rotation of the horizontal contravariant components before sending, with c = 2X/(1+X^2):
#   b1 = m[0]*a1 + m[1]*a2 + c*(m[2]*a1 + m[3]*a2)
#   b2 = m[4]*a1 + m[5]*a2 + c*(m[6]*a1 + m[7]*a2)
# ROT[p][edge] = (m0..m7)
_WE0 = (
    (1, 0, 0, 0, 0, 1, 1, 0),   # W: (a1, c a1 + a2)
    (1, 0, 0, 0, 0, 1, -1, 0),  # E: (a1, -c a1 + a2)
)
ROT = (
    (  # panel 0
        (1, 0, 0, 1, 0, 1, 0, 0),    # S: (a1 + c a2, a2)
        (1, 0, 0, -1, 0, 1, 0, 0),   # N: (a1 - c a2, a2)
    ) + _WE0,
    (  # panel 1
        (0, 1, 0, 0, -1, 0, 0, -1),  # S: (a2, -a1 - c a2)
        (0, -1, 0, 0, 1, 0, 0, -1),  # N: (-a2, a1 - c a2)
    ) + _WE0,
    (  # panel 2
        (-1, 0, 0, -1, 0, -1, 0, 0),  # S: (-a1 - c a2, -a2)
        (-1, 0, 0, 1, 0, -1, 0, 0),   # N: (-a1 + c a2, -a2)
    ) + _WE0,
    (  # panel 3
        (0, -1, 0, 0, 1, 0, 0, 1),   # S: (-a2, a1 + c a2)
        (0, 1, 0, 0, -1, 0, 0, 1),   # N: (a2, -a1 + c a2)
    ) + _WE0,
    (  # panel 4
        (1, 0, 0, 1, 0, 1, 0, 0),     # S: (a1 + c a2, a2)
        (-1, 0, 0, 1, 0, -1, 0, 0),   # N: (-a1 + c a2, -a2)
        (0, -1, -1, 0, 1, 0, 0, 0),   # W: (-c a1 - a2, a1)
        (0, 1, -1, 0, -1, 0, 0, 0),   # E: (-c a1 + a2, -a1)
    ),
    (  # panel 5
        (-1, 0, 0, -1, 0, -1, 0, 0),  # S: (-a1 - c a2, -a2)
        (1, 0, 0, -1, 0, 1, 0, 0),    # N: (a1 - c a2, a2)
        (0, 1, 1, 0, -1, 0, 0, 0),    # W: (c a1 + a2, -a1)
        (0, -1, 1, 0, 1, 0, 0, 0),    # E: (c a1 - a2, a1)
    ),
)


def landing_edge(panel: int, edge: int) -> int:
    """Edge of NEIGHBOR[panel][edge] on which data sent through `edge` of `panel` lands
    (MPI_Neighbor_alltoall on the dist-graph: block j of the sender where dest[j]==me)."""
    q = NEIGHBOR[panel][edge]
    return NEIGHBOR[q].index(panel)


def rotate(panel: int, edge: int, a1, a2, X):
    """convert_contra[edge](a1, a2, X) of `panel` (reference process_topology.py:137-175)."""
    m = ROT[panel][edge]
    c = 2.0 * X / (1.0 + X**2)
    b1 = m[0] * a1 + m[1] * a2 + c * (m[2] * a1 + m[3] * a2)
    b2 = m[4] * a1 + m[5] * a2 + c * (m[6] * a1 + m[7] * a2)
    return b1, b2


def route(sends):
    """sends[p][edge] -> recvs[p][edge] for the 6-panel sphere (what Ineighbor_alltoall delivers)."""
    recvs = [[None] * 4 for _ in range(6)]
    for p in range(6):
        for e in range(4):
            q = NEIGHBOR[p][e]
            recvs[q][landing_edge(p, e)] = sends[p][e]
    return recvs
