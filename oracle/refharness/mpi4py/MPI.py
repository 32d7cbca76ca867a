"""Threaded N-rank stand-in for ``mpi4py.MPI`` (golden-vector generation only).

TEST INFRASTRUCTURE.  This container has no MPI; the reference's cubed-sphere
path needs 6 ranks (one per panel).  N ranks are emulated by N Python threads of
one process with barrier-based collectives.  This is the stand-in specified in
SURVEY.md Appendix A.1 (written for the survey, contains no reference code); the
reference's own 13 ``ProcessTopologyTest`` cases pass under it.

Only ``oracle/refharness/gen_golden.py`` uses it, and only in the build
container (``/root/reference`` does not exist on the GPU box).
"""
import os
import threading

import numpy

DOUBLE = "d"
LONG = "l"
MAX = "max"
SUM = "sum"
COMM_NULL = None

_tls = threading.local()
_reg_lock = threading.Lock()
_registry = {}


def _set_rank(r):
    _tls.rank = r


def _world_rank():
    return getattr(_tls, "rank", 0)


class Request:
    def __init__(self, fn):
        self.fn = fn

    def Wait(self):
        self.fn()


class Comm:
    def __init__(self, members):
        self.members = list(members)
        self.barrier = threading.Barrier(len(self.members))
        self.slots = [None] * len(self.members)
        self.seq = [0] * len(self.members)

    @property
    def rank(self):
        return self.members.index(_world_rank())

    @property
    def size(self):
        return len(self.members)

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def _all(self, x):
        r = self.rank
        self.slots[r] = x
        self.barrier.wait()
        out = list(self.slots)
        self.barrier.wait()
        return out

    def bcast(self, x, root=0):
        return self._all(x)[root]

    def _reduce(self, vals, op):
        out = vals[0]
        if op == MAX:
            for v in vals[1:]:
                out = numpy.maximum(out, v)
            return out
        for v in vals[1:]:
            out = out + v
        return out

    def allreduce(self, x, op=SUM):
        return self._reduce(self._all(x), op)

    def Allreduce(self, s, r, op=SUM):
        sb = s[0] if isinstance(s, (list, tuple)) else s
        rb = r[0] if isinstance(r, (list, tuple)) else r
        res = self._reduce(self._all(numpy.array(sb, copy=True)), op)
        rb[...] = res

    def gather(self, x, root=0):
        a = self._all(x)
        return a if self.rank == root else None

    def scatter(self, x, root=0):
        a = self._all(x)
        return a[root][self.rank]

    def Barrier(self):
        self.barrier.wait()

    def barrier_(self):
        self.Barrier()

    def _shared(self, kind, key, factory):
        r = self.rank
        s = self.seq[r]
        self.seq[r] += 1
        k = (id(self), kind, s, key)
        with _reg_lock:
            if k not in _registry:
                _registry[k] = factory()
            return _registry[k]

    def Split(self, color=0, key=0):
        info = self._all((int(color), key, _world_rank()))
        mine = sorted([(k, w) for (c, k, w) in info if c == int(color)])
        members = [w for (_, w) in mine]
        return self._shared("split", int(color), lambda: Comm(members))

    def Create_dist_graph_adjacent(self, sources, destinations):
        g = self._shared("graph", 0, lambda: GraphComm(self.members))
        g.src[g.rank] = [self.members[s] for s in sources]
        g.dst[g.rank] = [self.members[d] for d in destinations]
        return g

    def Disconnect(self):
        pass


class GraphComm(Comm):
    def __init__(self, members):
        super().__init__(members)
        self.src = [None] * len(members)
        self.dst = [None] * len(members)

    def Ineighbor_alltoall(self, sendbuf, recvbuf):
        me = self.rank
        allsend = self._all(sendbuf)
        for i, s_world in enumerate(self.src[me]):
            s = self.members.index(s_world)
            j = self.dst[s].index(self.members[me])
            recvbuf[i] = allsend[s][j]
        self.barrier.wait()
        return Request(lambda: None)


COMM_WORLD = Comm(range(int(os.environ.get("FAKE_MPI_SIZE", "6"))))


def reset_world(n):
    """Replace COMM_WORLD by a fresh n-rank world (between independent runs)."""
    global COMM_WORLD
    with _reg_lock:
        _registry.clear()
    COMM_WORLD = Comm(range(n))
    return COMM_WORLD


def run_ranks(fn, n=None):
    n = n or COMM_WORLD.size
    results = [None] * n
    errors = [None] * n

    def tgt(r):
        _set_rank(r)
        try:
            results[r] = fn(r)
        except BaseException:
            import traceback

            errors[r] = traceback.format_exc()
            for c in [COMM_WORLD] + list(_registry.values()):
                try:
                    c.barrier.abort()
                except Exception:
                    pass

    ts = [threading.Thread(target=tgt, args=(r,)) for r in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return results, errors
