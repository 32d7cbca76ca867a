"""CPU oracle, "reference-style" flavour: the same phases as oracle/euler3d.py with the element operators applied as
the reference applies them - dense n^3 x n^3 Kronecker matrices and `@` (reference wx_factory/geometry/operators.py:157-183,
rhs/rhs_dfr.py:25-34, 50-71, 89-139) - instead of sum-factorised contractions.  12.6 MFLOP per element at n = 8 where
the factorised form needs 0.2: this is "what WxFactory would do on that host" in bench.py's cpu_baseline.

TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity status: PINNED with the class it derives from (tests/test_oracle_euler3d.py::test_dense_flavour_*).
"""
import numpy

from .euler3d import Euler3DOracle


class Euler3DOracleDense(Euler3DOracle):
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        n = self.n
        n2, n3 = n * n, n**3
        eye3, eye2 = numpy.eye(n3), numpy.eye(2 * n2)
        f = super()
        # row p of each matrix = response to the unit vector of node p, so that  a @ M  is the operator applied to a:
        # derivative_{x,y,z} = kron(I, I, D)^T, kron(I, D, I)^T, kron(D, I, I)^T, and likewise the other three families
        self.Md = [f.deriv(eye3, d) for d in range(3)]            # (n^3, n^3)
        self.Me = [f.extrap(eye3, d) for d in range(3)]           # (n^3, 2 n^2)
        self.Mc = [f.correct(eye2, d) for d in range(3)]          # (2 n^2, n^3)
        self.Mh = f.highfilter_k(eye3)                            # (n^3, n^3)

    def deriv(self, a, d):
        return a @ self.Md[d]

    def extrap(self, a, d):
        return a @ self.Me[d]

    def correct(self, f, d):
        return f @ self.Mc[d]

    def highfilter_k(self, a):
        return a @ self.Mh
