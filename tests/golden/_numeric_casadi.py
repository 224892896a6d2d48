"""Numeric stand-in for the `casadi` names the reference model-building code touches.

TEST INFRASTRUCTURE, used only by tests/golden/make_golden.py inside the build
container (where /root/reference exists and the real casadi does not).  It lets
the reference's own `obca.obca_mpc4/6/8` run UNMODIFIED with every decision
variable bound to a concrete number, so that `opti.minimize(expr)` receives the
objective VALUE and every `opti.subject_to(...)` receives (lower, value, upper)
of that constraint at the chosen point.  It contains no solver: `solve()` raises,
which drives the reference into its `except:` branch (src/obca.py:1062-1065).

What this pins: the NLP *definition* (objective and every constraint function,
including the reference's indexing quirks).  What it cannot pin: IPOPT's iterate
path.  See DESIGN.md "Oracle and parity".
"""
import math
import numpy as np

pi = math.pi


class V:
    """A float that records nothing but survives numpy/python arithmetic and
    turns comparisons into constraint records."""
    __slots__ = ("v",)
    __array_priority__ = 1000

    def __init__(self, v):
        self.v = float(v.v if isinstance(v, V) else v)

    # --- numpy interop: np.cos(V), np.float64 - V, ... -------------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        if method != "__call__":
            return NotImplemented
        a = [i.v if isinstance(i, V) else float(i) for i in inputs]
        table = {
            np.cos: lambda x: math.cos(x), np.sin: lambda x: math.sin(x),
            np.add: lambda x, y: x + y, np.subtract: lambda x, y: x - y,
            np.multiply: lambda x, y: x * y, np.true_divide: lambda x, y: x / y,
            np.negative: lambda x: -x, np.power: lambda x, y: x ** y,
        }
        if ufunc in table:
            return V(table[ufunc](*a))
        cmp = {np.equal: "__eq__", np.less_equal: "__ge__", np.greater_equal: "__le__",
               np.less: "__gt__", np.greater: "__lt__"}
        if ufunc in cmp:  # scalar OP V  ->  reflected on V
            return getattr(inputs[1], cmp[ufunc])(inputs[0])
        return NotImplemented

    @staticmethod
    def _f(o):
        return o.v if isinstance(o, V) else float(o)

    def __add__(s, o): return V(s.v + V._f(o))
    __radd__ = __add__
    def __sub__(s, o): return V(s.v - V._f(o))
    def __rsub__(s, o): return V(V._f(o) - s.v)
    def __mul__(s, o): return V(s.v * V._f(o))
    __rmul__ = __mul__
    def __truediv__(s, o): return V(s.v / V._f(o))
    def __rtruediv__(s, o): return V(V._f(o) / s.v)
    def __pow__(s, o): return V(s.v ** V._f(o))
    def __neg__(s): return V(-s.v)
    def __float__(s): return s.v
    # comparisons -> constraint records (lower, value, upper)
    def __eq__(s, o): return Con("eq", [s.v - V._f(o)], [0.0], [0.0]) if isinstance(o, V) \
        else Con("eq", [s.v], [V._f(o)], [V._f(o)])
    def __ge__(s, o): return _ineq(s, o, lower=True)
    def __gt__(s, o): return _ineq(s, o, lower=True)
    def __le__(s, o): return _ineq(s, o, lower=False)
    def __lt__(s, o): return _ineq(s, o, lower=False)
    __hash__ = None


def _ineq(s, o, lower):
    # CasADi Opti canonical form: a parametric (constant) side becomes the bound,
    # otherwise g = a - b with bound 0.
    inf = float("inf")
    if isinstance(o, V):
        val = s.v - o.v
        return Con("ineq", [val], [0.0 if lower else -inf], [inf if lower else 0.0])
    c = float(o)
    return Con("ineq", [s.v], [c if lower else -inf], [inf if lower else c])


class Con:
    def __init__(self, kind, val, lb, ub):
        self.kind, self.val, self.lb, self.ub = kind, list(val), list(lb), list(ub)


class M:
    """Dense column-major matrix of V with the MX indexing the reference uses."""

    def __init__(self, a):
        self.a = np.empty(np.shape(a), dtype=object)
        it = np.nditer(np.asarray(a, dtype=object), flags=["multi_index", "refs_ok"])
        for x in it:
            self.a[it.multi_index] = V(x.item())

    @property
    def shape(self):
        return self.a.shape

    def _wrap(self, r):
        if isinstance(r, np.ndarray):
            m = M.__new__(M)
            m.a = r if r.ndim == 2 else r.reshape(-1, 1)
            return m
        return r

    def __getitem__(self, k):
        if isinstance(k, tuple):
            r = self.a[k]
            if isinstance(r, np.ndarray) and r.ndim == 1:
                # keep orientation: row slice stays a row, column slice a column
                r = r.reshape(1, -1) if isinstance(k[0], (int, np.integer)) else r.reshape(-1, 1)
            return self._wrap(r)
        if isinstance(k, (int, np.integer)):  # CasADi linear (column-major) index
            rows = self.a.shape[0]
            return self.a[k % rows, k // rows]
        raise TypeError(k)

    def __setitem__(self, k, val):
        if isinstance(k, (int, np.integer)):
            rows = self.a.shape[0]
            k = (k % rows, k // rows)
        self.a[k] = V(val)

    def _flat(self):
        return [e.v for e in self.a.flatten(order="F")]

    def _cmp(self, o, kind, lower=None):
        mine = self._flat()
        if isinstance(o, M):
            other = o._flat()
            param = False
        else:
            other = [float(x.v if isinstance(x, V) else x) for x in np.ravel(np.asarray(o, dtype=object))]
            if len(other) == 1:
                other = other * len(mine)
            param = True
        inf = float("inf")
        assert len(other) == len(mine)
        if param:
            if kind == "eq":
                return Con("eq", mine, other, other)
            return Con("ineq", mine, other if lower else [-inf] * len(mine), [inf] * len(mine) if lower else other)
        d = [a - b for a, b in zip(mine, other)]
        z = [0.0] * len(d)
        if kind == "eq":
            return Con("eq", d, z, z)
        return Con("ineq", d, z if lower else [-inf] * len(d), [inf] * len(d) if lower else z)

    def __eq__(s, o): return s._cmp(o, "eq")
    def __ge__(s, o): return s._cmp(o, "ineq", lower=True)
    def __gt__(s, o): return s._cmp(o, "ineq", lower=True)
    def __le__(s, o): return s._cmp(o, "ineq", lower=False)
    def __lt__(s, o): return s._cmp(o, "ineq", lower=False)
    __hash__ = None


def MX(r, c=1):
    return M(np.zeros((r, c)))


def cos(x): return V(math.cos(V._f(x)))
def sin(x): return V(math.sin(V._f(x)))


class _NoSolver(Exception):
    pass


class _Debug:
    def __init__(self, opti): self.o = opti
    def value(self, e): return self.o._value(e)


class Opti:
    """Records objective value and constraint (lb, value, ub) triples in call order."""
    feed = None  # set by the generator: list of ndarrays consumed by successive variable() calls

    def __init__(self):
        self._feed = list(Opti.feed)
        self.objective = None
        self.cons = []
        self.debug = _Debug(self)
        Opti.last = self

    def variable(self, r=1, c=1):
        a = np.asarray(self._feed.pop(0), dtype=float).reshape(r, c)
        return M(a)

    def set_initial(self, var, val):
        pass

    def minimize(self, e):
        self.objective = V._f(e)

    def bounded(self, lb, e, ub):
        n = len(e._flat()) if isinstance(e, M) else 1
        val = e._flat() if isinstance(e, M) else [V._f(e)]
        return Con("ineq", val, [float(V._f(lb))] * n, [float(V._f(ub))] * n)

    def subject_to(self, c):
        assert isinstance(c, Con), type(c)
        self.cons.append(c)

    def solver(self, *a, **k):
        pass

    def solve(self):
        raise _NoSolver()

    def _value(self, e):
        if isinstance(e, M):
            return np.array([[x.v for x in row] for row in e.a])
        return V._f(e)
