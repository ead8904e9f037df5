"""LP solver selection -- drop-in for the reference's plug-in surface
(polytope/solvers.py:39-106, :149-158, :200-207).

Same contract as the reference:
  * `lpsolve(c, G, h, solver=None) -> dict(status=int, x=1-D float64 array or None, fun=float or None)`
    for  min c'x  s.t.  G x <= h,  x free;  status codes are scipy.optimize.linprog's.
  * `installed_solvers` (set of names) and `default_solver` (read at call time) are module
    globals; assign `solvers.default_solver = '...'` to switch backend.
  * unknown name -> Exception, known-but-absent -> RuntimeError, LP failure is NOT an
    exception (status != 0 with x = fun = None).

Backends here:
  'hip'    the MI355X engine (hand-written HIP kernels behind the C ABI of include/plp.h).
           Installed iff libplp_hip.so loads and a gfx950 device is visible.  It is OPT-IN
           (`solvers.default_solver = 'hip'` or `solver='hip'`), never the module default, and
           it never falls back to a CPU solver: without the library or the device, `lpsolve`
           raises RuntimeError exactly like a missing GLPK does in the reference
           (polytope/solvers.py:200-207).
  'scipy'  scipy.optimize.linprog called with the reference's argument convention
           (polytope/solvers.py:152-154).  The module default (the reference's rule picks glpk
           when cvxopt is importable, else scipy; this package ships no glpk binding, so the
           rule always lands on scipy); also what the CPU baseline in bench.py times.
  'glpk', 'mosek', 'gurobi' are recognised names (RuntimeError when absent) but not provided.
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)

installed_solvers = set()
try:
    from scipy import optimize as _optimize
    installed_solvers.add("scipy")
except ImportError:  # pragma: no cover
    _optimize = None

_KNOWN = ("hip", "scipy", "glpk", "mosek", "gurobi")
# The reference's rule (solvers.py:66-73) is "glpk if installed, else scipy".  This package provides no glpk
# backend ('glpk' is a recognised name that raises RuntimeError), so the rule always yields 'scipy' here.
# 'hip' is never made the default behind the user's back -- select it like any other backend:
#     from polytope_amd import solvers;  solvers.default_solver = 'hip'
# Selecting it on a machine without the library or a gfx950 device raises RuntimeError (no CPU fallback).
default_solver = "scipy"


def _probe_hip():
    from . import _lib
    if _lib.available():
        installed_solvers.add("hip")
    else:
        logger.info("polytope_amd: no gfx950 device / libplp_hip.so: solver 'hip' is not installed")


_probe_hip()


def lpsolve(c, G, h, solver=None):
    """Solve  min c'x  s.t.  G x <= h  (x free) with the chosen or the default backend.

    @param solver: one of 'hip', 'scipy' ('glpk', 'mosek', 'gurobi' are not provided);
        None means the module global `default_solver`, looked up at call time.
    @return: dict(status=int, x=ndarray (n,) or None, fun=float or None), status as in
        scipy.optimize.linprog (0 optimal, 1 iteration limit, 2 infeasible, 3 unbounded, 4 numerical)
    """
    name = default_solver if solver is None else solver
    if name == "hip":
        return _lp_hip(c, G, h)
    if name == "scipy":
        return _lp_scipy(c, G, h)
    if name in _KNOWN:
        _require(name)
    raise Exception('unknown LP solver "{s}".'.format(s=name))


def _require(name):
    if name not in installed_solvers:
        raise RuntimeError("solver {s} not in installed solvers: {have}".format(s=name, have=installed_solvers))


def _lp_scipy(c, G, h):
    _require("scipy")
    sol = _optimize.linprog(c, G, np.transpose(h), None, None, bounds=(None, None))
    return dict(status=sol.status, x=sol.x, fun=sol.fun)


# the reference's private names for the same things (its own tests call them: tests/polytope_test.py:545, :565-575)
_solve_lp_using_scipy = _lp_scipy
_assert_have_solver = _require


def _lp_hip(c, G, h):
    """One LP through the batched HIP kernel (a batch of one)."""
    _require("hip")
    from .batch import lpsolve_batch
    c = np.ascontiguousarray(c, dtype=np.float64).ravel()
    n = c.size
    G = np.ascontiguousarray(G, dtype=np.float64).reshape(-1, n)
    h = np.ascontiguousarray(h, dtype=np.float64).ravel()
    if G.shape[0] != h.size:
        raise ValueError("G and h have inconsistent shapes: %s, %s" % (G.shape, h.shape))
    res = lpsolve_batch(c[None], G[None], h[None])
    status = int(res["status"][0])
    if status != 0:
        return dict(status=status, x=None, fun=None)
    return dict(status=0, x=res["x"][0].copy(), fun=float(res["fun"][0]))


def lpsolve_many(cs, Gs, hs, solver=None):
    """Solve a list of LPs that share the column count: with 'hip' they go down as ONE batch
    (rows padded to the longest), with any other backend one lpsolve() call each.

    -> list of dicts as returned by lpsolve.
    """
    name = default_solver if solver is None else solver
    if name != "hip":
        return [lpsolve(c, G, h, solver=name) for c, G, h in zip(cs, Gs, hs)]
    _require("hip")
    from .batch import lpsolve_batch
    B = len(cs)
    if B == 0:
        return []
    n = np.asarray(cs[0]).size
    ms = np.array([np.asarray(h_).size for h_ in hs], dtype=np.int32)
    m_max = int(ms.max()) if B else 0
    c = np.zeros((B, n))
    G = np.zeros((B, max(m_max, 1), n))
    h = np.zeros((B, max(m_max, 1)))
    for k in range(B):
        c[k] = np.asarray(cs[k], dtype=np.float64).ravel()
        G[k, :ms[k]] = np.asarray(Gs[k], dtype=np.float64).reshape(ms[k], n)
        h[k, :ms[k]] = np.asarray(hs[k], dtype=np.float64).ravel()
    res = lpsolve_batch(c, G, h, m=ms)
    out = []
    for k in range(B):
        st = int(res["status"][k])
        if st == 0:
            out.append(dict(status=0, x=res["x"][k].copy(), fun=float(res["fun"][k])))
        else:
            out.append(dict(status=st, x=None, fun=None))
    return out
