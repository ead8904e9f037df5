#!/usr/bin/env python3
"""Builder-container check of the drop-in claim: the REFERENCE's own test file (/root/reference/tests/polytope_test.py,
read in place, never copied) run against this package under the reference's module names -- `polytope`,
`polytope.polytope`, `polytope.solvers`, `polytope.quickhull`, `polytope.prop2partition` all resolve to polytope_amd.
Every test function and every `*_test` method of its classes (pytest does not collect the latter: SURVEY section 4) is
called.  Expected to fail: what is out of scope (grid_region, enumerate_integral_points, rotation) and gurobi.

    python scripts/run_reference_tests.py [backend]        (default backend: scipy; 'hip' needs a GPU AND the reference)
"""
import importlib.util
import inspect
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/tests/polytope_test.py"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
if not os.path.exists(REF):
    raise SystemExit("the reference checkout is not here (this check runs in the build container only)")
import polytope_amd  # noqa: E402

for name, mod in {"polytope": polytope_amd, "polytope.polytope": polytope_amd.polytope, "polytope.solvers": polytope_amd.solvers,
                  "polytope.quickhull": polytope_amd.quickhull, "polytope.prop2partition": polytope_amd.prop2partition}.items():
    sys.modules[name] = mod
if len(sys.argv) > 1:
    polytope_amd.solvers.default_solver = sys.argv[1]
spec = importlib.util.spec_from_file_location("ref_tests", REF)
T = importlib.util.module_from_spec(spec)
spec.loader.exec_module(T)
OUT_OF_SCOPE = ("rotation", "grid_region", "enumerate_integral_points", "gurobi")
ran, ok, expected, bad = 0, 0, [], []


def run(name, fn):
    global ran, ok
    ran += 1
    try:
        fn()
        ok += 1
    except Exception as e:  # noqa: BLE001
        (expected if any(k in name for k in OUT_OF_SCOPE) else bad).append((name, repr(e)[:160]))


for name, obj in list(vars(T).items()):
    if inspect.isclass(obj) and obj.__module__ == "ref_tests":
        inst = obj()
        for mn, meth in inspect.getmembers(inst, predicate=inspect.ismethod):
            if mn.endswith("_test") or mn.startswith("test"):
                if hasattr(inst, "setUp"):
                    inst.setUp()
                run(name + "." + mn, meth)
    elif inspect.isfunction(obj) and obj.__module__ == "ref_tests" and (name.endswith("_test") or name.startswith("test_")):
        if not inspect.signature(obj).parameters:
            run(name, obj)
print("reference tests run against polytope_amd (backend %s): %d, passed %d" % (polytope_amd.solvers.default_solver, ran, ok))
for n, e in expected:
    print("  out of scope:", n, "--", e)
for n, e in bad:
    print("  FAILED:", n, "--", e)
sys.exit(1 if bad else 0)
