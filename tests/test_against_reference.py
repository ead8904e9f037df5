"""CPU, build container only: the host layer against THE REFERENCE ITSELF (imported in place from /root/reference, nothing of it is
copied; both sides on the scipy backend).  Skipped where the reference is not there (the GPU box).  The long forms are
scripts/soak_edges_cpu.py and scripts/soak_objects_cpu.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/polytope"), reason="the reference checkout is not on this machine")


def _run(script, *args, timeout=900):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + list(args), capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    return out.returncode, out.stdout


def test_edge_menu_against_the_reference():
    """empty / half-space / slab / flat / infeasible / cone / tiny / duplicated and zero rows, every pair of the menu through 19
    light operations at d = 1 (scripts/soak_edges_cpu.py 1 2 3: 5 558 operations): same pieces in the same order (rows 1e-9), same booleans and boxes, same exception classes."""
    rc, out = _run("soak_edges_cpu.py", "1")
    assert rc == 0 and "EDGE SOAK (cpu, scipy backend on both sides) OK" in out, out[-2000:]


def test_random_objects_against_the_reference():
    """random polytope triples in d = 1..4 through reduce / intersect / union / mldivide / envelope / is_convex / is_adjacent /
    is_subset / bounding_box / extreme (+ the Region forms and union(check_convex) at d <= 2): 6 trials."""
    rc, out = _run("soak_objects_cpu.py", "6", "3")
    assert rc == 0 and "OBJECT SOAK (cpu, scipy backend on both sides) OK" in out, out[-2000:]
