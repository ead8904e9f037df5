"""The C ABI used from plain C (tests/cabi/client.c): compiled with gcc against include/plp.h and linked to the
in-tree libplp_hip.so -- no Python objects, no torch on the other side of the boundary."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "client.c")
LIBDIR = os.path.join(ROOT, "polytope_amd")


def _build(tmp_path):
    exe = str(tmp_path / "cabi_client")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", LIBDIR, "-lplp_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_client_compiles_and_links(tmp_path):
    """CPU: the header is valid C99 and every symbol the client uses resolves; without a device the client
    says so and exits with code 2 (no CPU fallback)."""
    if not os.path.exists(os.path.join(LIBDIR, "libplp_hip.so")):
        pytest.skip("libplp_hip.so not built")
    exe = _build(tmp_path)
    sys.path.insert(0, ROOT)
    from polytope_amd import _lib
    if not _lib.available():
        p = subprocess.run([exe], capture_output=True, text=True)
        assert p.returncode == 2 and "NODEVICE" in p.stdout


@pytest.mark.gpu
def test_c_client_results(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    out = p.stdout.splitlines()
    assert p.returncode == 0, p.stdout + p.stderr
    assert "lp1 status 0 x -1 fun -1" in out
    assert "lp2 status 0 x -1 -1 | status 3" in out
    assert any(l.startswith("reduce keep 0x1d flags 4 nlp ") for l in out), out   # rows 0,2,3,4 kept, minrep
    assert "bbox status 0 lb 2.000000000 -1.000000000 ub 5.000000000 3.000000000 | status 1" in out, out
    assert "contains 1 0 1 0" in out      # the corner (1,1): A x - b = 0 < abs_tol counts as inside
    assert "quickhull facets 12 vertices 8" in out, out      # the cube's 6 faces as 12 triangles, interior points dropped
    # one piece: the square's 4 rows + the strip's first new row negated (table row 4 + 4 = 8: x <= 0.5), reduced (kind 1)
    assert "region_diff pieces 1: kind 1 rows 0 1 2 3 8" in out, out
    assert "envelope rc 2 (PLP_EUNSUPPORTED)" in out and out[-1] == "done"
