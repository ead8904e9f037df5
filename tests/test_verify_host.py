"""CPU: the verifier and the careful LP engine of the HIP library (polytope_amd/csrc/plp_verify.hpp: certify, basis_from_x,
careful_solve -- what stands behind every answer of plp_lp_solve_batch / plp_cheby_batch / plp_bbox_batch since round 6)
compiled for the HOST and held against the oracle (its certificate with binary128 residuals, its binary128 engine) on the
soak families, `dup` (rows 1e-16 .. 1e-5 rad apart: where the dictionary engines' answers used to be off by 1e-7 .. 1e-5)
first of all.  The header is the same source the device compiles."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL  # noqa: E402

dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.fixture(scope="module")
def vh(oracle, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("verify") / "libverify_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                           os.path.join(ROOT, "tests", "cabi", "verify_host.cpp")])
    L = C.CDLL(out)
    L.plpv_certify_from_x.argtypes = [C.c_int] * 4 + [dp] * 6 + [ip]
    L.plpv_certify_basis.argtypes = [C.c_int] * 4 + [dp] * 3 + [C.c_int, ip, dp, dp, dp]
    L.plpv_careful.argtypes = [C.c_int] * 4 + [dp] * 5 + [ip]
    L.plpv_solve_any_check.argtypes = [C.c_int, dp, dp]
    OL = oracle.lib()
    OL.plpo_lp_solve_raw.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, ip, ip]
    OL.plpo_lp_solve_q.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, ip, ip]
    OL.plpo_lp_solve.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, ip]
    return L, OL


def _raw(OL, c, G, h):
    m, n = G.shape
    x = np.empty(n)
    f = C.c_double()
    bs = np.zeros(n + 2, np.int32)
    st = OL.plpo_lp_solve_raw(m, n, _p(c), _p(G), _p(h), _p(x), C.byref(f), None, _p(bs, C.c_int))
    return st, x, f.value, bs


def _oracle(OL, c, G, h):
    m, n = G.shape
    x = np.empty(n)
    f = C.c_double()
    st = OL.plpo_lp_solve(m, n, _p(c), _p(G), _p(h), _p(x), C.byref(f), None)
    return st, x, f.value


def _careful(L, c, G, h):
    m, n = G.shape
    x = np.empty(n)
    f = C.c_double()
    it = C.c_int()
    st = L.plpv_careful(0, m, n, 0, _p(c), _p(G), _p(h), _p(x), C.byref(f), C.byref(it))
    return st, x, f.value, it.value


def _box_lps(A, b):
    d = A.shape[1]
    for i in range(d):
        for sgn in (1.0, -1.0):
            c = np.zeros(d)
            c[i] = sgn
            yield c


def _ext(A, b, x):
    return max(1.0, float(np.max(np.abs(x)))) if x is not None and np.all(np.isfinite(x)) else 1.0


@pytest.mark.parametrize("fam", ["random", "ragged", "unbounded", "scaled", "flat", "lattice", "dup"])
def test_certificate_and_careful_engine_against_the_oracle(vh, fam):
    """Every box LP of 12 shapes x 12 polytopes per family, three ways: (i) the raw double engine's basis through the
    library's certificate == the oracle's verdict on the same basis (binary128 residuals), polished x equal to 1e-12 of its
    extent; (ii) the basis read off the raw engine's x: when certified, the same value; (iii) the careful double-double
    engine from scratch == the oracle's answer (certificate or binary128): status exact, value 1e-9 of the extent."""
    L, OL = vh
    rng = np.random.default_rng({"random": 1, "ragged": 2, "unbounded": 3, "scaled": 4, "flat": 5, "lattice": 6, "dup": 7}[fam])
    n_lp = n_cert = n_ctry = n_xcert = n_xtry = 0
    for trial in range(12):
        d = int(rng.choice([2, 3, 4, 5, 6, 8, 10, 13, 16]))
        m = int(rng.integers(d + 1, 65))
        A, b, mr = SL.make(rng, 12, m, d, fam)
        for k in range(12):
            Ak, bk = np.ascontiguousarray(A[k, :mr[k]]), np.ascontiguousarray(b[k, :mr[k]])
            for c in _box_lps(Ak, bk):
                st, x, f, bs = _raw(OL, c, Ak, bk)
                so, xo, fo = _oracle(OL, c, Ak, bk)
                n_lp += 1
                if st in (0, 3):
                    n_ctry += 1
                    xc, fc = np.empty(d), C.c_double()
                    ok = L.plpv_certify_basis(0, Ak.shape[0], d, 0, _p(c), _p(Ak), _p(bk), st, _p(bs, C.c_int), None, _p(xc), C.byref(fc))
                    if ok:
                        n_cert += 1
                        # a certified answer is the oracle's answer (which certified the same basis, or took binary128)
                        if st == 0 and so == 0:
                            assert abs(fc.value - fo) <= 1e-9 * _ext(Ak, bk, xo), (fam, trial, k, c, fc.value, fo)
                        elif st == 3:
                            assert so == 3, (fam, trial, k, c, so)
                    if st == 0 and ok:
                        n_xtry += 1
                        xx, fx = np.empty(d), C.c_double()
                        okx = L.plpv_certify_from_x(0, Ak.shape[0], d, 0, _p(c), _p(Ak), _p(bk), _p(x), _p(xx), C.byref(fx), None)
                        if okx == 1:
                            n_xcert += 1
                            if so == 0:
                                assert abs(fx.value - fo) <= 1e-9 * _ext(Ak, bk, xo), (fam, trial, k, c, fx.value, fo)
                # the careful engine on a part of the LPs (it is the slow path), always where the oracle took binary128
                if fam == "dup" or (n_lp % 7) == 0:
                    sc, xk, fk, it = _careful(L, c, Ak, bk)
                    assert sc == so, (fam, trial, k, c, sc, so, fk, fo)
                    if so == 0:
                        assert abs(fk - fo) <= 1e-9 * _ext(Ak, bk, xo), (fam, trial, k, c, fk, fo)
    # the certificate passes everything the double engine gets right: all of it off `dup`
    if fam != "dup":
        assert n_cert == n_ctry, (fam, n_cert, n_ctry)
    # bases read off x: degenerate vertices may go to the careful engine
    print(fam, "LPs", n_lp, "certified from the basis", n_cert, "; optimal ones read off x:", n_xcert, "of", n_xtry)
    assert n_xcert >= (0.5 if fam in ("lattice", "dup") else 0.97) * n_xtry


def test_side_by_side_solves_give_the_serial_ones_bit_for_bit(vh):
    """The device solves M z = rhs and M' y = -c on two lanes with ONE routine whose instruction stream does not depend on which
    of the two it is (`lu_solve_any`); its results must be those of the serial `lu_solve` / `lu_solve_t` in every bit."""
    L, _ = vh
    rng = np.random.default_rng(11)
    for n in range(1, 18):
        for rep in range(20):
            M = rng.standard_normal((n, n))
            if rep % 3 == 0 and n > 1:
                M[rng.integers(n)] = np.eye(n)[rng.integers(n)]      # a free variable's row
            r = rng.standard_normal(n)
            assert L.plpv_solve_any_check(n, _p(np.ascontiguousarray(M)), _p(r)) in (0, -1), (n, rep)
