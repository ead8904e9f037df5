"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the native code that compiles for the host (SURVEY.md section 5:
sanitizers) -- the C oracle (oracle/plp_oracle.c, plp_oracle_q.c), the host builds of the lane engine (plp_lane_lp.hpp) and
of the verifier (plp_verify.hpp), and quickhull's main loop (plp_quickhull_host.hip: plain C++) over a CPU stand-in of its
device session.  One program (tests/cabi/sanitize_main.cpp), seeded inputs incl. rows a hair apart; it must exit 0 with no
sanitizer report."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_host_code_under_asan_and_ubsan(tmp_path):
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1", "-ffp-contract=off"]
    objs = []
    for src in ("oracle/plp_oracle.c", "oracle/plp_oracle_q.c"):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(["gcc", "-std=c11", "-c", os.path.join(ROOT, src), "-o", o] + san)
        objs.append(o)
    for src in ("tests/cabi/lane_lp_host.cpp", "tests/cabi/verify_host.cpp"):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(["g++", "-std=c++17", "-c", os.path.join(ROOT, src), "-o", o] + san)
        objs.append(o)
    exe = str(tmp_path / "sanitize_main")
    subprocess.check_call(["g++", "-std=c++17", "-x", "c++", os.path.join(ROOT, "tests", "cabi", "sanitize_main.cpp"), "-x", "none"] + objs +
                          ["-o", exe, "-lquadmath", "-lm", "-lpthread"] + san)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", PLP_QH_THREADS="2")
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert "sanitize_main: 0 failures" in p.stdout, p.stdout[-2000:]
