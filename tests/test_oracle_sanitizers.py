"""Sanitizer leg: the C++ oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C oracle asan` builds oracle/_build/libcrowdstep_oracle_asan.so; the oracle's known-answer tests
(the reference's eight tests restated + the hand-derived KATs) and the Zanlungo cross-check then run in a
child python with libasan preloaded and CS_ORACLE_SANITIZED=1, which makes tests/oracle_sim.py load that
build.  Any report from either sanitizer aborts the child (halt_on_error, -fno-sanitize-recover).
GPU AddressSanitizer is not available on this pool: the HIP engine is covered by its parity tests only.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.sep in out and os.path.exists(out) else None


def test_oracle_known_answers_under_asan_and_ubsan():
    asan = _libasan()
    assert asan, "gcc's libasan.so not found"
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": asan,
        "CS_ORACLE_SANITIZED": "1",
        # python itself leaks by design; everything else is fatal
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
        "OMP_NUM_THREADS": "2",
    })
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_reference_kats.py"),
                        os.path.join(ROOT, "tests", "test_zanlungo_restatement.py"),
                        "-k", "not openmp and not 1000"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
