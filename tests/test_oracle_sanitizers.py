"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no
sanitizer or race-detection story; the checker of this repository gets one).

`make -C oracle asan` builds liboracle_asan.so from the same sources; the known-answer tests and the golden-fixture
tests then run against it in a child interpreter (libasan has to be the first library of the process, so it is
LD_PRELOADed). Any report aborts the child (`halt_on_error`, `-fno-sanitize-recover`) and fails this test.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_is_clean_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes in this toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    env = dict(os.environ)
    env.update({
        "SF_ORACLE_LIB": lib,
        "LD_PRELOAD": asan + ":" + ubsan,
        # python itself leaks by design; everything else is fatal
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
    })
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_golden.py")],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in out
