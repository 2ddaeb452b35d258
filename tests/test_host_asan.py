"""AddressSanitizer + UBSan build of the host-only C++ of the library (descriptor lowering: exact
polynomial tables, parity-class folding, isotropic-group folding, validation) driven over the descriptor
zoo on the CPU box (SURVEY.md §5 "ASan host build"; GPU ASan is not available on the pool).  The
sanitized library is test infrastructure (`csrc/hosttest/`), never loaded by the product."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "linpde-gp_amd", "csrc")
LIB = os.path.join(CSRC, "hosttest", "liblpgp_hosttest_asan.so")


def test_lowering_under_address_sanitizer():
    gxx = shutil.which(os.environ.get("CXX", "g++"))
    if gxx is None:
        pytest.skip("no host C++ compiler")
    libasan = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    subprocess.run(["bash", os.path.join(CSRC, "build.sh"), "--host-asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_host_asan_worker.py"), LIB],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-4000:]
    assert "descriptors evaluated against the oracle" in res.stdout
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr
