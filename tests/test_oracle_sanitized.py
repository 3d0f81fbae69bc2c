"""CPU: the oracle's C sources under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY 5 aux: the reference has no
race / memory checker; the build's checker for its own CPU code is this).  The golden-vector and oracle tests are re-run in
a child process against oracle/_build/libvnect_oracle_asan.so; any out-of-bounds access, use-after-free or undefined
arithmetic in oracle/*.c aborts the child.  CPU box only: GPU sanitizers are not available on the pool."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_clean_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "_build", "libvnect_oracle_asan.so")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    libubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    env = dict(os.environ, VNECT_ORACLE_SO=so, LD_PRELOAD=libasan + ":" + libubsan, VNECT_ORACLE_THREADS="8",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_golden.py"), os.path.join(ROOT, "tests", "test_oracle_post.py"),
                        os.path.join(ROOT, "tests", "test_oracle_net.py")],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail
