"""CPU sanitizer runs (GPU sanitizers are not available on this pool and not wanted):
  * the host-side C++ of the product (constant generator, host half of thompson_init, table-cache reader/writer) under
    AddressSanitizer + UBSan: `make -C kid_amd/csrc asan` (tests/native/host_asan_check.cpp);
  * the oracle's C restatement built with the same sanitizers (`make -C oracle asan`) and driven through a bounded
    part of this very test-suite in a child process with libasan preloaded.  The whole non-GPU suite under the
    sanitizers is a manual run (a few minutes; command below), its log is committed under profiles/."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    return subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True, check=True).stdout.strip()


def test_host_side_cxx_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "kid_amd", "csrc"), "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "host_asan_check ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def sanitizer_env():
    """Environment of a child python that loads the sanitizer build of the oracle.
    Full suite by hand:  env $(python -c 'import tests.test_sanitizers as t; print(t.env_line())') python -m pytest tests -m 'not gpu' -q"""
    env = dict(os.environ)
    env["LD_PRELOAD"] = _lib("libasan.so") + ":" + _lib("libubsan.so")
    env["ASAN_OPTIONS"] = "detect_leaks=0"                 # CPython itself is not leak-clean
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["KIDMP_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "_asan", "libthompson_oracle.so")
    return env


def env_line():
    e = sanitizer_env()
    return " ".join("%s=%s" % (k, e[k]) for k in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS", "KIDMP_ORACLE_LIB"))


def test_oracle_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "-s"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # unit functions, the 17-digit goldens (every branch of mp_thompson incl. the aerosol-aware ones) and the P32n build
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_units.py"), os.path.join(ROOT, "tests", "test_oracle_goldens.py"),
                        os.path.join(ROOT, "tests", "test_oracle_p32n.py")],
                       capture_output=True, text=True, timeout=1500, env=sanitizer_env(), cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    # the child really ran the sanitizer build
    chk = subprocess.run([sys.executable, "-c", "import oracle.oracle as o; print(o.lib()._name)"], capture_output=True, text=True,
                         env=sanitizer_env(), cwd=ROOT, timeout=300)
    assert "_asan" in chk.stdout, chk.stdout + chk.stderr
