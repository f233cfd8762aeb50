"""The host-side block layer under AddressSanitizer + UndefinedBehaviorSanitizer (CPU box).

`make asan` builds comms_blocks.cpp, runner.cpp and fir_designer.cpp with -fsanitize=address,undefined into
libpcx_blocks_asan.so; the block and designer suites then run against it in a child interpreter with the sanitizer
runtimes preloaded.  Any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_block_layer_is_clean_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes with this gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pothoscomms_amd", "csrc"), "asan"])
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": asan + ":" + ubsan,
        "PCX_BLOCKS_LIBRARY": os.path.join(ROOT, "pothoscomms_amd", "libpcx_blocks_asan.so"),
        # leak checking would report the interpreter's own arenas; everything else halts on the first report
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=99",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=98",
    })
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_blocks_cpu.py"), os.path.join(ROOT, "tests", "test_designer_cpu.py"),
                        os.path.join(ROOT, "tests", "test_arith_cpu.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    report = r.stdout[-3000:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, report
    assert r.returncode == 0, report
    assert " passed" in r.stdout
