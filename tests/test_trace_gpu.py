"""ROCTx ranges of the C ABI (include/pcx.h pcx_trace): under `rocprofv3 --marker-trace` every data-plane call made while
tracing is on shows up as a range named after the entry point; with tracing off nothing is emitted."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tracing_can_be_switched_on_and_off_without_a_profiler(dev):
    import numpy as np
    from pothoscomms_amd import _lib
    L = _lib.load()
    x = np.random.default_rng(1).standard_normal((3000, 2)).astype(np.float32)
    want = dev.conj(x)
    _lib.check(L.pcx_trace(1))
    try:
        assert np.array_equal(dev.conj(x), want)
        assert dev.FreqDemod("complex_float32").process(x).shape == (3000,)
    finally:
        _lib.check(L.pcx_trace(0))
    assert np.array_equal(dev.conj(x), want)


def test_ranges_show_up_in_a_rocprofv3_marker_trace(tmp_path):
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        pytest.skip("rocprofv3 not installed")
    out = str(tmp_path / "trace")
    r = subprocess.run([prof, "--marker-trace", "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                        sys.executable, os.path.join(ROOT, "tools", "trace_demo.py")],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and "trace demo done" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    files = glob.glob(os.path.join(out, "**", "*marker*.csv"), recursive=True)
    assert files, (os.listdir(out), r.stderr[-1500:])
    text = "".join(open(f).read() for f in files)
    for name in ("pcx_fir_process_dev", "pcx_fft_transform_dev", "pcx_fmchain_process_dev", "pcx_conj_dev", "pcx_freqdemod_process"):
        assert name in text, name
    assert text.count("pcx_fir_process_dev") == 1           # the call made after pcx_trace(0) left no range
