"""CPU suite: the C-ABI library loads and exports exactly what include/pcx.h declares.
No compute call is made here (there is no GPU in the build container); argument-error paths
that return before any HIP call are exercised to pin the error behaviour of the factories
and setters (Pothos::InvalidArgumentException <-> PCX_ERR_ARG)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "pcx.h")).read()
    return sorted(set(re.findall(r"PCX_API\s+[\w\s\*]+?\b(pcx_\w+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = header_symbols()
    for must in ("pcx_fir_create", "pcx_fir_set_taps", "pcx_fir_process", "pcx_fir_process_dev", "pcx_fft_create",
                 "pcx_fft_transform_dev", "pcx_freqdemod_process_dev", "pcx_rotate_dev", "pcx_scale_dev", "pcx_abs_dev",
                 "pcx_conj_dev", "pcx_fmchain_process_dev", "pcx_last_error"):
        assert must in syms
    assert len(syms) >= 40


def test_library_exports_every_declared_symbol(pcx):
    lib = pcx._lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    # and the binding's signature table covers the header exactly
    assert sorted(pcx._lib.SIGNATURES) == header_symbols()


def test_library_has_no_unexpected_exports():
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "pothoscomms_amd", "libpcx_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("pcx_"))
    assert exported == header_symbols()


def test_product_does_not_link_or_import_the_oracle():
    """The product path must not route through oracle/ (no CPU fallback)."""
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "pothoscomms_amd", "libpcx_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "pcx_ref" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pothoscomms_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "libpcx_oracle" not in text, f


def test_version_and_error_string(pcx):
    lib = pcx._lib.load()
    assert b"gfx950" in lib.pcx_version()
    assert isinstance(pcx._lib.last_error(), str)


def test_factory_and_setter_argument_errors(pcx):
    """FIRFilterFactory / setTaps / setDecimation / setInterpolation / FFTFactory error behaviour."""
    L = pcx._lib.load()
    h = C.c_void_p()
    assert L.pcx_fir_create(99, 1, 1, C.byref(h)) == pcx._lib.ERR_ARG            # unsupported types
    assert L.pcx_fir_create(pcx.F32, 0, 1, C.byref(h)) == pcx._lib.ERR_ARG       # COMPLEX taps, real stream
    assert "unsupported types" in pcx._lib.last_error()
    assert L.pcx_fir_create(pcx.F32, 1, 1, C.byref(h)) == 0
    assert L.pcx_fir_set_taps(h, None, 0) == pcx._lib.ERR_ARG
    assert "taps cannot be empty" in pcx._lib.last_error()
    assert L.pcx_fir_set_decimation(h, 0) == pcx._lib.ERR_ARG
    assert "decimation cannot be 0" in pcx._lib.last_error()
    assert L.pcx_fir_set_interpolation(h, 0) == pcx._lib.ERR_ARG
    assert L.pcx_fir_set_algo(h, 17) == pcx._lib.ERR_ARG
    # geometry mirrors updateInternals: K = ceil(ntaps / L), inputRequire = M + K - 1
    taps = (C.c_double * (2 * 61))()
    assert L.pcx_fir_set_taps(h, taps, 61) == 0
    assert L.pcx_fir_set_interpolation(h, 3) == 0 and L.pcx_fir_set_decimation(h, 2) == 0
    k, r = C.c_size_t(), C.c_size_t()
    assert L.pcx_fir_get_geometry(h, C.byref(k), C.byref(r)) == 0
    assert (k.value, r.value) == (21, 22)
    # nothing to do -> returns before touching the device
    c, p = C.c_size_t(7), C.c_size_t(7)
    assert L.pcx_fir_process_dev(h, None, 10, None, 100, C.byref(c), C.byref(p), None) == 0
    assert (c.value, p.value) == (0, 0)
    assert L.pcx_fir_destroy(h) == 0
    f = C.c_void_p()
    assert L.pcx_fft_create(pcx.I32, 64, 0, C.byref(f)) == pcx._lib.ERR_ARG      # FFTFactory: unsupported type
    # (2 x a prime beyond one workgroup's LDS used to be rejected here; it now takes the chirp-z plan, whose tables need the device)
    # (complex_int16 beyond 32768 bins used to be PCX_ERR_UNSUPPORTED; it now takes the stage-per-launch plan, whose tables need the device)
    assert L.pcx_fft_create(pcx.F32, (1 << 26) * 3, 0, C.byref(f)) == pcx._lib.ERR_UNSUPPORTED      # beyond every plan
    assert L.pcx_fft_create(pcx.F32, 0, 0, C.byref(f)) == pcx._lib.ERR_ARG
    assert L.pcx_rotate_dev(42, 1.0, 0.0, None, None, 0, None) == pcx._lib.ERR_ARG


def test_python_wrappers_raise_like_the_reference(dev):
    with pytest.raises(ValueError):
        dev.FirFilter("float32", "COMPLEX")
    with pytest.raises(ValueError):
        dev.FirFilter("complex_float32", "SOMETHING")
    with pytest.raises(ValueError):
        dev.FirFilter("complex_uint9", "REAL")
    f = dev.FirFilter("complex_int16", "REAL")
    with pytest.raises(ValueError):
        f.set_taps([])
    with pytest.raises(ValueError):
        f.set_decimation(0)
    assert f.geometry() == (1, 1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: a missing extension is an ImportError naming the build step."""
    from pothoscomms_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libpcx_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_blocks_library_exports_every_symbol_of_pcx_blocks_h():
    """include/pcx_blocks.h (runner ABI of the host-side block layer) <-> libpcx_blocks.so, both directions"""
    import subprocess
    src = open(os.path.join(ROOT, "include", "pcx_blocks.h")).read()
    declared = sorted(set(re.findall(r"PCXB_API\s+[\w\s\*]+?\b(pcxb_\w+)\s*\(", src)))
    assert len(declared) >= 25 and "pcxb_work_ports" in declared
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "pothoscomms_amd", "libpcx_blocks.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("pcxb_"))
    assert exported == declared
