"""pothoscomms_amd -- MI355X (gfx950) device path for the PothosComms streaming-DSP blocks
/comms/fir_filter, /comms/fft, /comms/freq_demod, /comms/{rotate,scale,abs,conjugate}.

Layout
  csrc/            hand-written HIP kernels + the extern "C" boundary (include/pcx.h)
  libpcx_hip.so    built in-tree by `make -C pothoscomms_amd/csrc` (__graft_entry__.build())
  _lib.py          ctypes binding (raises if the library is missing: no CPU fallback)
  device.py        handle wrappers for numpy (host) / torch (device memory) buffers
  taps.py          windowed-sinc tap design for the synthetic workloads
  stream.py        overlap-save sharding of one stream across the node's GPUs (RCCL halo)
"""
from . import _lib  # noqa: F401
from ._lib import (F32, F64, I8, I16, I32, I64, FIR_AUTO, FIR_DIRECT, FIR_EXACT, FIR_OLS_FFT,  # noqa: F401
                   InvalidArgument, PcxError, Unsupported)

__version__ = "0.1.0"
