"""Does the power-capped FIR kernel gain from running on fewer CUs at a higher clock?  Launches the headline
FIR on streams created with hipExtStreamCreateWithCUMask and times 300 launches each (host clock around a
stream synchronise).  PCX_OLS_SLOTS is set per run so the persistent grid matches the CUs in use."""
import ctypes as C, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ncu = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ["PCX_OLS_SLOTS"] = str(4 * ncu)
import torch
from pothoscomms_amd import _lib, device, taps as tp
hip = C.CDLL("libamdhip64.so.7") if False else None
d = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = _lib.load()
import ctypes.util
hipname = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
hip = C.CDLL(hipname)
stream = C.c_void_p()
# 256 CUs = 8 XCDs x 32; the mask is indexed by CU id, XCD-interleaved on this part: keep the first ncu/8 CUs of every XCD
words = (C.c_uint32 * 8)()
per = ncu // 8
bits = 0
for cu in range(256):
    xcd, idx = cu % 8, cu // 8
    if idx < per:
        words[cu // 32] |= (1 << (cu % 32))
rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, words)
assert rc == 0, rc
n, K = 64 * 1024 * 1024, 255
f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps()); f.set_algo(_lib.FIR_OLS_FFT)
lead = (-(K - 1)) % 16
xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(xa, seed=1)
x = xa[lead:]; y = torch.empty((n, 2), dtype=torch.float32, device=d)
torch.cuda.synchronize()
c, p = C.c_size_t(), C.c_size_t()
def launch():
    _lib.check(L.pcx_fir_process_dev(f._h, C.c_void_p(x.data_ptr()), n + K - 1, C.c_void_p(y.data_ptr()), n, C.byref(c), C.byref(p), stream))
for _ in range(300): launch()
hip.hipStreamSynchronize(stream)
t0 = time.perf_counter()
for _ in range(600): launch()
hip.hipStreamSynchronize(stream)
dt = (time.perf_counter() - t0) / 600
out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[-1].split(",")
print("CUs %3d: %.4f ms/launch  %.1f Gsamples/s   (after the run: sclk %s, %s W)" % (ncu, dt * 1e3, n / dt / 1e9, out[5], out[9]))
