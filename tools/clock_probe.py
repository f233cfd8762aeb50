"""Sample GPU clocks/power (rocm-smi) while one FIR variant runs back to back for a few seconds.
    PCX_OLS_VARIANT=12 python tools/clock_probe.py
"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import _lib, device, taps as tp

d = torch.device("cuda", 0)
n, K = 64 * 1024 * 1024, 255
f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps()); f.set_algo(_lib.FIR_OLS_FFT)
lead = (-(K - 1)) % 16
xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(xa, seed=1)
x = xa[lead:]; y = torch.empty((n, 2), dtype=torch.float32, device=d)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().splitlines()[-1])
        except Exception as e:
            samples.append("err %s" % e)
        time.sleep(0.3)
t = threading.Thread(target=sampler); t.start()
for _ in range(200): f.process_dev(x, y)
torch.cuda.synchronize()
t0 = time.time(); iters = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 6.0:
    for _ in range(200): f.process_dev(x, y)
    iters += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; t.join()
print("variant %s: %.4f ms/launch over %d launches" % (os.environ.get("PCX_OLS_VARIANT", "default"), e0.elapsed_time(e1) / iters, iters))
hdr = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[0]
print(hdr)
for s in samples[2:12]: print(s)
