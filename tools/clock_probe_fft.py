"""rocm-smi clocks/power while the 4096-point FFT (65,536 frames), the fused FM chain or rotate loop for 5 s."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import device, taps as tp
d = torch.device("cuda", 0)
n = 64 * 1024 * 1024
x = torch.empty((n + 128, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(x, seed=3)
y = torch.empty((n, 2), dtype=torch.float32, device=d)
yr = torch.empty((n,), dtype=torch.float32, device=d)
fft = device.Fft("complex_float32", 4096, False)
ch = device.FmChain(); ch.set_phase(tp.C4_PHASE); ch.set_taps(tp.c4_taps(), False)
cases = {"fft4096": lambda: fft.transform_dev(x, y, 16384), "fmchain": lambda: ch.process_dev(x[2:], yr, n + 126, n),
         "rotate": lambda: device.rotate(x, 0.7, scalar=device.F32, out=y, n=n)}
for name, fn in cases.items():
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().splitlines()[-1]); time.sleep(0.3)
    t = threading.Thread(target=sampler); t.start()
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.time(); it = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    while time.time() - t0 < 5.0:
        for _ in range(100): fn()
        it += 100; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; t.join()
    f = [s.split(",") for s in samples[3:10]]
    print("%-8s %.4f ms/launch  sclk %s  power %s W" % (name, e0.elapsed_time(e1) / it, f[len(f)//2][5], f[len(f)//2][9]), flush=True)
