import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import device as dev, _lib
variant = sys.argv[1]
d = torch.device("cuda", 0)
n = 1 << 16
x = torch.empty((n, 2), dtype=torch.float32, device=d); dev.fill_uniform_f32_dev(x, seed=4)
y = torch.empty(n, dtype=torch.float32, device=d)
s = torch.cuda.Stream(d)
h = dev.FreqDemod("complex_float32")
def run():
    h.reset(); h.process_dev(x, y, n, stream=s)
if variant == "eager_on_null":
    h.reset(); h.process_dev(x, y, n); 
elif variant == "no_eager_reset":
    with torch.cuda.stream(s): h.process_dev(x, y, n, stream=s)
else:
    with torch.cuda.stream(s): run()
torch.cuda.synchronize(); want = y.clone()
if variant in ("no_eager_reset",):
    with torch.cuda.stream(s): run()
    torch.cuda.synchronize(); want = y.clone()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
st = C.c_void_p(s.cuda_stream)
g, ge = C.c_void_p(), C.c_void_p()
assert hip.hipStreamBeginCapture(st, 1) == 0
run()
assert hip.hipStreamEndCapture(st, C.byref(g)) == 0
assert hip.hipGraphInstantiate(C.byref(ge), g, None, None, 0) == 0
for rep in range(3):
    if variant != "no_fill":
        y.fill_(float("nan"))
    torch.cuda.synchronize()
    assert hip.hipGraphLaunch(ge, st) == 0
    torch.cuda.synchronize()
    print(variant, "replay", rep, float(y[0]), float(want[0]), int((y != want).sum()))
