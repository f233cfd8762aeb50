import numpy as np, sys
sys.path.insert(0, '.')

from pothoscomms_amd import device as dev
from oracle import oracle
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1317
rng = np.random.default_rng(7000 + seed)
scalar = [oracle.F32, oracle.I16, oracle.F64, oracle.I8, oracle.I32][seed % 5]
n = int(rng.integers(1, 60000))
ph = np.cumsum(rng.uniform(-1.5, 1.5, n))
x = (np.stack([np.cos(ph), np.sin(ph)], 1) * rng.uniform(0.5, 1.5, (n, 1))).astype(oracle.NP_SCALAR[scalar])
ref_blk, gpu_blk = oracle.FreqDemod(scalar), dev.FreqDemod((scalar, True))
cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 8)))]))
print("n", n, "cuts", cuts)
for a, b in zip(cuts[:-1], cuts[1:]):
    ref, got = ref_blk.work(x[a:b]), gpu_blk.process(x[a:b])
    d = np.asarray(got, np.float64) - np.asarray(ref, np.float64)
    d = (d + np.pi) % (2 * np.pi) - np.pi
    i = int(np.argmax(np.abs(d)))
    print(a, b, "maxerr", abs(d[i]) / np.pi, "at", i, "got", got[i], "ref", ref[i], "x", x[a + i - 1: a + i + 1] if a + i > 0 else x[:1])
