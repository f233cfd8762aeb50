"""Tap design and synthetic streams for the BASELINE workloads.

The reference designs taps with the un-vendored `spuce` library behind
/comms/fir_designer (filter/FIRDesigner.cpp:387-477); that library is not in the
reference tree, so the workloads use this self-contained Hann-windowed-sinc
recipe instead.  Everything is float64 numpy and deterministic.
"""
import numpy as np


def lowpass(num_taps, cutoff):
    """Hann-windowed sinc low-pass, cutoff in cycles/sample (0 < cutoff < 0.5), unity DC gain."""
    n = np.arange(num_taps, dtype=np.float64)
    m = n - (num_taps - 1) / 2.0
    h = 2.0 * cutoff * np.sinc(2.0 * cutoff * m)
    if num_taps > 1:
        h *= 0.5 - 0.5 * np.cos(2.0 * np.pi * (n + 1) / (num_taps + 1))   # Hann without zero end points
    return h / np.sum(h)


def complex_bandpass(num_taps, cutoff, center):
    """Low-pass prototype shifted to `center` cycles/sample: complex taps (COMPLEX tapsType)."""
    n = np.arange(num_taps, dtype=np.float64)
    return lowpass(num_taps, cutoff) * np.exp(2j * np.pi * center * n)


# the BASELINE.json configurations, as (taps, description)
def c0_taps():   # configs[0]: 63 taps, 1 Mi samples
    return complex_bandpass(63, 0.1, 0.05)


def c1_taps():   # configs[1] and [3]: 255 taps
    return complex_bandpass(255, 0.05, 0.05)


def c4_taps():   # configs[4]: 127 real taps
    return lowpass(127, 0.1)


C4_PHASE = 0.7


def fm_test_signal(n, seed=5, start=0):
    """configs[4] input: x[n] = exp(j phi[n]), phi[n] = phi[n-1] + 2 pi (0.02 + 0.01 sin(2 pi n / 1000)),
    plus uniform noise of amplitude 1e-3 so the envelope never vanishes.  Returns complex64."""
    idx = np.arange(start, start + n, dtype=np.float64)
    inc = 2.0 * np.pi * (0.02 + 0.01 * np.sin(2.0 * np.pi * idx / 1000.0))
    # closed form of the running sum so any window [start, start+n) is reproducible
    base = 2.0 * np.pi * 0.02 * (idx + 1)
    # sum_{m<=n} sin(2 pi m/1000) via the Dirichlet-type closed form
    w = 2.0 * np.pi / 1000.0
    ssum = (np.sin(w * (idx + 1) / 2.0) * np.sin(w * idx / 2.0)) / np.sin(w / 2.0)
    phi = base + 2.0 * np.pi * 0.01 * ssum
    del inc
    rng = np.random.default_rng(seed + start)
    noise = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)) * 1e-3
    return (np.exp(1j * phi) + noise).astype(np.complex64)
