"""PCIe-inclusive rate of the host-buffer entry points: what a Pothos work() call pays (255-tap complex_float32 FIR).

  1. the C ABI (pcx_fir_process) on pageable numpy buffers (staged through the bounce buffer) and on page-locked ones (in place);
  2. the /comms/fir_filter block, work() after work() from NATIVE code (pcxb_work_loop: a C++ scheduler's loop, no Python per call):
       a. on the block's own page-locked port buffers,
       b. on the framework's "circular" input buffer -- pageable, mapped twice (filter/FIRFilter.cpp:196-199) -- which the block
          page-locks where it lies on first sight; the window is placed ACROSS the wrap,
       c. on a private (heap) copy of the same window: pageable and not lockable, staged by the CPU -- what (b) cost before round 5;
  3. the PCIe roof of the box, measured here: H2D || D2H of 128 MiB each on two streams (copy engines both ways).
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pothoscomms_amd import _lib, blocks, device, taps as tp

L = _lib.load()
K = 255


def pinned(shape, dtype=np.float32):
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _lib.check(L.pcx_host_alloc(C.byref(p), nbytes))
    return np.ctypeslib.as_array((C.c_char * nbytes).from_address(p.value)).view(dtype).reshape(shape), p


def best_of(fn, reps, rounds=3):
    fn()
    ts = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ts.append((time.perf_counter() - t0) / reps)
    return min(ts)


def pcie_roof(nbytes=128 << 20):
    """(H2D alone, D2H alone, both at once) GB/s per direction: the copy engines, two streams (pcx_pcie_probe)"""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    _lib.check(L.pcx_pcie_probe(nbytes, 5, C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


if __name__ == "__main__":
    up, down, both = pcie_roof()
    roof = max(up, down)          # each direction's ceiling: the link is symmetric and full duplex (bench.py measure_host_path)
    print("PCIe roof of this box (copy engines, 128 MiB): H2D alone %.1f GB/s, D2H alone %.1f, H2D || D2H on two streams %.1f per direction; roof = %.1f" % (up, down, both, roof))
    rng = np.random.default_rng(0)
    print("-- 1. C ABI, pcx_fir_process")
    for n in (1 << 18, 1 << 20, 1 << 22, 1 << 24):
        f = device.FirFilter("complex_float32", "COMPLEX")
        f.set_taps(tp.c1_taps())
        x = rng.uniform(-1, 1, (n + K - 1, 2)).astype(np.float32)
        y = np.empty((n, 2), np.float32)
        c, p = C.c_size_t(), C.c_size_t()
        dt = best_of(lambda: _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, n + K - 1, y.ctypes.data, n, C.byref(c), C.byref(p))), max(3, (1 << 24) // n))
        xp, _ = pinned(x.shape)
        yp, _ = pinned(y.shape)
        xp[:] = x
        dp = best_of(lambda: _lib.check(L.pcx_fir_process(f._h, xp.ctypes.data, n + K - 1, yp.ctypes.data, n, C.byref(c), C.byref(p))), max(5, (1 << 25) // n))
        print("n=%9d  pageable (staged) %7.3f ms %5.2f Gsamples/s | page-locked (in place) %7.3f ms %5.2f Gsamples/s = %4.1f GB/s each way = %.2f of the roof"
              % (n, dt * 1e3, n / dt / 1e9, dp * 1e3, n / dp / 1e9, 8 * n / dp / 1e9, 8 * n / dp / 1e9 / roof))
    print("-- 2. /comms/fir_filter work(), native loop")
    for n in (1 << 18, 1 << 20, 1 << 22, 1 << 24):
        reps = max(5, (1 << 25) // n)
        res = []
        for mode in ("own pinned port buffers", "framework circular, page-locked where it lies", "private copy of the window (staged)"):
            blk = blocks.make("/comms/fir_filter", "complex_float32", "COMPLEX")
            blk.call("setTaps", tp.c1_taps())
            blk.activate()
            yout, _ = blk.port_buffer(1, (n, 2), np.float32)
            circ = None
            if mode.startswith("own"):
                xin, _ = blk.port_buffer(0, (n + K - 1, 2), np.float32)
            else:
                circ = blocks.CircularBuffer(2 * (n + K) * 8)
                cap = circ.size // 8
                xin = circ.view((cap - n // 2) * 8, (n + K - 1) * 8, np.float32).reshape(-1, 2)      # across the wrap
                if mode.startswith("private"):
                    xin = np.empty_like(xin)
            xin[:] = rng.uniform(-1, 1, xin.shape).astype(np.float32)
            blk.work_loop(xin.ctypes.data, n + K - 1, yout.ctypes.data, n, 3)
            best = None
            for _ in range(3):
                t, c, p = blk.work_loop(xin.ctypes.data, n + K - 1, yout.ctypes.data, n, reps)
                assert (c, p) == (n, n)
                best = t / reps if best is None else min(best, t / reps)
            res.append(best)
            blk.close()
            if circ is not None:
                circ.close()
        print("n=%9d  own pinned ports %7.3f ms %5.2f Gs/s | circular, page-locked where it lies %7.3f ms %5.2f Gs/s (%.2f of the roof) | not lockable (staged) %7.3f ms %5.2f Gs/s"
              % (n, res[0] * 1e3, n / res[0] / 1e9, res[1] * 1e3, n / res[1] / 1e9, 8 * n / res[1] / 1e9 / roof, res[2] * 1e3, n / res[2] / 1e9))

    print("-- 3. /comms/fir_filter with setDevices: the call's samples split over shards (here: all on device 0, halos by peer copies), native loop")
    print("   (one device has ONE PCIe link: the shards' transfers share it -- what this shows is the cost of scatter + pass + gather against the")
    print("    in-place call; on a node each device brings its own link)")
    for n in (1 << 22, 1 << 24):
        reps = max(5, (1 << 25) // n)
        row = []
        for devs in ([], [0, 0], [0, 0, 0, 0]):
            blk = blocks.make("/comms/fir_filter", "complex_float32", "COMPLEX")
            blk.call("setTaps", tp.c1_taps())
            if devs:
                blk.call("setDevices", devs)
            blk.activate()
            xin, _ = blk.port_buffer(0, (n + K - 1, 2), np.float32)
            yout, _ = blk.port_buffer(1, (n, 2), np.float32)
            xin[:] = rng.uniform(-1, 1, xin.shape).astype(np.float32)
            blk.work_loop(xin.ctypes.data, n + K - 1, yout.ctypes.data, n, 3)
            best = None
            for _ in range(3):
                t, c, p = blk.work_loop(xin.ctypes.data, n + K - 1, yout.ctypes.data, n, reps)
                assert c == p and c >= n - len(devs), (c, p, n)
                best = t / reps if best is None else min(best, t / reps)
            passes = blk.call("getShardPasses") if devs else 0
            row.append("%s %7.3f ms %5.2f Gs/s%s" % ("single device" if not devs else "%d shards" % len(devs), best * 1e3, n / best / 1e9,
                                                    " (%d sharded passes)" % passes if devs else ""))
            blk.close()
        print("n=%9d  " % n + " | ".join(row))
