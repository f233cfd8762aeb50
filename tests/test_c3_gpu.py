"""GPU suite: BASELINE.json configs[3] at its own workload, as far as ONE GPU allows -- a 512 Mi-sample complex_float32 stream in
8 overlap-save shards of 67,108,864 samples, 255 taps, the 254-sample halo moved between neighbouring shards on every pass -- and
the same split for the fused chain of configs[4] (halo of K = 127 samples).

The eight shards live on device 0 (NodeStream([0] * 8, PEER_COPY): RCCL refuses two ranks on one device, so the halo goes by peer
copies; offsets, gate words, launch-per-shard and event ordering are the code an 8-GPU node runs, the transport is not).  8 GiB of
samples: 4 GiB in + 4 GiB out, against 288 GB of HBM.

What is checked, on two passes with different data and NaN in every halo slot before each:
  * every one of the 7 seams, 4,096 outputs either side, against the oracle's FIR loop (filter/FIRFilter.cpp:286-302, whose window
    n .. n+K-1 at :296-299 is what defines the halo) -- the oracle's input is cut from the NEIGHBOURING shards' own samples, so a
    halo that is not the left neighbour's tail cannot pass;
  * the stream start (shard 0 keeps the stream's own history);
  * every shard bit-identical to a plain pcx_fir_process_dev call on its completed [halo | shard] buffer.
"""
import ctypes as C
import time

import numpy as np
import pytest

from tests.util import TOL, ang_err, nerr

pytestmark = pytest.mark.gpu

G = 8
SHARD = 64 * 1024 * 1024
W = 4096


def _d2h(L, ptr, rows, cols, dtype=np.float32):
    from pothoscomms_amd import _lib
    a = np.empty((rows, cols) if cols else (rows,), dtype)
    _lib.check(L.pcx_memcpy_d2h(a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), a.nbytes, None))
    return a


def _fill(L, ns, K, seed, fm=None):
    """every shard's [K-1 history | C samples] straight from the node-wide stream (shard g starts at sample g*C) -- the counter-hash
    stream of the bench, or (fm: a device tensor of K-1+C samples) the FM test signal at a different amplitude in every shard --
    then NaN over every halo slot but shard 0's: only the exchange of the pass can make the seams right"""
    import torch

    from pothoscomms_amd import _lib
    halo = K if getattr(ns, "chain", False) else K - 1
    nan = np.full((halo, 2), np.nan, np.float32)
    for g in range(ns.nshards):
        i, _, s, _ = ns.buffers(g)
        if fm is None:
            _lib.check(L.pcx_fill_uniform_f32_dev(C.c_void_p(i), 2 * (K - 1 + ns.C), seed, 2 * g * ns.C, C.c_void_p(s)))
        else:
            # the tile of the test signal divides the shard length, so every shard would hold the SAME samples and a halo taken from
            # the wrong shard would go unnoticed: each shard (and each pass) gets its own amplitude -- not its own phase: a phase jump
            # at the seam could bring the filter output close to zero there, where its angle is ill-conditioned
            xg = fm * (1.0 + 0.05 * g + 0.001 * seed)
            torch.cuda.synchronize()
            _lib.check(L.pcx_memcpy_d2d(C.c_void_p(i), C.c_void_p(xg.data_ptr()), 8 * (K - 1 + ns.C), C.c_void_p(s)))
            _lib.check(L.pcx_stream_sync(C.c_void_p(s)))
            del xg
        if g > 0:
            _lib.check(L.pcx_memcpy_h2d(C.c_void_p(i - 8 * (halo - (K - 1))), nan.ctypes.data_as(C.c_void_p), nan.nbytes, C.c_void_p(s)))
        _lib.check(L.pcx_stream_sync(C.c_void_p(s)))


def _oracle_fir(o, taps, x, n, complex_taps=True):
    blk = o.Fir(o.F32, True, complex_taps)
    blk.set_taps(taps)
    blk.activate()
    ref, c, p, _ = blk.work(x, n)
    assert p == n
    return ref


@pytest.mark.parametrize("submit_threads", [False, True], ids=["one_thread", "submit_threads"])
def test_c3_fir_eight_shards_of_64Mi_on_one_device(oracle, submit_threads):
    """submit_threads: every device's share of a pass queued by a thread of its own (pcx_shard_set_submit_threads) -- same seams, same bits"""
    import torch

    from pothoscomms_amd import _lib, device, taps as tp
    L = _lib.load()
    h = tp.c1_taps()
    K, Cs = len(h), SHARD
    assert K == 255
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_submit_threads(submit_threads)
    ns.set_taps(h)
    ns.configure(Cs)
    plain = device.FirFilter("complex_float32", "COMPLEX")
    plain.set_taps(h)
    lead = (-(K - 1)) % 16
    xa = torch.zeros((lead + K - 1 + Cs, 2), dtype=torch.float32, device="cuda:0")
    y = torch.empty((Cs, 2), dtype=torch.float32, device="cuda:0")
    y_shard = torch.empty_like(y)
    for rep, seed in enumerate((4, 41)):      # C3 is seed 4
        _fill(L, ns, K, seed)
        t0 = time.perf_counter()
        ns.step()
        ns.sync()                             # raises if a gated launch gave up waiting for its halo
        dt = time.perf_counter() - t0
        print("configs[3] on one device, pass %d: 8 x %d samples in %.3f ms (%.1f Gsamples/s)" % (rep, Cs, dt * 1e3, G * Cs / dt / 1e9))
        bufs = [ns.buffers(g) for g in range(G)]
        # stream start: shard 0 filters from the stream's own K-1 history
        x0 = _d2h(L, bufs[0][0], K - 1 + W, 2)
        assert nerr(_d2h(L, bufs[0][1], W, 2), _oracle_fir(oracle, h, x0, W)) <= TOL
        # and the stream's end: the last W outputs of the last shard (first / last samples and every shard boundary)
        xe = _d2h(L, bufs[G - 1][0] + 8 * (Cs - W), W + K - 1, 2)
        assert nerr(_d2h(L, bufs[G - 1][1] + 8 * (Cs - W), W, 2), _oracle_fir(oracle, h, xe, W)) <= TOL
        for g in range(1, G):
            # outputs g*C - W .. g*C + W - 1 of the stream: their inputs, cut from shard g-1's and shard g's OWN samples
            left = _d2h(L, bufs[g - 1][0] + 8 * (Cs - W), W + K - 1, 2)              # in_{g-1}[C-W : C+K-1]
            right = _d2h(L, bufs[g][0] + 8 * (K - 1), W, 2)                          # in_g[K-1 : K-1+W]
            ref = _oracle_fir(oracle, h, np.concatenate([left, right]), 2 * W)
            got = np.concatenate([_d2h(L, bufs[g - 1][1] + 8 * (Cs - W), W, 2), _d2h(L, bufs[g][1], W, 2)])
            assert np.isfinite(got).all(), "seam %d pass %d" % (g, rep)
            assert nerr(got, ref) <= TOL, "seam %d pass %d: %g" % (g, rep, nerr(got, ref))
            # and the halo slot holds exactly the left neighbour's tail
            assert np.array_equal(_d2h(L, bufs[g][0], K - 1, 2), left[W:]), "halo %d pass %d" % (g, rep)
        for g in range(G):
            i, o, s, d = bufs[g]
            _lib.check(L.pcx_memcpy_d2d(C.c_void_p(xa[lead:].data_ptr()), C.c_void_p(i), 8 * (K - 1 + Cs), None))
            _lib.check(L.pcx_memcpy_d2d(C.c_void_p(y_shard.data_ptr()), C.c_void_p(o), 8 * Cs, None))
            torch.cuda.synchronize()
            assert plain.process_dev(xa[lead:], y) == (Cs, Cs)
            torch.cuda.synchronize()
            assert torch.equal(y, y_shard), "shard %d pass %d differs from a plain call on its completed buffer" % (g, rep)
    ns.close()


def _oracle_chain(o, taps, phase, x, n):
    r = o.rotate(x, phase)
    blk = o.Fir(o.F32, True, False)
    blk.set_taps(taps)
    blk.activate()
    yy, c, p, _ = blk.work(r, n)
    assert p == n
    return o.FreqDemod(o.F32).work(yy)


def test_c3_split_of_the_fused_chain_eight_shards_of_64Mi(oracle):
    """Rotate -> FIR(127 real taps) -> FreqDemod (configs[4]) over the same 8 x 64 Mi split: halo of K = 127 samples (the FIR's K-1 and
    the sample FreqDemod's _prev needs, demod/FreqDemod.cpp:63-65), one extra output in front of every shard but the first"""
    import torch

    from pothoscomms_amd import _lib, device, taps as tp
    L = _lib.load()
    h = tp.c4_taps()
    K, Cs = len(h), SHARD
    assert K == 127
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_chain(True, tp.C4_PHASE)
    ns.set_taps(h, complex_taps=False)
    ns.configure(Cs)
    plain = device.FmChain()
    plain.set_phase(tp.C4_PHASE)
    plain.set_taps(h, False)
    pad = (K + 31) // 32 * 32 - K
    xa = torch.zeros((16 + K + Cs, 2), dtype=torch.float32, device="cuda:0")
    y = torch.empty((Cs + 1,), dtype=torch.float32, device="cuda:0")
    y_shard = torch.empty_like(y)
    # SURVEY 8d's C4 signal (FM, small noise: the envelope never vanishes, so the 1e-5 bar on the ANGLE is meaningful -- on a stream of
    # uniform noise the filter output comes arbitrarily close to zero and its angle is ill-conditioned there): 1 Mi samples from
    # taps.fm_test_signal, tiled on the device; the phase jumps at tile and shard seams are part of the stream both sides see
    tile = torch.from_numpy(tp.fm_test_signal(1 << 20).view(np.float32).reshape(-1, 2)).to("cuda:0")
    fm = tile.repeat((K - 1 + Cs + (1 << 20) - 1) // (1 << 20), 1)[:K - 1 + Cs].contiguous()
    del tile
    for rep, seed in enumerate((5, 51)):
        _fill(L, ns, K, seed, fm)
        ns.step()
        ns.sync()
        bufs = [ns.buffers(g) for g in range(G)]
        x0 = _d2h(L, bufs[0][0], K - 1 + W, 2)
        assert ang_err(_d2h(L, bufs[0][1], W, 0), _oracle_chain(oracle, h, tp.C4_PHASE, x0, W)) <= TOL
        xe = _d2h(L, bufs[G - 1][0] + 8 * (Cs - W - 1), W + 1 + K - 1, 2)          # the stream's end (one output in front seeds the demodulator)
        assert ang_err(_d2h(L, bufs[G - 1][1] + 4 * (Cs - W), W, 0), _oracle_chain(oracle, h, tp.C4_PHASE, xe, W + 1)[1:]) <= TOL
        for g in range(1, G):
            # outputs g*C - W .. g*C + W - 1 and the one before them (the demodulator's predecessor), inputs from the shards' own samples
            left = _d2h(L, bufs[g - 1][0] + 8 * (Cs - W - 1), W + 1 + K - 1, 2)      # in_{g-1}[C-W-1 : C+K-1]
            right = _d2h(L, bufs[g][0] + 8 * (K - 1), W, 2)
            ref = _oracle_chain(oracle, h, tp.C4_PHASE, np.concatenate([left, right]), 2 * W + 1)[1:]
            got = np.concatenate([_d2h(L, bufs[g - 1][1] + 4 * (Cs - W), W, 0), _d2h(L, bufs[g][1], W, 0)])
            assert np.isfinite(got).all(), "seam %d pass %d" % (g, rep)
            assert ang_err(got, ref) <= TOL, "seam %d pass %d: %g" % (g, rep, ang_err(got, ref))
            # the halo slot (K samples, one in front of the history) holds exactly the left neighbour's tail
            assert np.array_equal(_d2h(L, bufs[g][0] - 8, K, 2), left[W:]), "halo %d pass %d" % (g, rep)
        for g in range(G):
            i, o, s, d = bufs[g]
            extra = 0 if g == 0 else 1        # shard 0: [K-1 history | C] -> C outputs; the others: [K halo | C] -> 1 + C, the first dropped
            lead = (pad + extra) % 16
            n_in, n_out = K - 1 + extra + Cs, Cs + extra
            _lib.check(L.pcx_memcpy_d2d(C.c_void_p(xa[lead:].data_ptr()), C.c_void_p(i - 8 * extra), 8 * n_in, None))
            _lib.check(L.pcx_memcpy_d2d(C.c_void_p(y_shard.data_ptr()), C.c_void_p(o - 4 * extra), 4 * n_out, None))
            torch.cuda.synchronize()
            plain.reset()
            assert plain.process_dev(xa[lead:], y, n_in, n_out) == (n_out, n_out)
            torch.cuda.synchronize()
            assert torch.equal(y[:n_out], y_shard[:n_out]), "chain shard %d pass %d differs from a plain call on its completed buffer" % (g, rep)
    ns.close()
