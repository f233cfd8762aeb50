/*
 * pcx.h -- C ABI of libpcx_hip.so: the MI355X (gfx950) device path behind the
 * PothosComms streaming-DSP blocks.
 *
 * The reference has no FFI: each block is a C++ class whose work() runs the
 * arithmetic inline.  This header is the boundary a Pothos
 * plugin TU (pothos_plugin/) calls from inside
 * work()/setters instead of those loops.  One entry point per reference
 * loop / setter it replaces, cited as <file>:<line> of the reference tree.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ types, no exceptions; every call
 *     returns PCX_OK (0) or a negative pcx_status and records a message
 *     retrievable with pcx_last_error() (thread-local).
 *   - scalar types are pcx_scalar codes; a complex stream is interleaved
 *     (re, im) pairs of that scalar -- the memory layout of std::complex<T>.
 *   - "elements" are stream elements (one complex pair = one element).
 *   - *_dev variants take DEVICE pointers and a hipStream_t (as void*) and
 *     only enqueue work; the plain variants take HOST pointers (the
 *     BufferChunk memory of a Pothos port) and return after the result is in
 *     `out`.  Page-locked host memory (pcx_host_alloc, hipHostMalloc,
 *     hipHostRegister -- what the module's BufferManagers hand out) is
 *     processed IN PLACE: the kernels read and write it over PCIe, both
 *     directions at once, on a launch shape of their own (a link-bound call
 *     runs on a few dozen workgroups that walk many pieces each, so that one
 *     piece's loads travel beside the previous piece's stores).  Memory the
 *     FRAMEWORK owns can be page-locked where it lies (pcx_host_register,
 *     pcx_host_register_mapping).  Pageable memory is staged through a
 *     page-locked bounce buffer and a device workspace owned by the handle
 *     (CPU copy, pinned H2D, kernel, pinned D2H, CPU copy).
 *   - *_create and the setters return with everything they zero or upload
 *     COMPLETE on the device: a handle can be used at once on any stream.
 *   - every handle owns a non-blocking stream for its host-pointer calls, so
 *     blocks on different actor threads overlap on the device; nothing runs
 *     on the legacy default stream.
 *   - handles are not thread-safe; one handle per block instance, exactly as
 *     Pothos serialises work() and setters on one actor.
 *   - a handle is bound to ONE device: the device current (pcx_set_device) on
 *     the thread that CREATES it.  Later calls from any thread run on that
 *     device and leave the caller's current device unchanged, so a single
 *     Pothos process can place blocks on different GPUs.  The stateless maps
 *     run on the calling thread's current device.
 *   - ordering: a *_dev call on another stream than the handle's previous
 *     call is ordered behind it (an event), carried state and all; setters
 *     that rewrite device tables first wait for the handle's outstanding
 *     work; reset() is enqueued behind the previous call and ahead of the
 *     next.  Setters cannot be captured into a hipGraph; *_dev calls and
 *     reset() can, once the handle's tables are uploaded (its first call
 *     after a setter) and on the stream of the handle's previous call.
 *     A handle with carried state (pcx_freqdemod, pcx_fmchain) keeps it in
 *     two device slots it alternates between, and a captured call replays
 *     with the slots it was captured with: capture reset() in front of the
 *     calls (every replay starts a new stream), or an EVEN number of calls
 *     per handle (every replay continues where the previous one ended).
 *   - the library reads no environment variable.
 *   - input and output buffers of one call must not overlap, with two exceptions the reference
 *     relies on or that cost nothing: the same-size element-wise maps (rotate, scale, conj, arith)
 *     accept out == in exactly (Arithmetic forwards input 0's buffer, Arithmetic.cpp:157-158), and
 *     the FFT accepts out == in.  abs/angle (narrower output), FIR, FreqDemod and the fused chain
 *     read what another lane may already have overwritten: no aliasing.
 *   - there is NO CPU fallback: a type/size the device path does not implement
 *     returns PCX_ERR_UNSUPPORTED.  After round 3 nothing of the path does: every numBins the reference
 *     constructs has a plan for every type FFTFactory accepts (complex_int16 beyond one workgroup's LDS runs
 *     kf_work's stages one launch each, bit-exact; transforms beyond 2^26 bins are refused), the FIR's
 *     PCX_FIR_OLS_FFT request is refused only for geometries it does not cover (PCX_FIR_AUTO always runs).
 */
#ifndef PCX_H
#define PCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCX_API __attribute__((visibility("default")))

typedef enum pcx_scalar {
    PCX_F64 = 0, PCX_F32 = 1, PCX_I64 = 2, PCX_I32 = 3, PCX_I16 = 4, PCX_I8 = 5,
    /* unsigned element types: accepted by pcx_arith* only (arithmeticFactory, Arithmetic.cpp:284-296) */
    PCX_U64 = 6, PCX_U32 = 7, PCX_U16 = 8, PCX_U8 = 9
} pcx_scalar;

typedef enum pcx_status {
    PCX_OK = 0,
    PCX_ERR_ARG = -1,          /* Pothos::InvalidArgumentException territory */
    PCX_ERR_UNSUPPORTED = -2,  /* valid in the reference, not implemented on device */
    PCX_ERR_HIP = -3,          /* a HIP runtime call failed */
    PCX_ERR_STATE = -4         /* call sequence error (e.g. process before taps upload) */
} pcx_status;

PCX_API const char *pcx_last_error(void);
PCX_API const char *pcx_version(void);

/* ---- Pothos::Util::floatToQ / fromQ for INTEGER element types --------------------------------------------------------------
 * Call sites in the reference: filter/FIRFilter.cpp:300 (fromQ<OutType>(y_n)) and :348 (floatToQ<QTapsType>(taps)),
 * math/Rotate.cpp:21,74, math/Scale.cpp:21,73.  The header that defines them, PothosCore's include/Pothos/Util/QFormat.hpp, is
 * NOT under /root/reference (CMakeLists.txt:8 only names the package), and the reference tests that go through it
 * (math/TestRotate.cpp:53, math/TestScale.cpp:52: multiples of 10 times 0, +-0.5, +-1, tolerance 1) leave TWELVE readings
 * standing (profiles/r02/qformat_enumeration.txt):
 *     fractional bits n     half the Q word (PCX_Q_FRAC_HALF_Q)    |  half the ELEMENT word (PCX_Q_FRAC_HALF_ELEM)
 *     floatToQ<T>(x)        T(ldexp(x, n)), the cast truncating    |  rounding to nearest (ties away from zero)
 *     fromQ<T>(q)           q >> n (floor)  |  q / 2^n (toward zero)  |  (q + 2^(n-1)) >> n (nearest, ties up)
 * with Q the widened accumulator type of the factories (int8 -> int16, int16 -> int32, int32 -> int64, int64 -> int64:
 * FIRFilter.cpp:377-382, Rotate.cpp:143-157, Scale.cpp:142-157).  Every integer FIR, Rotate and Scale path of this library takes
 * the reading as a parameter; the all-zero pcx_qformat -- HALF_Q, TRUNCATE, FLOOR, the builder's recollection of the header -- is the
 * built-in default (kDefaultQFormat, csrc/pcx_internal.hpp: the ONE line to change the day the header is read).  Products and sums
 * wrap modulo 2^bits(Q) under every reading (std::complex<intN> arithmetic), the result of fromQ is truncated to the element
 * width.  Floating-point element types are not affected: both functions are plain casts there. */
typedef enum pcx_q_frac { PCX_Q_FRAC_HALF_Q = 0, PCX_Q_FRAC_HALF_ELEM = 1 } pcx_q_frac;
typedef enum pcx_q_to { PCX_Q_TRUNCATE = 0, PCX_Q_NEAREST = 1 } pcx_q_to;
typedef enum pcx_q_from { PCX_Q_FLOOR = 0, PCX_Q_TOWARD_ZERO = 1, PCX_Q_ROUND = 2 } pcx_q_from;
typedef struct pcx_qformat {
    int frac;          /* pcx_q_frac */
    int float_to_q;    /* pcx_q_to */
    int from_q;        /* pcx_q_from */
} pcx_qformat;
/* the process-wide reading: what handles created AFTERWARDS start with and what the stateless pcx_rotate* / pcx_scale* use.
 * NULL restores the built-in default.  Not synchronised with running calls: set it before the blocks are made. */
PCX_API int pcx_set_qformat(const pcx_qformat *q);
PCX_API int pcx_get_qformat(pcx_qformat *q);

/* ---- device plumbing (for hosts without their own HIP runtime binding) ---- */
PCX_API int pcx_device_count(int *count);
PCX_API int pcx_set_device(int ordinal);
PCX_API int pcx_get_device(int *ordinal);   /* the calling thread's current device */
PCX_API int pcx_dev_alloc(void **dptr, size_t bytes);
PCX_API int pcx_dev_free(void *dptr);
PCX_API int pcx_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
PCX_API int pcx_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
PCX_API int pcx_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
/* what kind of memory a buffer lives in -- a block's port buffer may be pageable host memory, one of the page-locked slabs of
 * pcx_host_alloc, or (an edge between two blocks of this module) device memory the CPU must not touch */
typedef enum pcx_pointer_kind_t { PCX_PTR_PAGEABLE = 0, PCX_PTR_PAGE_LOCKED = 1, PCX_PTR_DEVICE = 2 } pcx_pointer_kind_t;
PCX_API int pcx_pointer_kind(const void *p, int *kind);
/* bytes of `src` -- any of the three kinds -- into ordinary host memory, complete on return (FIRFilter::work's burst flush,
 * FIRFilter.cpp:263-272, builds its zero-padded tail on the host) */
PCX_API int pcx_memcpy_to_host(void *dst_host, const void *src, size_t bytes);
PCX_API int pcx_stream_sync(void *stream);
/* ROCTx ranges around every data-plane entry point (pcx_*_process[_dev], pcx_fft_transform[_dev], the maps,
 * pcx_shard_scatter/step/gather), named after the function: `rocprofv3 --marker-trace --kernel-trace` then shows which
 * call each kernel belongs to.  Off by default; pcx_trace(1) loads the ROCTx library (librocprofiler-sdk-roctx.so.1,
 * else libroctx64.so.4) and returns PCX_ERR_UNSUPPORTED when there is none; pcx_trace(0) switches the ranges off.
 * Process-wide. */
PCX_API int pcx_trace(int on);
/* page-locked host memory for port buffers: a Pothos BufferManager that hands out slabs from
 * pcx_host_alloc lets the plain (host-pointer) entry points copy at PCIe line rate instead of
 * staging pageable memory */
PCX_API int pcx_host_alloc(void **hptr, size_t bytes);
PCX_API int pcx_host_free(void *hptr);
/* Page-lock memory the FRAMEWORK owns, where it lies (hipHostRegister, portable + mapped): afterwards the host-pointer entry points
 * treat it like a pcx_host_alloc slab (in place over PCIe, nothing staged).  The FIR's input inside Pothos is such memory: the
 * reference asks the framework for its "circular" manager (filter/FIRFilter.cpp:196-199), whose buffer is pageable and mapped twice
 * back to back so that a window may run across the wrap.
 *   pcx_host_register(ptr, bytes)    [ptr, ptr + bytes), page-aligned by the caller.  Memory that is already page-locked is PCX_OK.
 *   pcx_host_unregister(ptr)         the range registered at ptr (a range this library did not register: PCX_ERR_ARG).
 *   pcx_host_register_mapping(p, bytes, max_bytes, &base, &len)
 *                                    finds the mapping(s) of this process that hold [p, p + bytes) (/proc/self/maps), widens the range
 *                                    over ADJACENT mappings of the same shared file object -- the two halves of a double-mapped
 *                                    circular buffer are one object mapped twice -- and page-locks all of it; *base / *len say what
 *                                    was locked (pass *base to pcx_host_unregister).  A range beyond max_bytes (0: 1 GiB), memory that
 *                                    is not a shared mapping (a heap arena, a stack: locking one would pin whatever else lives
 *                                    there) and memory somebody else page-locked leave *base NULL and return PCX_OK: nothing to undo.
 * Registrations are COUNTED: a range this library holds already (two blocks on one buffer, one block under a second window)
 * gets one more holder and the same *base / *len, and pcx_host_unregister only unlocks when the last holder has let go.
 *   pcx_host_mapping_alive(base, &alive)
 *                                    is what was mapped at base when it was locked (device and inode of the shared object, the whole
 *                                    range, read-write) still mapped there?  The lock belongs to the MAPPING: after the framework has
 *                                    unmapped a buffer, a new one at the same address is not page-locked, whatever the old entry says.
 *                                    A holder asks when its block is activated, and lets go of what is no longer alive.
 *   pcx_host_release_range(p, bytes) for the OWNER of the memory, in front of munmap: every registration of this library that overlaps
 *                                    [p, p + bytes) is dropped, whoever holds it, and unlocked.
 * The /comms/fir_filter block calls pcx_host_register_mapping the first time it sees a pageable port buffer and whenever the
 * buffer's address leaves what it has locked; it lets go in deactivate() (a topology that is re-committed re-allocates its buffers
 * between deactivate and activate) and in its destructor, and checks what it still holds in activate(). */
PCX_API int pcx_host_register(void *ptr, size_t bytes);
PCX_API int pcx_host_unregister(void *ptr);
PCX_API int pcx_host_register_mapping(const void *p, size_t bytes, size_t max_bytes, void **base, size_t *len);
PCX_API int pcx_host_mapping_alive(const void *base, int *alive);
PCX_API int pcx_host_release_range(const void *p, size_t bytes);
/* synthetic stream generator on the device: the same splitmix64 counter hash as
 * the oracle's orc_fill_uniform_f32 (uniform [-1,1), bit-identical values) */
PCX_API int pcx_fill_uniform_f32_dev(float *dst_dev, size_t n_scalars, uint64_t seed, uint64_t offset, void *stream);
/* measurement aid (no reference counterpart): the PCIe roof of this box as the copy engines see it.  `bytes` of page-locked host
 * memory each way, `reps` transfers queued back to back behind a warm-up one: host -> device alone, device -> host alone, and BOTH at
 * once on two streams of the probe's own -- GB/s per direction.  bench.py prices the host-pointer path (secondary.host_path, bound "pcie") on the
 * FASTER of the two directions alone (the link is full duplex: what one direction carries alone is the ceiling of each; *both_gbs is
 * reported beside it, it depends on which HIP runtime the process loaded), measured in the same run.  Allocates and frees 2 x bytes
 * of host and of device memory; blocks until done. */
PCX_API int pcx_pcie_probe(size_t bytes, int reps, double *h2d_gbs, double *d2h_gbs, double *both_gbs);
/* measurement aid (no reference counterpart): ONE wave on `stream` spins for spin_us microseconds and writes the shader clock it
 * ran at, in MHz, to *mhz_dev (shader cycles from s_memtime over the 100 MHz s_memrealtime).  Queued on a stream of its own beside
 * a running workload it reads the clock the workload is held at by the package power cap -- bench.py uses it to turn the profiled
 * VALU instruction count of a kernel into a share of SIMD issue time (`roofline.valu`). */
PCX_API int pcx_clock_probe_dev(float *mhz_dev, unsigned spin_us, void *stream);

/* ===================================================================== *
 *  /comms/fir_filter      filter/FIRFilter.cpp
 * ===================================================================== */
typedef struct pcx_fir pcx_fir;

typedef enum pcx_fir_algo {
    PCX_FIR_AUTO = 0,    /* OLS_FFT when it applies and pays, else DIRECT */
    PCX_FIR_DIRECT = 1,  /* time-domain LDS-tiled dot product (FMA) */
    PCX_FIR_OLS_FFT = 2, /* frequency-domain overlap-save on 4096-sample blocks: complex_float32 (K <= 8193 -- beyond
                            2049 taps the taps in partitions of 2048 against the previous windows' spectra, decimating
                            and interpolating filters as well, K per polyphase row), real float32 (K <= 8193, likewise),
                            complex_float64 and M = L = 1 (K <= 4097), complex_int16 / complex_int8 and real float64 / int16 / int8 with
                            M = L = 1 (K <= 4097; integers bit-exact: the rounded double-precision sums are the integer
                            convolution); anything else -> PCX_ERR_UNSUPPORTED */
    PCX_FIR_EXACT = 3    /* time-domain, reference accumulation order, no FMA:
                            bit-identical to FIRFilter.cpp:295-300 for float types */
} pcx_fir_algo;

/* FIRFilterFactory(dtype, tapsType), FIRFilter.cpp:369-384.
 * complex_taps = 1 is tapsType "COMPLEX" (complex element types only). */
PCX_API int pcx_fir_create(int scalar, int is_complex, int complex_taps, pcx_fir **out);
PCX_API int pcx_fir_destroy(pcx_fir *h);
/* setTaps, FIRFilter.cpp:138-144 + updateInternals :327-354.  `taps` holds ntaps
 * doubles (REAL) or ntaps (re,im) double pairs (COMPLEX).  ntaps == 0 -> PCX_ERR_ARG. */
PCX_API int pcx_fir_set_taps(pcx_fir *h, const double *taps, size_t ntaps);
/* setDecimation / setInterpolation, FIRFilter.cpp:151-168.  0 -> PCX_ERR_ARG. */
PCX_API int pcx_fir_set_decimation(pcx_fir *h, size_t decim);
PCX_API int pcx_fir_set_interpolation(pcx_fir *h, size_t interp);
PCX_API int pcx_fir_set_algo(pcx_fir *h, int algo);
/* the Q-format reading of THIS filter (integer element types; see pcx_qformat): the taps are quantised again with it and every
 * later call shifts and rounds by it.  NULL: the process-wide reading of pcx_set_qformat. */
PCX_API int pcx_fir_set_qformat(pcx_fir *h, const pcx_qformat *q);
/* K = ceil(ntaps/L) (FIRFilter.cpp:335) and _inputRequire = M+K-1 (:353) */
PCX_API int pcx_fir_get_geometry(const pcx_fir *h, size_t *K, size_t *input_require);
/* which algorithm the last process call ran (pcx_fir_algo) */
PCX_API int pcx_fir_last_algo(const pcx_fir *h);
/* How many of the device's 1024 resident workgroup slots the handle's persistent launches may take (a multiple of 128; default
 * 1024 = the whole device).  For handles that run SIDE BY SIDE on one device -- several shards of a stream on one GPU, two filter
 * blocks of one topology -- so that their launches share the device instead of queueing behind one another's workgroups
 * (two 32 Mi-sample launches: 0.2196 ms with 1024 each, 0.1988 with 512 each; one 64 Mi launch 0.1971). */
PCX_API int pcx_fir_set_slots(pcx_fir *h, unsigned slots);
/*
 * The filter loop, FIRFilter.cpp:278-308.  `in` points at the front of the input
 * buffer: in_elems elements of which the first K-1 are history (the reference's
 * `x = in + (K-1)`, :281).  out_cap = room in the output buffer, in elements.
 *   N         = min((in_elems-(K-1))/M, out_cap/L)*M          (:278)
 *   *consumed = N                                              (:307)
 *   *produced = (N/M)*L                                        (:308)
 * Burst flush (:263-272) is the caller's job: it passes the zero-padded buffer.
 */
PCX_API int pcx_fir_process(pcx_fir *h, const void *in, size_t in_elems, void *out, size_t out_cap,
                            size_t *consumed, size_t *produced);
PCX_API int pcx_fir_process_dev(pcx_fir *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                size_t *consumed, size_t *produced, void *stream);
/*
 * The same call over ONE SHARD of a stream that is split across devices: the first K-1 samples of `in_dev` -- the
 * history slot, FIRFilter.cpp:281,305-307 -- are the HALO, still on its way from the left neighbour when the call is queued.
 * ONE launch covers the whole shard: the blocks at the front, the only ones that read the halo, are computed last and not before
 * the 32-bit word *gate_dev has reached gate_value (signed distance, so a pass counter needs no reset).  The caller queues
 * pcx_gate_signal_dev(gate_dev, gate_value, s) on the stream that carries the halo, behind the transfer; everything else of the
 * shard is filtered while the halo is in flight.
  * The wait is bounded: after two seconds without the signal the held blocks run on whatever the halo slot holds and the word
 * behind the gate word (gate_dev[1]) is set to 0xDEAD -- a gate takes two 32-bit words.
 * ORDER OF QUEUEING.  Queue the transfer and the signal BEFORE this call (pcx_shard_step and the RCCL driver do): then nothing the
 * gate waits for can sit behind the gated launch.  A signal queued AFTER it must not share the launch's HARDWARE queue -- HIP
 * maps a process's streams onto four of them, round robin, and a packet waits for every earlier packet of its queue whatever
 * stream it came from: a signal behind the launch in the same queue waits for the launch, which waits for the signal (measured:
 * every fourth stream a process creates times out, tools/gate_queue_probe.py).  A stream of ANOTHER PRIORITY
 * (hipStreamCreateWithPriority) has queues of its own: use one for a signal that has to be queued late -- or put the gate words
 * in page-locked host memory (hipHostMalloc; the waiting workgroup reads them in place with system-scope loads) and open the gate
 * from the host with a plain 32-bit store, as the host-driven driver of pothoscomms_amd/stream.py does.
 *   *gated = 1: queued as described.   *gated = 0: this configuration has no gated kernel (anything but complex_float32 with
 *   M = L = 1 and K <= 2049, or a call of fewer than ~2048 blocks) and NOTHING has been queued: the caller waits for the halo on
 *   `stream` itself (an event) and calls pcx_fir_process_dev.
 */
PCX_API int pcx_fir_process_dev_gated(pcx_fir *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                      size_t *consumed, size_t *produced, const void *gate_dev, unsigned gate_value, void *stream,
                                      int *gated);
/* a one-thread kernel on `stream`: *gate_dev <- value (system-scope release) */
PCX_API int pcx_gate_signal_dev(void *gate_dev, unsigned value, void *stream);

/* ===================================================================== *
 *  /comms/fft             fft/FFT.cpp, fft/FFTAux.h, fft/kissfft.hh, fft/kiss_fft.c
 * ===================================================================== */
typedef struct pcx_fft pcx_fft;
/* FFTFactory(dtype, numBins, inverse), FFT.cpp:83-93: scalar in {F64, F32, I16}
 * (always complex).  Forward = exp(-j..), inverse = exp(+j..), float paths
 * unscaled (kissfft.hh:81-161), int16 path scaled by 1/radix per stage
 * (kiss_fft.c:61, _kiss_fft_guts.h:73-78). */
PCX_API int pcx_fft_create(int scalar, size_t num_bins, int inverse, pcx_fft **out);
PCX_API int pcx_fft_destroy(pcx_fft *h);
/* FFTAux::transform over `nframes` back-to-back frames (FFT::work does one,
 * FFT.cpp:66-71; a device caller batches whole frames per call). */
PCX_API int pcx_fft_transform(pcx_fft *h, const void *in, void *out, size_t nframes);
PCX_API int pcx_fft_transform_dev(pcx_fft *h, const void *in_dev, void *out_dev, size_t nframes, void *stream);

/* ===================================================================== *
 *  /comms/freq_demod      demod/FreqDemod.cpp
 * ===================================================================== */
typedef struct pcx_freqdemod pcx_freqdemod;
/* FreqDemodFactory(dtype), FreqDemod.cpp:80-93: complex<scalar> in, scalar out */
PCX_API int pcx_freqdemod_create(int scalar, pcx_freqdemod **out);
PCX_API int pcx_freqdemod_destroy(pcx_freqdemod *h);
/* activate(): _prev = 0, FreqDemod.cpp:44-47 */
PCX_API int pcx_freqdemod_reset(pcx_freqdemod *h);
/* the loop FreqDemod.cpp:60-67: out[i] = angle(in[i]*prev); prev = conj(in[i]);
 * prev is carried across calls inside the handle (device side) */
PCX_API int pcx_freqdemod_process(pcx_freqdemod *h, const void *in, void *out, size_t n);
PCX_API int pcx_freqdemod_process_dev(pcx_freqdemod *h, const void *in_dev, void *out_dev, size_t n, void *stream);

/* ===================================================================== *
 *  /comms/rotate, /comms/scale, /comms/abs, /comms/conjugate   (math/)
 *  Stateless maps; n counts stream elements times dtype.dimension().
 * ===================================================================== */
/* arrayRotate, Rotate.cpp:15-23.  (phasor_re, phasor_im) is std::polar(1.0, phase)
 * BEFORE floatToQ (Rotate.cpp:74); pass (0,0) for a block whose setPhase was never
 * called (value-initialised _phasor).  Complex element types only (:143-157). */
PCX_API int pcx_rotate(int scalar, double phasor_re, double phasor_im, const void *in, void *out, size_t n);
PCX_API int pcx_rotate_dev(int scalar, double phasor_re, double phasor_im, const void *in_dev, void *out_dev, size_t n, void *stream);
/* arrayScale, Scale.cpp:15-23 with factorScaled = floatToQ(factor) (:70-74); real factor */
PCX_API int pcx_scale(int scalar, int is_complex, double factor, const void *in, void *out, size_t n);
PCX_API int pcx_scale_dev(int scalar, int is_complex, double factor, const void *in_dev, void *out_dev, size_t n, void *stream);
/* the same two maps under an explicit Q-format reading (integer element types; q == NULL: the process-wide one) */
PCX_API int pcx_rotate_q(int scalar, double phasor_re, double phasor_im, const pcx_qformat *q, const void *in, void *out, size_t n);
PCX_API int pcx_rotate_q_dev(int scalar, double phasor_re, double phasor_im, const pcx_qformat *q, const void *in_dev, void *out_dev, size_t n,
                             void *stream);
PCX_API int pcx_scale_q(int scalar, int is_complex, double factor, const pcx_qformat *q, const void *in, void *out, size_t n);
PCX_API int pcx_scale_q_dev(int scalar, int is_complex, double factor, const pcx_qformat *q, const void *in_dev, void *out_dev, size_t n,
                            void *stream);
/* Abs.cpp:40-43 via getAbs, FxptHelpers.hpp:36-49; out is the real scalar type */
PCX_API int pcx_abs(int scalar, int is_complex, const void *in, void *out, size_t n);
PCX_API int pcx_abs_dev(int scalar, int is_complex, const void *in_dev, void *out_dev, size_t n, void *stream);
/* Conjugate.cpp:36-39; complex element types only */
PCX_API int pcx_conj(int scalar, const void *in, void *out, size_t n);
PCX_API int pcx_conj_dev(int scalar, const void *in_dev, void *out_dev, size_t n, void *stream);

/* /comms/angle (SURVEY 8f "next"): math/Angle.cpp:23-26 via getAngle, FxptHelpers.hpp:14-29;
 * complex element types only, out is the real scalar type */
PCX_API int pcx_angle(int scalar, const void *in, void *out, size_t n);
PCX_API int pcx_angle_dev(int scalar, const void *in_dev, void *out_dev, size_t n, void *stream);

/* /comms/arithmetic (SURVEY 8f "next"): math/Arithmetic.cpp:70-110, out[i] = in0[i] OP in1[i] with the
 * C++ operator of the element type (real or std::complex of f64, f32, (u)int8..64).  The block's
 * work() (:205-231) left-folds its N input ports: call once per extra port with in0 = out.  `out`
 * may be exactly in0 or in1 (the reference forwards input 0's buffer, :157-158).  Integer x / 0
 * traps in the reference; here it yields 0.  Unknown op or type: PCX_ERR_ARG ("unsupported args", :297). */
typedef enum pcx_arith_op { PCX_ARITH_ADD = 0, PCX_ARITH_SUB = 1, PCX_ARITH_MUL = 2, PCX_ARITH_DIV = 3 } pcx_arith_op;
PCX_API int pcx_arith(int scalar, int is_complex, int op, const void *in0, const void *in1, void *out, size_t n);
PCX_API int pcx_arith_dev(int scalar, int is_complex, int op, const void *in0_dev, const void *in1_dev, void *out_dev, size_t n, void *stream);
/* /comms/split_complex, /comms/combine_complex: utility/SplitComplex.cpp:10-18, utility/CombineComplex.cpp:10-17;
 * scalar = the real type of the planes (f64, f32, int64..int8: splitComplexFactory :60-70) */
PCX_API int pcx_split_complex(int scalar, const void *in, void *re, void *im, size_t n);
PCX_API int pcx_split_complex_dev(int scalar, const void *in_dev, void *re_dev, void *im_dev, size_t n, void *stream);
PCX_API int pcx_combine_complex(int scalar, const void *re, const void *im, void *out, size_t n);
PCX_API int pcx_combine_complex_dev(int scalar, const void *re_dev, const void *im_dev, void *out_dev, size_t n, void *stream);

/* ===================================================================== *
 *  Fused FM-demod chain  Rotate -> FIR -> FreqDemod in one kernel
 *  (BASELINE.json configs[4]); equals the three blocks above connected in a
 *  topology: Rotate.cpp:15-23 -> FIRFilter.cpp:286-302 (M=L=1) -> FreqDemod.cpp:60-67
 *  complex_float32 in, float32 out.
 * ===================================================================== */
typedef struct pcx_fmchain pcx_fmchain;
PCX_API int pcx_fmchain_create(pcx_fmchain **out);
PCX_API int pcx_fmchain_destroy(pcx_fmchain *h);
PCX_API int pcx_fmchain_set_phase(pcx_fmchain *h, double phase);
/* REAL (complex_taps=0) or COMPLEX taps, as pcx_fir_set_taps */
PCX_API int pcx_fmchain_set_taps(pcx_fmchain *h, const double *taps, size_t ntaps, int complex_taps);
PCX_API int pcx_fmchain_reset(pcx_fmchain *h);
/* PCX_FIR_AUTO (default), PCX_FIR_DIRECT (LDS-tiled time domain) or PCX_FIR_OLS_FFT (K <= 2048) */
PCX_API int pcx_fmchain_set_algo(pcx_fmchain *h, int algo);
PCX_API int pcx_fmchain_last_algo(const pcx_fmchain *h);
PCX_API int pcx_fmchain_set_slots(pcx_fmchain *h, unsigned slots);      /* as pcx_fir_set_slots */
/* in_elems input samples with K-1 history in front -> in_elems-(K-1) demodulated
 * outputs; FreqDemod's prev is carried in the handle */
PCX_API int pcx_fmchain_process(pcx_fmchain *h, const void *in, size_t in_elems, void *out, size_t out_cap,
                                size_t *consumed, size_t *produced);
PCX_API int pcx_fmchain_process_dev(pcx_fmchain *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                    size_t *consumed, size_t *produced, void *stream);
/* one shard of a sharded chain in ONE launch, as pcx_fir_process_dev_gated: the halo in front of the shard is K samples here
 * (the FIR's K-1 and the one FreqDemod's `_prev` needs, FreqDemod.cpp:63-65); gated for K <= 2048 and long calls */
PCX_API int pcx_fmchain_process_dev_gated(pcx_fmchain *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                          size_t *consumed, size_t *produced, const void *gate_dev, unsigned gate_value, void *stream,
                                          int *gated);

/* ===================================================================== *
 *  ONE complex_float32 stream over the GPUs of a node  (BASELINE.json configs[3])
 *
 *  The reference has no multi-device story; what makes this possible is in its loop: output n reads inputs
 *  n .. n+K-1 only (FIRFilter.cpp:296-299) and a work() call leaves the last K-1 inputs un-consumed as the next
 *  call's history (:305-307).  A stream of G*C samples is split into G contiguous shards of C samples, one per
 *  device; before every pass shard g receives the LAST K-1 samples of shard g-1 into the slot in front of its own
 *  samples (RCCL ncclSend/ncclRecv, one group per pass, (K-1)*8 bytes per boundary), shard 0 keeps the stream's
 *  own history.  Each shard is ONE kernel launch per pass: everything behind its first block is filtered while the halo is in
 *  flight, the first block last, behind a gate word the halo stream sets (pcx_fir_process_dev_gated).
 *  One process drives all devices (ncclCommInitAll): a Pothos block that owns a pcx_shard spreads its stream over
 *  the node from inside one work() call.  M = L = 1, complex_float32 (the north-star path).
 *
 *  Call order:  create -> set_taps -> configure(C) -> { fill the shard inputs (pcx_shard_buffers gives the device
 *  pointers and the per-device stream to fill them on; or pcx_shard_scatter from one host buffer) -> step }* ->
 *  gather / read the outputs -> destroy.  pcx_shard_step only enqueues; pcx_shard_sync waits.
 * ===================================================================== */
typedef struct pcx_shard pcx_shard;
typedef enum pcx_shard_transport {
    PCX_SHARD_RCCL = 0,      /* ncclSend/ncclRecv over xGMI; one distinct device per shard (RCCL is loaded on first use).  What one GPU could
                              * show about its cost: RCCL's send/recv kernel needs a slot on the device, and beside a gated launch that fills
                              * it (128-VGPR workgroups, RCCL's waves are allocated 136) it gets one only when that launch's first workgroups
                              * exit -- the pass then ends ~18 us late (+9 %; DESIGN.md 6, profiles/r04/rccl_cost_probe.txt).  The rank driver
                              * (pothoscomms_amd/stream.py) hides that by exchanging the NEXT batch's halo; this driver does not. */
    PCX_SHARD_PEER_COPY = 1  /* hipMemcpyPeerAsync between the devices (peer access is switched on where the devices allow it): no kernel
                              * that must find a slot -- two shards on one device cost +1.2 % over one launch.  Several shards may share a
                              * device (how a 1-GPU box rehearses G > 1).  bench.py --driver native --native-transport peer|rccl compares. */
} pcx_shard_transport;
/* devices: `nshards` ordinals, or NULL for 0 .. nshards-1 */
PCX_API int pcx_shard_create(int nshards, const int *devices, int transport, pcx_shard **out);
PCX_API int pcx_shard_destroy(pcx_shard *s);
/* FIRFilter::setTaps on every device's filter, FIRFilter.cpp:138-144; REAL (complex_taps = 0) or COMPLEX taps */
PCX_API int pcx_shard_set_taps(pcx_shard *s, const double *taps, size_t ntaps, int complex_taps);
PCX_API int pcx_shard_set_algo(pcx_shard *s, int algo);                 /* pcx_fir_algo, default PCX_FIR_AUTO */
/* enable != 0: the shards run the fused chain Rotate(phase) -> FIR -> FreqDemod (pcx_fmchain, BASELINE configs[4]) instead of the
 * FIR alone: float32 outputs, a halo of K samples (the FIR's K-1 and FreqDemod's one, FreqDemod.cpp:63-65), every pass from the
 * reset state (the first output of the stream is arg of a zero, FreqDemod.cpp:44-47).  Call before pcx_shard_configure. */
PCX_API int pcx_shard_set_chain(pcx_shard *s, int enable, double phase);
/* allocate, on every device, [halo (K-1) | shard_elems samples] and shard_elems outputs; the halo of shard 0 (the
 * stream's history) starts as zeros, as after FIRFilter::activate */
PCX_API int pcx_shard_configure(pcx_shard *s, size_t shard_elems);
PCX_API int pcx_shard_info(const pcx_shard *s, int *nshards, size_t *K, size_t *shard_elems, int *transport);
/* shard g: in_dev -> K-1 halo samples followed by the shard's own samples; out_dev -> its shard_elems outputs;
 * stream -> the per-device stream (hipStream_t) the pass runs on: fill in_dev on it, or synchronise before a step */
PCX_API int pcx_shard_buffers(pcx_shard *s, int g, void **in_dev, void **out_dev, void **stream, int *device);
/* host_stream: K-1 history samples followed by nshards*shard_elems samples; only shard 0 receives a halo from here */
PCX_API int pcx_shard_scatter(pcx_shard *s, const void *host_stream, size_t elems);
/* one pass over every shard: the halo exchange and ONE launch per shard (pcx_fir_process_dev_gated); configurations without a
 * gated kernel run body, exchange, head as two launches */
PCX_API int pcx_shard_step(pcx_shard *s);
/* pcx_shard_step in its two halves -- the exchange of the halos of what the shard buffers hold NOW (+ the gate signals behind it), and
 * the pass over it -- for DOUBLE-BUFFERED streaming over two handles A and B on the same devices:
 *     fill B (batch k+1);  pcx_shard_compute(A)  [batch k, its exchange posted one turn earlier];  pcx_shard_post_exchange(B);  swap
 * so that the halos of batch k+1 travel while batch k is filtered.  It matters for the RCCL transport, whose send/recv kernel finds a
 * slot beside a gated launch only when that launch's first workgroups exit (PCX_SHARD_RCCL above): a pass that waits for its OWN exchange
 * ends ~18 us late, a pass whose exchange was posted a turn earlier does not (measured for the rank driver, pothoscomms_amd/stream.py
 * PingPongFir: +4 % over the plain launch instead of +9-11 %).  Each handle has its own streams, so the host may queue compute(A) first.
 * After pcx_shard_post_exchange the shard buffers of that handle must not be written until its pcx_shard_compute has been queued (the
 * exchange is reading their tails; compute orders later writers behind it).  compute without a posted exchange, a second post
 * without a compute between, and pcx_shard_scatter / pcx_shard_configure / pcx_shard_set_taps / pcx_shard_set_chain / pcx_shard_set_algo
 * on a handle whose exchange is posted are PCX_ERR_STATE (the setters would change, or free, what the posted pass is about to use). */
PCX_API int pcx_shard_post_exchange(pcx_shard *s);
PCX_API int pcx_shard_compute(pcx_shard *s);
/* enable != 0: one SUBMIT THREAD per DEVICE.  Queueing a pass costs the host 16-20 us per shard from one thread (the cross-stream waits,
 * the records, the gate signal, the launch): 135-165 us for eight shards against a pass of 195 us at 64 Mi samples per shard.  With
 * submit threads every device's share of a pass is queued by a thread of its own, bound to that device (several shards on one device:
 * one thread, in shard order -- threads that call into ONE device's runtime only queue behind its locks); pcx_shard_step /
 * post_exchange / compute still return when everything is queued, and everything the header says about ordering holds unchanged.  The
 * threads spin for ~0.4 ms behind a pass (a stream of passes finds them awake) and sleep after that; they are joined by
 * pcx_shard_destroy or by enable = 0.  Off by default.  PCX_ERR_STATE while an exchange is posted. */
PCX_API int pcx_shard_set_submit_threads(pcx_shard *s, int enable);
/* enable = 0: every shard as TWO launches per pass -- the body while the halo is in flight, the head behind an event on the halo
 * stream -- instead of one gated launch (the default, enable = 1).  The gated launch relies on the halo transfer and its signal,
 * which pcx_shard_step queues BEFORE the launch, reaching the device before it: that is how the runtime submits (in order, from the
 * calling thread) in its default mode.  With AMD_DIRECT_DISPATCH=0 every stream is submitted by a thread of its own and a signal can
 * land behind the launch it is to release, in the same hardware queue: the gate then opens on its two-second bound only and
 * pcx_shard_gather / pcx_shard_sync report PCX_ERR_STATE (measured: intermittently, one test run in three on this stack; never in the
 * default mode, nor with GPU_MAX_HW_QUEUES=1 / 8 or HSA_ENABLE_SDMA=0).  A process that has
 * to run in that mode switches the gate off; the two-launch form costs 5-14 % of a pass. */
PCX_API int pcx_shard_set_gated(pcx_shard *s, int enable);
/* the nshards*shard_elems outputs in stream order (waits for the pass).  PCX_ERR_STATE when a shard's gated launch gave up
 * waiting for its halo during the passes since the last gather / sync (the two-second bound of pcx_fir_process_dev_gated): the
 * outputs are copied all the same, the seam's are wrong, and the condition is cleared by being reported. */
PCX_API int pcx_shard_gather(pcx_shard *s, void *host_out, size_t elems);
/* waits for everything queued on the shards' streams; reports (and clears) a gate timeout like pcx_shard_gather */
PCX_API int pcx_shard_sync(pcx_shard *s);

#ifdef __cplusplus
}
#endif
#endif /* PCX_H */
