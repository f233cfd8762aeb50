/*
 * pcx_blocks.h -- runner ABI of libpcx_blocks.so: drives the MI355X-backed /comms blocks
 * (pothoscomms_amd/csrc/blocks/comms_blocks.cpp) the way the Pothos scheduler would, for hosts
 * without PothosCore (the test-suite, language bindings).  Inside a real Pothos install the
 * blocks are loaded as a plugin module instead and this ABI is not used.
 *
 * One call = one scheduler action on one block: make (BlockRegistry::make), call a registered
 * setter/getter by name (Proxy call), activate(), one work() on planted port buffers followed
 * by propagateLabels().  Buffers are HOST memory, exactly like Pothos BufferChunks.
 */
#ifndef PCX_BLOCKS_H
#define PCX_BLOCKS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCXB_API __attribute__((visibility("default")))

typedef struct pcxb_block pcxb_block;

/* Label::data payload kinds */
enum { PCXB_NONE = 0, PCXB_SIZE = 1, PCXB_DOUBLE = 2, PCXB_STRING = 3 };
typedef struct pcxb_label {
    char id[32];       /* Label::id */
    uint64_t index;    /* Label::index */
    uint64_t width;    /* Label::width */
    int kind;          /* which of the fields below holds Label::data */
    uint64_t uval;
    double dval;
    char sval[32];
} pcxb_label;

PCXB_API const char *pcxb_last_error(void);
/* registry: 1 if a factory is registered under `path` (e.g. "/comms/fir_filter") */
PCXB_API int pcxb_registry_has(const char *path);
PCXB_API size_t pcxb_registry_count(void);
PCXB_API const char *pcxb_registry_path(size_t i);
/* how many arguments the factory registered at `path` takes; -1: no such path */
PCXB_API long pcxb_registry_arity(const char *path);

/* BlockRegistry::make(path, dtype, ...).  dtype is a Pothos DType name ("complex_float32"),
 * dimension its vector dimension.  The remaining factory arguments by block:
 *   fir_filter: sarg = tapsType ("REAL"/"COMPLEX");  arithmetic: sarg = operation ("ADD"/"SUB"/"MUL"/"DIV");
 *   fft: num_bins, inverse;  others: none */
PCXB_API int pcxb_make(const char *path, const char *dtype, size_t dimension, const char *sarg, size_t num_bins,
                       int inverse, pcxb_block **out);
PCXB_API int pcxb_destroy(pcxb_block *b);

/* registered calls (registerCall names): one argument of the given kind, or a getter */
PCXB_API int pcxb_call_double(pcxb_block *b, const char *name, double v);
PCXB_API int pcxb_call_size(pcxb_block *b, const char *name, size_t v);
PCXB_API int pcxb_call_bool(pcxb_block *b, const char *name, int v);
PCXB_API int pcxb_call_string(pcxb_block *b, const char *name, const char *v);
PCXB_API int pcxb_call_taps(pcxb_block *b, const char *name, const double *taps, size_t n, int is_complex);
PCXB_API int pcxb_call_sizes(pcxb_block *b, const char *name, const size_t *v, size_t n);   /* std::vector<size_t> (setPreload) */
PCXB_API int pcxb_get_sizes(pcxb_block *b, const char *name, size_t *out, size_t cap, size_t *n);
PCXB_API int pcxb_get_double(pcxb_block *b, const char *name, double *out);
PCXB_API int pcxb_get_size(pcxb_block *b, const char *name, size_t *out);
PCXB_API int pcxb_get_bool(pcxb_block *b, const char *name, int *out);
PCXB_API int pcxb_get_string(pcxb_block *b, const char *name, char *out, size_t cap);
PCXB_API int pcxb_get_taps(pcxb_block *b, const char *name, double *out, size_t cap_doubles, size_t *n, int is_complex);

/* the calls the block registered (registerCall): their number, the i-th name, and how many arguments the named one takes (-1: none) */
PCXB_API int pcxb_call_count(pcxb_block *b, size_t *count);
PCXB_API int pcxb_call_name(pcxb_block *b, size_t i, char *out, size_t cap);
PCXB_API long pcxb_call_arity(pcxb_block *b, const char *name);
PCXB_API int pcxb_activate(pcxb_block *b);
PCXB_API int pcxb_deactivate(pcxb_block *b);
/* Topology::connect(src, "signal", dst, "slot") for signal -> setter wiring (the designer's "tapsChanged" ->
 * /comms/fir_filter "setTaps", filter/TestFIRFilter.cpp:48): every later emission calls the slot synchronously.
 * dst must outlive src's emissions. */
PCXB_API int pcxb_connect_signal(pcxb_block *src, const char *signal, pcxb_block *dst, const char *slot);
/* the first input / output port's type (indexed port 0, or the first named port) and the buffer managers the block requests */
PCXB_API int pcxb_port_dtype(pcxb_block *b, int is_output, char *name, size_t cap, size_t *dimension, size_t *bytes);
PCXB_API int pcxb_buffer_manager(pcxb_block *b, int is_output, char *name, size_t cap, size_t *buffer_size);
/* What the scheduler does with the manager a block returns: draw the next port buffer from it (round-robin over the
 * manager's slabs; a slab is at least max(min_bytes, the manager's bufferSize) long).  The device-backed blocks hand
 * out page-locked slabs (*pinned = 1), on which the C ABI runs its kernels directly -- a work() loop over these
 * buffers pays no staging copy.  The memory belongs to the block and lives until pcxb_destroy. */
PCXB_API int pcxb_acquire_buffer(pcxb_block *b, int is_output, size_t min_bytes, void **ptr, size_t *bytes, int *pinned);
/* The same for an EDGE src.output(0) -> dst.input(0), with the negotiation a scheduler does: dst is asked first with src's port
 * domain, then src with dst's; two blocks of this module share the domain "pcx-hip" and get slabs in DEVICE memory (*kind = 2:
 * host code must not touch them; 1 = page-locked host, 0 = pageable), so samples flowing between them never cross PCIe.  The
 * memory belongs to dst and lives until pcxb_destroy(dst). */
PCXB_API int pcxb_link_buffer(pcxb_block *src, pcxb_block *dst, size_t min_bytes, void **ptr, size_t *bytes, int *kind);
/* The FRAMEWORK's "circular" buffer (what Pothos hands a block that asks for BufferManager::make("circular"), as the reference
 * FIR does, filter/FIRFilter.cpp:196-199): `bytes` (rounded up to whole pages, *actual) of PAGEABLE shared memory mapped twice back
 * to back, so that base[i] and base[*actual + i] are the same byte and a window of up to *actual bytes may start anywhere in the
 * first mapping.  The stand-in for that manager on this side of the boundary: the blocks see it exactly as they see Pothos's --
 * ordinary host memory they did not allocate -- and /comms/fir_filter page-locks it where it lies on first sight
 * (pcx_host_register_mapping).  Destroy it AFTER the blocks that saw it. */
PCXB_API int pcxb_circular_create(size_t bytes, void **base, size_t *actual);
PCXB_API int pcxb_circular_destroy(void *base);
/* the reserve a block asked for at construction time (FFT: numBins); SIZE_MAX = none */
PCXB_API int pcxb_initial_reserve(pcxb_block *b, size_t *reserve);

/*
 * One work() call: plant `in` (in_elems elements, labels attached) and `out` (room for
 * out_elems), set workInfo().minElements = min(in_elems, out_elems), call work(), then
 * propagateLabels() on the labels inside the consumed region.
 *   *reserve = value passed to setReserve during this call, SIZE_MAX if it was not called
 *   posted labels (at most cap) are copied to `posted`, *nposted = their number
 */
PCXB_API int pcxb_work(pcxb_block *b, const void *in, size_t in_elems, const pcxb_label *labels, size_t nlabels,
                       void *out, size_t out_elems, size_t *consumed, size_t *produced, size_t *reserve,
                       pcxb_label *posted, size_t cap, size_t *nposted);

/* Measurement aid: `reps` work() calls on the SAME planted buffers back to back from native code (no labels), wall seconds of the
 * loop in *seconds, the last call's consume / produce in *consumed / *produced -- what a C++ scheduler's loop pays per call, without
 * the per-call cost of a scripting-language binding around pcxb_work. */
PCXB_API int pcxb_work_loop(pcxb_block *b, const void *in, size_t in_elems, void *out, size_t out_elems, size_t reps, double *seconds,
                            size_t *consumed, size_t *produced);

/*
 * Blocks with several ports (arithmetic: N indexed inputs; split_complex: outputs "re","im";
 * combine_complex: inputs "re","im").  Ports are numbered indexed-first, then the named ones in
 * the order the block declared them.  `preloaded` = elements queued on an input by activate()
 * (Arithmetic's feedback preload): the host prepends that many zero elements to the port's stream.
 */
PCXB_API int pcxb_num_ports(pcxb_block *b, int is_output, size_t *count);
PCXB_API int pcxb_port_info(pcxb_block *b, int is_output, size_t i, char *name, size_t name_cap, char *dtype, size_t dtype_cap,
                            size_t *dimension, size_t *bytes, size_t *preloaded);
/* one work() call on planted buffers for every port: workInfo().minElements = minimum over the
 * indexed ports, minAllElements over all of them; no labels on this path */
PCXB_API int pcxb_work_ports(pcxb_block *b, size_t nin, const void *const *ins, const size_t *in_elems, size_t nout,
                             void *const *outs, const size_t *out_elems, size_t *consumed, size_t *produced);

#ifdef __cplusplus
}
#endif
#endif /* PCX_BLOCKS_H */
