// pcx_sched.hpp -- dynamic block assignment for the persistent-workgroup kernels.
//
// Why: a launch of fir_cf32_ols4096_kernel used to give every workgroup the same number of blocks (grid stride).  The
// CUs do not run at the same rate (XCDs differ in their distance to the HBM stacks, clocks move under the power cap), so
// the launch ended when the slowest workgroup did while the others idled: measured on MI355X (tools/ols_lab.hip,
// profiles/r02/ols_lab.md) 0.2164 ms per 64 Mi samples with the static stride, 0.2062 with a four-fold oversubscribed
// grid (the hardware dispatcher balancing), 0.2025 with the scheme below.
//
// Scheme: the blocks of a launch are dealt in chunks of kChunk = 2 (blocks q and q + nchunks: the strided pairing measured
// faster than contiguous pairs).  A workgroup draws its next chunk with ONE atomic on a device-wide counter -- a single
// word serves about 88 draws per microsecond, one draw per block (80 per microsecond at the headline rate) saturated it
// and ran 10 % SLOWER than the static stride -- issued right behind the loads of the chunk's last block so the in-order
// vmcnt queue lets the loads be waited for without it, and consumed at the end of that block.  The value travels from
// lane 0 to the workgroup through one LDS word written in front of a barrier the block's pipeline already has.
//
// Books: no host-side reset.  The counter pair {draws, finished workgroups} lives in the handle (8 bytes, zeroed once).
// Every workgroup ends on exactly one draw past the end and then bumps `finished`; the workgroup that finds itself last
// zeroes both words for the next launch (launches of one handle are stream-ordered, pcx_api.hip ctx_enter).
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {

constexpr unsigned kSchedChunk = 2;

struct SchedState {   // device memory, owned by the handle
    unsigned draws;
    unsigned finished;
};

// one per workgroup; `slot` is a __shared__ unsigned of the kernel
struct BlockDealer {
    SchedState *st;
    unsigned *slot;
    unsigned nchunks, chunk, sub, pending;
    size_t nblocks;

    // first chunk (all lanes call; contains one barrier).  Returns false when there is nothing for this workgroup.
    __device__ __forceinline__ bool begin(SchedState *state, unsigned *lds_slot, size_t nblocks_, int lane)
    {
        st = state; slot = lds_slot; nblocks = nblocks_;
        nchunks = (unsigned)((nblocks_ + kSchedChunk - 1) / kSchedChunk);
        sub = 0; pending = 0;
        if (lane == 0) *slot = atomicAdd(&st->draws, 1u);
        __syncthreads();
        chunk = *slot;
        return chunk < nchunks;
    }
    __device__ __forceinline__ size_t block() const { return (size_t)chunk + (size_t)sub * nchunks; }
    __device__ __forceinline__ bool last_of_chunk() const { return sub + 1 >= kSchedChunk || (size_t)chunk + (size_t)(sub + 1) * nchunks >= nblocks; }
    // behind the loads of the current block: start the draw for the next chunk (lane 0, last block of a chunk only)
    __device__ __forceinline__ void draw(int lane)
    {
        if (lane == 0 && last_of_chunk()) pending = atomicAdd(&st->draws, 1u);
    }
    // somewhere later in the block, IN FRONT of a barrier every lane passes before advance(): publish the draw
    __device__ __forceinline__ void publish(int lane)
    {
        if (lane == 0 && last_of_chunk()) *slot = pending;
    }
    // end of the block (all lanes): false when the workgroup is done
    __device__ __forceinline__ bool advance()
    {
        if (!last_of_chunk()) { sub++; return true; }
        chunk = *slot;
        sub = 0;
        return chunk < nchunks;
    }
    // every exit path of the kernel (all lanes call, lane 0 acts): the last workgroup of the launch resets the books
    __device__ __forceinline__ void finish(int lane)
    {
        if (lane == 0 && atomicAdd(&st->finished, 1u) == gridDim.x - 1) {
            atomicExch(&st->draws, 0u);
            atomicExch(&st->finished, 0u);
        }
    }
};

// One spelling for both walks, so a kernel is written once: BlockWalk<false> is the grid stride, BlockWalk<true> the dealer.
//     BlockWalk<DYN> walk;
//     if (!walk.begin(sched, &slot, nblocks, lane)) { walk.finish(lane); return; }      // FIRST: begin() holds a barrier
//     for (;;) { b = walk.block(); loads; walk.draw(lane); ...; walk.publish(lane); <a barrier>; ...; if (!walk.advance()) break; }
//     walk.finish(lane);
template <bool DYN>
struct BlockWalk;
template <>
struct BlockWalk<true> : BlockDealer {};
template <>
struct BlockWalk<false> {
    size_t b, n;
    __device__ __forceinline__ bool begin(SchedState *, unsigned *, size_t nblocks_, int) { b = blockIdx.x; n = nblocks_; return b < n; }
    __device__ __forceinline__ size_t block() const { return b; }
    __device__ __forceinline__ void draw(int) {}
    __device__ __forceinline__ void publish(int) {}
    __device__ __forceinline__ bool advance() { b += gridDim.x; return b < n; }
    __device__ __forceinline__ void finish(int) {}
};

// The same dealing for a kernel that prefetches the NEXT block's input while it works on the current one (fft4096_kernel, which
// the product no longer runs -- fft.hip: the plain family kernel measured faster -- and the diagnostic library keeps for that A/B):
// the next block's index must be known at the START of a block, so the draw for chunk c+1 is issued during the FIRST
// block of chunk c, published in front of one of that block's barriers and read at the start of the chunk's second block.
// (Only the launch's last chunk can be a single block, and nothing follows it.)
struct AheadDealer {
    SchedState *st;
    unsigned *slot;
    unsigned nchunks, chunk, sub, pending, next_chunk;
    size_t nblocks;

    __device__ __forceinline__ bool begin(SchedState *state, unsigned *lds_slot, size_t nblocks_, int lane)
    {
        st = state; slot = lds_slot; nblocks = nblocks_;
        nchunks = (unsigned)((nblocks_ + kSchedChunk - 1) / kSchedChunk);
        sub = 0; pending = 0; next_chunk = ~0u;
        if (lane == 0) *slot = atomicAdd(&st->draws, 1u);
        __syncthreads();
        chunk = *slot;
        __syncthreads();   // the slot is rewritten during the first block
        return chunk < nchunks;
    }
    __device__ __forceinline__ bool pair() const { return (size_t)chunk + nchunks < nblocks; }   // the chunk holds two blocks
    __device__ __forceinline__ size_t block() const { return (size_t)chunk + (size_t)sub * nchunks; }
    // start of a block (all lanes): is there a block after this one, and which
    __device__ __forceinline__ bool next(size_t *nb)
    {
        if (sub == 0 && pair()) { *nb = (size_t)chunk + nchunks; return true; }
        next_chunk = pair() ? *slot : ~0u;      // second block of a pair: the draw published during the first
        if (next_chunk < nchunks) { *nb = next_chunk; return true; }
        return false;
    }
    __device__ __forceinline__ void draw(int lane) { if (lane == 0 && sub == 0 && pair()) pending = atomicAdd(&st->draws, 1u); }
    __device__ __forceinline__ void publish(int lane) { if (lane == 0 && sub == 0 && pair()) *slot = pending; }
    __device__ __forceinline__ bool advance()
    {
        if (sub == 0 && pair()) { sub = 1; return true; }
        chunk = next_chunk;
        sub = 0;
        return chunk < nchunks;
    }
    __device__ __forceinline__ void finish(int lane)
    {
        if (lane == 0 && atomicAdd(&st->finished, 1u) == gridDim.x - 1) {
            atomicExch(&st->draws, 0u);
            atomicExch(&st->finished, 0u);
        }
    }
};

}  // namespace pcx
