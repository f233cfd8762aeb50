// pcx_sched.hpp -- dynamic block assignment for the persistent-workgroup kernels.
//
// Why: a launch of fir_cf32_ols4096_kernel used to give every workgroup the same number of blocks (grid stride).  The
// CUs do not run at the same rate (XCDs differ in their distance to the HBM stacks, clocks move under the power cap), so
// the launch ended when the slowest workgroup did while the others idled.  Round 2 dealt the blocks from ONE counter word,
// two blocks per draw because a single word serves only ~88 draws per microsecond -- exactly one draw per block at the
// headline rate.  What that left on the table, measured with per-workgroup time stamps (tools/ols_lab3.hip,
// profiles/r03/ols_lab3.md):
//   * the FIRST draw of the 1024 workgroups took 6 us (median) to drain through the one word -- 3 % of a 200 us launch
//     spent before the first load;
//   * dealing pairs leaves the end of the launch ragged by two blocks: mean idle time of a workgroup behind the last
//     block 11-13 us;
//   * `atomicAdd` inside `if (lane == 0)` is rewritten by the compiler's atomic optimiser into add + s_waitcnt vmcnt(0) +
//     v_readfirstlane: wave 0 stalled on the whole round trip, queued behind the block's 16 loads, in every drawing block
//     (and a lane-0-only atomic between the loads and their first use makes the compiler wait for it together with the last
//     load: it cannot know whether the branch issued it).
//
// Scheme now:
//   * the first block of workgroup w is block w: no atomic before the first load;
//   * kShards = 16 counters, 128 bytes apart.  Workgroup w draws from counter g = (w >> 3) & 15 -- under the round-robin
//     dispatch that is 2 CUs of every XCD, all four slots of each -- whose n-th draw is block  gridDim + 16 n + g.  No word
//     sees more than 1/16 of the draws, so every block is drawn singly and the launch's end is ragged by one block, not two;
//   * a workgroup whose counter has run dry ends.  (Taking over the neighbouring counter once, so that the sixteen groups
//     cannot drift apart, measured the same -- 0.1963 vs 0.1965 ms, every group holds the same 64 workgroups on 16 CUs spread
//     over all XCDs -- and cost the fused and the real-stream kernels registers they do not have; left out.)
//   * the draw is issued behind the first butterflies of a block, when none of the block's loads is outstanding, and is left
//     in flight until publish() parks its value in LDS in front of the inverse transform's barriers (the file is built with
//     the atomic optimiser off, csrc/Makefile; the pipeline's barriers order LDS only, fft4096.hpp lds_barrier, so none of
//     them drains the atomic either).
// Same box, 64 Mi samples, 255 taps (profiles/r03/ols_lab3_*.txt): 0.2019-0.2026 ms (r02) -> 0.1973-0.1982 (static first
// block) -> 0.1963-0.1971 (+ sixteen counters, single blocks).
//
// Books: no host-side reset.  The state lives in the handle (kSchedBytes, zeroed once).  Every workgroup bumps `finished`
// on exit; the workgroup that finds itself last zeroes the counters for the next launch (launches of one handle are
// stream-ordered, pcx_api.hip ctx_enter).
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {

constexpr unsigned kShards = 16;
constexpr unsigned kShardStride = 32;   // words: one 128-byte line per counter

struct SchedState {   // device memory, owned by the handle
    unsigned ctr[kShards * kShardStride];
    unsigned finished;
};
static_assert(sizeof(SchedState) <= 4096, "a handle allocates kSchedBytes = 4096 (pcx_internal.hpp)");

// one per workgroup; `slot` is a __shared__ unsigned of the kernel
struct BlockDealer {
    SchedState *st;
    unsigned *slot;
    unsigned shard, pending, cur, nblocks;

    // first block (all lanes call).  Returns false when there is nothing for this workgroup.
    __device__ __forceinline__ bool begin(SchedState *state, unsigned *lds_slot, size_t nblocks_, int lane)
    {
        st = state; slot = lds_slot; nblocks = (unsigned)nblocks_;      // < 2^32 - 2^20 (the launchers' cap)
        shard = (blockIdx.x >> 3) & (kShards - 1);
        pending = 0;
        cur = blockIdx.x;
        return cur < nblocks;
    }
    __device__ __forceinline__ size_t block() const { return (size_t)cur; }
    __device__ __forceinline__ unsigned block_of_draw(unsigned n) const { return gridDim.x + n * kShards + shard; }
    // behind the first butterflies of the block (no load outstanding): start the draw for the next block (lane 0)
    __device__ __forceinline__ void draw(int lane)
    {
        if (lane == 0) pending = atomicAdd(counter(), 1u);
    }
    // the shard's counter through a scalar base: the address is wave-uniform, and said so it costs no VGPR pair across the block
    __device__ __forceinline__ unsigned *counter() const { return st->ctr + __builtin_amdgcn_readfirstlane(shard * kShardStride); }
    // later in the block, IN FRONT of a barrier every lane passes before advance(): park the draw in LDS
    __device__ __forceinline__ void publish(int lane)
    {
        if (lane == 0) *slot = block_of_draw(pending);
    }
    // end of the block (all lanes): false when the workgroup is done
    __device__ __forceinline__ bool advance()
    {
        cur = __builtin_amdgcn_readfirstlane(*slot);
        return cur < nblocks;
    }
    // every exit path of the kernel (all lanes call, lane 0 acts): the last workgroup of the launch resets the books
    __device__ __forceinline__ void finish(int lane)
    {
        if (lane == 0 && atomicAdd(&st->finished, 1u) == gridDim.x - 1) {
            for (unsigned g = 0; g < kShards; g++) atomicExch(&st->ctr[g * kShardStride], 0u);
            atomicExch(&st->finished, 0u);
        }
    }
};

// A gate for the blocks of a launch that read the halo slot of a sharded stream (pcx_shard.hip, stream.py): the launch covers the
// WHOLE shard, walks its blocks in reverse order so that the blocks at the front -- the only ones whose windows reach into the halo
// -- are dealt last, and holds those until a 32-bit word in device memory has reached `value` (signed distance: the word carries
// the pass number, nobody resets it).  The word is written by a one-thread kernel queued behind the halo transfer on its stream
// (pcx_gate_signal_dev).  One launch per shard and pass instead of a body launch plus a head launch behind an event: the second
// launch's start-up, its ragged end and the kernel boundary cost 5-14 % of a pass (profiles/r02/shard_probe.txt).
// Visibility: the halo bytes were written by another kernel or by a copy engine and are complete before the signalling kernel
// starts (stream order); the word is polled with system-scope loads, and a system-scope acquire behind the poll drops whatever
// this CU's L1 holds (the XCD's L2 cannot be stale for local memory: MTYPE RW lines are probed).
struct Gate {
    const unsigned *word;     // nullptr: no gate, natural block order
    unsigned value, blocks;   // blocks 0 .. blocks-1 wait
};
// The wait is BOUNDED: a signal that never comes (a caller's bug, a failed transfer) must not hang the device.  After kGateTimeout
// the block goes ahead on whatever the halo slot holds and leaves kGateTimedOut in the word behind the gate word, where the host
// finds it (pcx_shard_sync and pcx_shard_gather report PCX_ERR_STATE and clear it; stream.py check_gate()).
constexpr unsigned long long kGateTimeoutTicks = 200000000ull;     // 2 s of s_memrealtime (100 MHz)
constexpr unsigned kGateTimedOut = 0xDEADu;
__device__ __forceinline__ void gate_wait(const Gate &g, int lane)
{
    if (lane == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(__hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - g.value) < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (__builtin_amdgcn_s_memrealtime() - t0 > kGateTimeoutTicks) {
                __hip_atomic_store(const_cast<unsigned *>(g.word) + 1, kGateTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// One spelling for both walks, so a kernel is written once: BlockWalk<false> is the grid stride, BlockWalk<true> the dealer.
//     BlockWalk<DYN> walk;
//     if (!walk.begin(sched, &slot, nblocks, lane)) { walk.finish(lane); return; }
//     for (;;) { b = walk.block(); loads; first butterflies; walk.draw(lane); ...; walk.publish(lane); <a barrier>; ...; if (!walk.advance()) break; }
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

// Dealing for a kernel that prefetches the NEXT block's input while it works on the current one (fft4096_kernel, which the product
// no longer runs -- fft.hip: the plain family kernel measured faster -- and the diagnostic library keeps for that A/B): the next
// block's index must be known at the START of a block, so blocks are dealt in strided pairs (q, q + nchunks) from counter 0, the
// draw for chunk c+1 issued during the FIRST block of chunk c, published in front of one of that block's barriers and read at the
// start of the chunk's second block.  (Only the launch's last chunk can be a single block, and nothing follows it.)
constexpr unsigned kAheadChunk = 2;
struct AheadDealer {
    SchedState *st;
    unsigned *slot;
    unsigned nchunks, chunk, sub, pending, next_chunk;
    size_t nblocks;

    __device__ __forceinline__ bool begin(SchedState *state, unsigned *lds_slot, size_t nblocks_, int lane)
    {
        st = state; slot = lds_slot; nblocks = nblocks_;
        nchunks = (unsigned)((nblocks_ + kAheadChunk - 1) / kAheadChunk);
        sub = 0; pending = 0; next_chunk = ~0u;
        if (lane == 0) *slot = atomicAdd(&st->ctr[0], 1u);
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
    __device__ __forceinline__ void draw(int lane) { if (lane == 0 && sub == 0 && pair()) pending = atomicAdd(&st->ctr[0], 1u); }
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
            atomicExch(&st->ctr[0], 0u);
            atomicExch(&st->finished, 0u);
        }
    }
};

}  // namespace pcx
