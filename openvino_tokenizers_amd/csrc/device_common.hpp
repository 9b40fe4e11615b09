// device_common.hpp -- wave-level helpers and the structs shared by host and device code.
//
// gfx950 only: wave64, one 1-D block = 4 waves.  Cross-lane traffic goes through __ballot / __shfl
// (DPP/readlane on CDNA4); every LDS hand-off between lanes of a wave is fenced with wave_sync().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ovtk {

constexpr int kWave = 64;
constexpr int kBlockThreads = 256;  // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlockThreads / kWave;

// Marker for an unused entry of the staging id buffer ("slotted" rows, see encode kernel).
constexpr int32_t kEmptyId = INT32_MIN;

// Status block of one run, written by the kernels, read by the host after the final sync.  The
// deferred-piece list is sharded (one region + one counter per shard, each counter on its own 128-byte
// line: a single device-scope counter saturates at ~90 atomics/us on gfx950).
constexpr int kShardHeaderBytes = 16;  // row-shard exchange wire: i32 n_ids, i32 n_rows, 2 x i32 0 (ops_kernels.hpp)
constexpr int kShards = 16;
constexpr int kCounterStride = 32;  // int32 slots between shard counters
struct RunStatus {
    int32_t n_items;       // (row, string) work items in output order
    int32_t stage_need;    // staging-buffer entries the batch needs
    int32_t n_out;         // final number of output elements (token ids / pieces / chars)
    int32_t n_exact;       // pieces handed to the exact-heap kernel
    uint32_t scratch_used; // bytes taken from the exact kernel's scratch pool
    uint32_t flags;        // kFlag*
    uint32_t ticket[2];    // "last block done" tickets of prep_rows_kernel / count_scan_kernel
    uint32_t rows_done;    // bit s: row-ticket shard s is exhausted (lookup_kernel)
    int32_t n_pending;     // rows the span / rows kernel left to lookup_kernel<kFused>
    int32_t n_store_probe; // deferred pieces merge_kernel looked up in the piece store (a sample: one wave in 64 counts) ...
    int32_t n_store_hit;   // ... and found there
    int32_t width;         // dense output (ovtk_encode_dense_*): the row width row_width_kernel settled on
    uint32_t width_ticket; // ... and its "last block done" ticket
    int32_t n_unresolved;  // the short path, HOST SIDE (RowsRun::finish sums the shards' deferred counts into it for the handle's predictors): pieces
                           // filed for merge_kernel / wordpiece_deferred_kernel -- with EncodeWork::span_sums what neither the memo nor the
                           // piece store holds (the span kernel looks its misses up in the store itself)
    int32_t short_path;    // (host side, for the handle's predictors: 1 the call's first set of launches did it, 2 a second set was needed, 3 started over the long way)
    int32_t pad[16];
    int32_t shard_count[kShards * kCounterStride];  // [s * kCounterStride] = deferred pieces pushed to shard s
    int32_t stage_top[kShards * kCounterStride];    // [s * kCounterStride] = staging entries handed out in region s
    int32_t row_ticket[kShards * kCounterStride];   // [s * kCounterStride] = rows of range s handed to waves (lookup_kernel)
    int32_t batch_ticket[kShards * kCounterStride]; // [s * kCounterStride] = 64-piece batches of shard s handed out (merge_kernel)
    uint32_t done_ticket[kShards * kCounterStride]; // [s * kCounterStride] = blocks of shard s that are through (last_block_done_sharded)
};
static_assert(sizeof(RunStatus) % 16 == 0, "tile_cnt stands behind the status block and is read 16 bytes at a time (compact_kernel)");
constexpr uint32_t kFlagItemsOverflow = 1u;     // more work items than the workspace holds
constexpr uint32_t kFlagStageOverflow = 2u;     // staging buffer too small
constexpr uint32_t kFlagDeferOverflow = 4u;     // deferred-piece list too small
constexpr uint32_t kFlagScratchOverflow = 8u;   // exact kernel scratch pool too small
constexpr uint32_t kFlagOutCapacity = 16u;      // caller's output buffer too small
constexpr uint32_t kFlagRange = 32u;            // an input offset left its buffer
constexpr uint32_t kFlagExactOverflow = 64u;    // exact-path piece list too small
constexpr uint32_t kFlagDidNotRun = 0x80000000u;  // host side only: what the pinned status block holds until compact_kernel has written it
constexpr uint32_t kFlagTailPending = 128u;     // merge_kernel's folded tail left the exact pieces / row scan to separate launches

__device__ __forceinline__ int lane_id() { return int(threadIdx.x) & (kWave - 1); }
__device__ __forceinline__ int wave_in_block() { return int(threadIdx.x) >> 6; }

#ifdef OVTK_PROBE
// Diagnostic build (tools/probe_merge.py): wall_clock64() stamps per wave of merge_kernel (slots 5 / 6: the wave's largest symbol count / merge steps), plain stores (atomics on one
// address would serialise the waves they are meant to time).
static __device__ unsigned long long g_ts[8192][12];
#define PROBE(i)                                                                                                          \
    do {                                                                                                                  \
        const int pw_ = (int(blockIdx.y) * int(gridDim.x) + int(blockIdx.x)) * kWavesPerBlock + wave_in_block();         \
        if (lane_id() == 0 && pw_ < 8192) g_ts[pw_][i] = (unsigned long long)wall_clock64();                             \
    } while (0)
#else
#define PROBE(i)
#endif

// Orders the LDS accesses of the lanes of one wave -- the wave-level counterpart of __syncthreads() for data handed
// from lane to lane through LDS.  Deliberately NOT a fence: a wavefront-scope fence makes the compiler drain every
// outstanding global load and store (s_waitcnt vmcnt(0)) at each hand-off.  LDS operations of one wave execute in
// issue order; what is needed is that the compiler keeps its memory operations on their side of this point (the asm
// memory clobber) and that earlier LDS results have landed (lgkmcnt).
__device__ __forceinline__ void wave_sync() {
#ifdef OVTK_SIMT_EMULATOR
    __builtin_amdgcn_wave_barrier();
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#endif
}
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }  // tells the compiler too
// Load of a kernel INPUT (never written during the launch) at a wave-uniform address: through the constant address
// space, so that it becomes a scalar load (s_load, tracked by lgkmcnt) even after the kernel has issued stores --
// a vector load here would be waited for with vmcnt(0) and drain every prefetch in flight.
template <typename T>
__device__ __forceinline__ T uniform_load(const T* p) {
#ifdef OVTK_SIMT_EMULATOR
    return *p;
#else
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
#endif
}

// Streaming accesses -- the text a row is read from once, the staged ids a later kernel reads once -- carry the non-temporal
// hint (global_load / global_store ... nt): they do not age the lines the XCD's 4 MiB L2 should keep, the memo and merge
// tables every wave probes.  (Counted in round 3, tools/fetch_calib.hip for the units: the round-2 lookup kernel fetched 118 MB to
// read 34.6 MB of text -- the rest were memo probes whose lines the streams had pushed out.)
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if defined(OVTK_SIMT_EMULATOR)
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}
template <typename T>
__device__ __forceinline__ void stream_store(T* p, T v) {
#if defined(OVTK_SIMT_EMULATOR)
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

// Bits [sh, sh + 32) of the 64-bit value hi:lo, sh in 0..31 -- ONE v_alignbit_b32.  Written as a 64-bit shift the compiler
// emits v_lshrrev_b64 on a register pair it first has to assemble; the scanners and the key fetch do this for every dword.
__device__ __forceinline__ uint32_t funnel_shr(uint32_t lo, uint32_t hi, int sh) {
#ifdef OVTK_SIMT_EMULATOR
    return uint32_t(((static_cast<unsigned long long>(hi) << 32) | lo) >> sh);
#else
    return __builtin_amdgcn_alignbit(hi, lo, uint32_t(sh));
#endif
}

// Publishing data to other workgroups of the same launch (MI355X_MICROARCH.md "inter-workgroup visibility"): the
// producer's plain stores -> __syncthreads() -> ONE lane: agent-scope release + vmcnt drain -> device-scope atomic
// ticket; the block that draws the last ticket does an agent-scope acquire -> __syncthreads() -> plain loads.
__device__ __forceinline__ void publish_release() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifndef OVTK_SIMT_EMULATOR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// Waits until this wave's outstanding global loads, stores and atomics have completed.
__device__ __forceinline__ void drain_vmem() {
#ifndef OVTK_SIMT_EMULATOR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void publish_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
// Set bits of `m` below this lane's: the lane's rank among the lanes of a ballot (v_mbcnt_lo / _hi: two instructions, where
// popcount(m & lanemask_lt()) is two ANDs and two counts)
__device__ __forceinline__ int rank_below(unsigned long long m) {
#ifndef OVTK_SIMT_EMULATOR
    return int(__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)));
#else
    return __popcll(m & lanemask_lt());
#endif
}

// ---- DPP building blocks (VALU-speed cross-lane moves; __shfl goes through the LDS crossbar) ----
constexpr int kDppRowShl = 0x100, kDppRowShr = 0x110, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;
template <int CTRL, int ROW_MASK = 0xF, bool BOUND = true>
__device__ __forceinline__ int dpp_mov0(int v) {  // lanes without a valid source / outside ROW_MASK receive 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, BOUND);
}
// Inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16, then two row broadcasts.
__device__ __forceinline__ int wave_incl_sum(int v) {
    v += dpp_mov0<kDppRowShr + 1>(v);
    v += dpp_mov0<kDppRowShr + 2>(v);
    v += dpp_mov0<kDppRowShr + 4>(v);
    v += dpp_mov0<kDppRowShr + 8>(v);
    v += dpp_mov0<kDppRowBcast15, 0xA, false>(v);
    v += dpp_mov0<kDppRowBcast31, 0xC, false>(v);
    return v;
}
// Inclusive prefix maximum over the 64 lanes, non-negative values (0 is the identity the DPP moves fill in).
__device__ __forceinline__ int wave_incl_max(int v) {
    int t;
    t = dpp_mov0<kDppRowShr + 1>(v); v = t > v ? t : v;
    t = dpp_mov0<kDppRowShr + 2>(v); v = t > v ? t : v;
    t = dpp_mov0<kDppRowShr + 4>(v); v = t > v ? t : v;
    t = dpp_mov0<kDppRowShr + 8>(v); v = t > v ? t : v;
    t = dpp_mov0<kDppRowBcast15, 0xA, false>(v); v = t > v ? t : v;
    t = dpp_mov0<kDppRowBcast31, 0xC, false>(v); v = t > v ? t : v;
    return v;
}
// Value of lane `src` (wave-uniform index) as a scalar.
__device__ __forceinline__ int wave_readlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ unsigned long long wave_readlane(unsigned long long v, int src) {
    const unsigned lo = unsigned(__builtin_amdgcn_readlane(int(unsigned(v)), src));
    const unsigned hi = unsigned(__builtin_amdgcn_readlane(int(unsigned(v >> 32)), src));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
__device__ __forceinline__ int wave_sum(int v) { return wave_readlane(wave_incl_sum(v), kWave - 1); }
// 64-bit value of the previous / next lane inside a row of 16 lanes (0 at the row's edge).
__device__ __forceinline__ unsigned long long row_prev(unsigned long long v) {
    const unsigned lo = unsigned(dpp_mov0<kDppRowShr + 1>(int(unsigned(v))));
    const unsigned hi = unsigned(dpp_mov0<kDppRowShr + 1>(int(unsigned(v >> 32))));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
__device__ __forceinline__ unsigned long long row_next(unsigned long long v) {
    const unsigned lo = unsigned(dpp_mov0<kDppRowShl + 1>(int(unsigned(v))));
    const unsigned hi = unsigned(dpp_mov0<kDppRowShl + 1>(int(unsigned(v >> 32))));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
// Value of lane l-1 / l+1 of the whole wavefront (0 for lane 0 / lane 63): GFX9's DPP wavefront shifts.
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return uint32_t(dpp_mov0<kDppWaveShr1>(int(v))); }
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return uint32_t(dpp_mov0<kDppWaveShl1>(int(v))); }
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        int t = __shfl_xor(v, d);
        v = t > v ? t : v;
    }
    return v;
}
// Minimum of a 64-bit key over the wave (all lanes receive it).
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        unsigned long long t = __shfl_xor(v, d);
        v = t < v ? t : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_bcast(T v, int src) { return __shfl(v, src); }

}  // namespace ovtk
