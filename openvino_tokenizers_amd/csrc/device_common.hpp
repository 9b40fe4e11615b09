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
constexpr int kShards = 16;
constexpr int kCounterStride = 32;  // int32 slots between shard counters
struct RunStatus {
    int32_t n_items;       // (row, string) work items in output order
    int32_t stage_need;    // staging-buffer entries the batch needs
    int32_t n_out;         // final number of output elements (token ids / pieces / chars)
    int32_t n_exact;       // pieces handed to the exact-heap kernel
    uint32_t scratch_used; // bytes taken from the exact kernel's scratch pool
    uint32_t flags;        // kFlag*
    int32_t pad[26];
    int32_t shard_count[kShards * kCounterStride];  // [s * kCounterStride] = deferred pieces pushed to shard s
};
constexpr uint32_t kFlagItemsOverflow = 1u;     // more work items than the workspace holds
constexpr uint32_t kFlagStageOverflow = 2u;     // staging buffer too small
constexpr uint32_t kFlagDeferOverflow = 4u;     // deferred-piece list too small
constexpr uint32_t kFlagScratchOverflow = 8u;   // exact kernel scratch pool too small
constexpr uint32_t kFlagOutCapacity = 16u;      // caller's output buffer too small
constexpr uint32_t kFlagRange = 32u;            // an input offset left its buffer
constexpr uint32_t kFlagExactOverflow = 64u;    // exact-path piece list too small

__device__ __forceinline__ int lane_id() { return int(threadIdx.x) & (kWave - 1); }
__device__ __forceinline__ int wave_in_block() { return int(threadIdx.x) >> 6; }

// Orders LDS/global accesses of the lanes of one wave (compiler + hardware) -- the wave-level
// counterpart of __syncthreads().
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Inclusive prefix sum over the 64 lanes (Hillis-Steele on __shfl_up; 6 steps).
__device__ __forceinline__ int wave_incl_sum(int v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        int t = __shfl_up(v, d);
        if (l >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
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
