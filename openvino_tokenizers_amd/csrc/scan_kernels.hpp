// scan_kernels.hpp -- device-wide exclusive scans (the parallel form of the reference's running
// `ragged_offset`, e.g. src/bpe_tokenizer.cpp:141-161): per-tile sums -> one block scans the tile
// sums -> per-tile rescan + apply.  Three small launches; every element is read twice.
#pragma once

#include "device_common.hpp"

namespace ovtk {

constexpr int kScanThreads = 1024;
constexpr int kScanPerThread = 4;
constexpr int kTileThreads = 256;
constexpr int kTileElems = kTileThreads * kScanPerThread;  // 1024 elements per block

// Exclusive scan of f(0..n) by ONE block of THREADS threads: put(i, prefix) for every i, returns the
// total to every thread.  64-bit accumulation (callers clamp / flag).
template <int THREADS, int PER = kScanPerThread, class F, class Put>
__device__ __forceinline__ long long block_exclusive_scan(int n, F&& f, Put&& put) {
    __shared__ long long wave_tot[THREADS / kWave];
    __shared__ long long carry_s;
    const int tid = int(threadIdx.x), l = lane_id(), wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int tile = 0; tile < n; tile += THREADS * PER) {
        const int i0 = tile + tid * PER;
        long long v[PER], s = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            v[j] = (i0 + j < n) ? (long long)f(i0 + j) : 0;
            s += v[j];
        }
        long long incl = s;  // inclusive scan of s over the wave
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            long long t = __shfl_up(incl, d);
            if (l >= d) incl += t;
        }
        if (l == kWave - 1) wave_tot[wv] = incl;
        __syncthreads();
        long long before = carry_s;
        for (int k = 0; k < wv; ++k) before += wave_tot[k];
        long long run = before + incl - s;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (i0 + j < n) put(i0 + j, run);
            run += v[j];
        }
        __syncthreads();
        if (tid == THREADS - 1) carry_s = run;  // last thread's running sum = total so far
        __syncthreads();
    }
    return carry_s;
}

template <class LenF>
static __global__ __launch_bounds__(kTileThreads) void tile_reduce_kernel(long long n, LenF f, long long* tile_sums) {
    __shared__ long long part[kTileThreads / kWave];
    const long long i0 = (long long)blockIdx.x * kTileElems + (long long)threadIdx.x * kScanPerThread;
    long long s = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j)
        if (i0 + j < n) s += f(i0 + j);
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane_id() == 0) part[wave_in_block()] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int k = 0; k < kTileThreads / kWave; ++k) t += part[k];
        tile_sums[blockIdx.x] = t;
    }
}

// One block: exclusive scan of the tile sums in place; fin(total) runs on thread 0.
template <class Fin>
static __global__ __launch_bounds__(kScanThreads) void tile_scan_kernel(int n_tiles, long long* tile_sums, Fin fin) {
    const long long total = block_exclusive_scan<kScanThreads>(
        n_tiles, [&](int t) -> long long { return tile_sums[t]; }, [&](int t, long long off) { tile_sums[t] = off; });
    if (threadIdx.x == 0) fin(total);
}

// apply(i, offset, len) is called for every element with its global exclusive offset.
template <class LenF, class ApplyF>
static __global__ __launch_bounds__(kTileThreads) void tile_apply_kernel(long long n, LenF f, const long long* tile_offs,
                                                                         ApplyF apply, const RunStatus* status,
                                                                         uint32_t skip_flags) {
    __shared__ long long part[kTileThreads / kWave];
    if (status->flags & skip_flags) return;
    const long long i0 = (long long)blockIdx.x * kTileElems + (long long)threadIdx.x * kScanPerThread;
    long long v[kScanPerThread], s = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
        v[j] = (i0 + j < n) ? (long long)f(i0 + j) : 0;
        s += v[j];
    }
    long long incl = s;
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        long long t = __shfl_up(incl, d);
        if (l >= d) incl += t;
    }
    if (l == kWave - 1) part[wave_in_block()] = incl;
    __syncthreads();
    long long run = tile_offs[blockIdx.x] + incl - s;
    for (int k = 0; k < wave_in_block(); ++k) run += part[k];
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
        if (i0 + j < n) apply(i0 + j, run, v[j]);
        run += v[j];
    }
}

// f(i) for every i < n: one LANE per element (an element's work is a serial walk: the tile kernels above give a thread
// kScanPerThread elements in a row, a quarter of the lanes there could be), or one WAVE per element (all 64 lanes call f(i): an
// element is a string whose bytes the lanes share).  Round 5: UTF8Validate, StringTensorPack and TrieTokenizer ran their walks
// inside the tile kernels -- twice in the apply pass -- at 2.1 / 0.9 / 15.6 ms for a config-2 batch (tools/ops_timing.py).
template <class F>
static __global__ __launch_bounds__(kTileThreads) void each_kernel(long long n, F f, const RunStatus* status, uint32_t skip_flags) {
    if (status && (status->flags & skip_flags)) return;
    const long long i = (long long)blockIdx.x * kTileThreads + threadIdx.x;
    if (i < n) f(i);
}
template <class F>
static __global__ __launch_bounds__(kTileThreads) void each_wave_kernel(long long n, F f, const RunStatus* status, uint32_t skip_flags) {
    if (status && (status->flags & skip_flags)) return;
    const long long stride = (long long)gridDim.x * (kTileThreads / kWave);
    for (long long i = (long long)blockIdx.x * (kTileThreads / kWave) + wave_in_block(); i < n; i += stride) f(i);
}

}  // namespace ovtk
