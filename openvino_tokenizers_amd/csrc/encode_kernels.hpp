// encode_kernels.hpp -- the gfx950 kernels of RegexSplit, BPETokenizer and their fusion.
//
// Pipeline of one BPETokenizer / fused encode call (host side: api_common.hpp run_rows_to_ids), 6 launches:
//   prep_rows_kernel            validates the offsets; staging capacity of every wave's rows; the block that finishes
//                               last scans them into per-wave arena offsets
//   lookup_kernel<kFused>       ONE WAVE PER ROW (persistent waves, rows strided): split scanner -> pieces -> memo probe
//   lookup_kernel<kPieces>      same for pre-split pieces (the BPETokenizer op contract)
//       a piece found in the memo (tables.hpp PieceEntry: BPE(piece) precomputed for every vocabulary
//       token) writes its ids straight into the row's staging stretch; any other piece reserves
//       len + |end_suffix| staging entries and goes to the deferred list (per-wave LDS buffer, flushed 64 at a
//       time into one of kShards list regions)
//   merge_kernel                dense batches of 64 deferred pieces: path F (lane per piece), path W (wave
//                               per piece), ties / oversized pieces -> exact list           (bpe_device.hpp)
//   exact_kernel                path X, one lane per piece
//   count_scan_kernel           per-tile sums of the row counts, last block scans them = the reference's running
//                               `ragged_offset` (bpe_tokenizer.cpp:141-161) at tile granularity
//   compact_kernel              row offset = tile offset + in-tile prefix (rebuilt by the wave), begins/ends, and
//                               staging -> caller's ids buffer (unused entries = kEmptyId are squeezed out by ballot
//                               compaction)
// The fused encode of the pattern families (span_kernel.hpp) runs lookup_span_kernel -> lookup_kernel<kFused> (only_pending: the rows the
// span kernel leaves) -> merge_kernel -> compact_kernel.  With EncodeWork::fold_tail the exact pieces and the row scan are merge_kernel's
// last block's (no exact / count_scan launches); with EncodeWork::tile_sums (round 5: up to 1 024 tiles) there is no last block either --
// an exact piece is its lane's, and compact_kernel derives a tile's offset from the tile counts itself.
#pragma once

#include <cstddef>
#include <type_traits>

#include "bpe_device.hpp"
#include "device_common.hpp"
#include "scan_kernels.hpp"
#include "split_device.hpp"
#include "split_seq_device.hpp"

namespace ovtk {

struct RowsIn {
    const int32_t* ragged_begins;
    const int32_t* ragged_ends;
    int32_t n_rows;
    const int32_t* begins;
    const int32_t* ends;
    int32_t n_strings;
    const uint8_t* chars;
    int64_t n_chars;
    const uint8_t* skips;  // bool per string, or nullptr
};

// A piece that missed the memo.  len <= kPieceKeyBytes: k0/k1 hold its bytes (the memo key); always: begin/len
// locate it in `chars`.
struct alignas(32) DeferredPiece {
    uint64_t k0, k1;
    int32_t stage_pos, row, begin, len;
};
struct ExactPiece { int32_t begin, len, stage_pos, row; };

constexpr int kMissBuf = 96;   // per-wave LDS buffer of deferred pieces: flushed 64 at a time, or early when a batch would not fit
                               // (3 KB per wave instead of 4: with the 2.3 KB scan window a seventh block fits a CU's LDS)

constexpr int kRowTile = 64;  // rows per tile of the final offset scan
constexpr int32_t kRowPending = -1;  // row_used: lookup_span_kernel / lookup_rows_kernel left the row to lookup_kernel<kFused>

struct EncodeWork {
    int32_t fold_tail;      // merge_kernel's last block also runs exact pieces + the row scan (no exact / count_scan launches)
    int32_t compact_split;  // compact_kernel<RaggedSink>: waves that share a work item of kCompactRows rows (a power of two; > 1 for few, long rows)
    int32_t tile_sums;      // (with fold_tail, the large path) ... or nobody does: merge_kernel / wordpiece_deferred_kernel end without
                            // a ticket, an exact piece is worked out by the lane that finds it, and compact_kernel derives a tile's
                            // offset from tile_cnt itself (the sum of the counts in front of it) and the total in its block 0
    long long out_cap;      // caller's ids capacity (the folded tail's capacity check)
    int32_t only_pending;   // lookup_kernel<kFused>: take only the rows the span / rows kernel marked kRowPending in row_used
    int32_t small;          // the whole call is ONE launch of encode_small_kernel (one block): the fields below are set
    int32_t* out_ids;       //   caller's ids / begins / ends (device pointers) ...
    int32_t* out_begins;
    int32_t* out_ends;
    RunStatus* host_status; //   ... and the pinned status block the kernel itself fills (no memset / copy dispatches)
    int32_t status_words;   //   dwords of `status` to zero at the kernel's start (RunStatus + the tile counts behind it)
    RunStatus* next_status; // the large path (compact_kernel is its last kernel): the OTHER of the workspace's two status blocks,
                            // zeroed by compact_kernel for the workspace's next call, while this call's status goes to host_status
                            // from the kernel -- no memset and no copy dispatch (status_words dwords); nullptr: the host does both
    int32_t rows_per_ticket;  // lookup_kernel, allocator mode: 0 = static rows per wave, else rows handed out per ticket
    int32_t rows_per_wave;    // lookup_rows_kernel: wave w owns the rows [w * rows_per_wave, (w + 1) * rows_per_wave)
    int32_t* pending_rows;    // [n_rows] or nullptr: the rows lookup_rows_kernel left to the generic kernel, status->n_pending of
                              //          them in no particular order (nullptr: the generic kernel looks for kRowPending in row_used)
    int32_t n_waves;        // persistent waves of the prep / lookup launches (wave w owns rows w, w + n_waves, ...)
    long long* wave_off;    // [n_waves + 1] staging arena of each wave (exclusive scan of its rows' capacities), or
                            // nullptr: the lookup kernel takes staging chunks from kShards bump allocators itself
    int32_t stage_region;   // allocator mode: entries per allocator region (stage_cap / kShards)
    int32_t* row_stage;     // [n_rows]     staging offset of each row (set by the lookup kernel)
    int32_t* row_cnt;       // [n_rows]     ids produced by each row
    int32_t* tile_cnt;      // [n_tiles] or nullptr: the counts summed per tile of kRowTile rows while merge_kernel runs
                            //           (atomics into the zeroed tail of the status buffer) -- what the folded tail scans
    int32_t* row_emit;      // [n_rows] (with tile_cnt): ids the lookup kernel itself wrote for the row, i.e. row_cnt
                            //           before merge_kernel's additions (a copy: merge blocks sum it while others add)
    int32_t* row_used;      // [n_rows]     staging entries the row occupies (> row_cnt: it has unused entries)
    long long* tile_off;    // [n_tiles]    output offset of each tile of kRowTile rows
    int32_t* stage;         // staging entries: i32 -- or u16 in the same buffer (stage16) --, kEmptyId / 0xFFFF = unused
    int32_t stage_cap;
    int32_t stage16;        // every id the call can produce is below 65535 (BPE handles with narrow ids): the staged ids cross HBM
                            // as two bytes each way instead of four (stage_put / stage_get)
    DeferredPiece* deferred;  // kShards regions of shard_cap entries
    int32_t shard_cap;
    ExactPiece* exact;
    int32_t exact_cap;
    uint8_t* scratch;
    uint32_t scratch_cap;
    RunStatus* status;
    // ---- the short path (round 6; span_kernel.hpp "the short path")
    int32_t span_sums;       // lookup_span_kernel looks the pieces the memo does not hold up in the piece store itself, counts what is left
                             // (the deferred list's counters) and sums its rows' ids into tile_cnt; the kernels behind it add to these (lookup_kernel
                             // with only_pending: its rows' counts, its misses) instead of summing row_emit once more
    int32_t skip_mask;       // kernels of the middle that were not launched: kSkipPending lookup_kernel<kFused> for left-over rows, kSkipMerge
                             // merge_kernel / wordpiece_deferred_kernel -- compact_kernel writes nothing when one of them had work
    int32_t merge_hint;      // (host side) pieces the last call of the handle left for merge_kernel / wordpiece_deferred_kernel, < 0: unknown
    int32_t launch_mask;     // (host side) what the middle launches now: kLaunchSpan its first kernel, kLaunchPending lookup_kernel<kFused> for the
                             // rows that one leaves, kLaunchMerge merge_kernel / wordpiece_deferred_kernel
};

constexpr uint32_t kFatalFlags = kFlagRange | kFlagStageOverflow;
constexpr int kSkipPending = 1, kSkipMerge = 2;   // EncodeWork::skip_mask
constexpr int kLaunchSpan = 1, kLaunchPending = 2, kLaunchMerge = 4, kLaunchAll = 7;   // EncodeWork::launch_mask

// One staging entry (w.stage16 is a kernel argument: the branch is scalar).
__device__ __forceinline__ void stage_put(const EncodeWork& w, int pos, int32_t id) {
    if (w.stage16) reinterpret_cast<uint16_t*>(w.stage)[pos] = uint16_t(id);
    else w.stage[pos] = id;
}
__device__ __forceinline__ void stage_clear(const EncodeWork& w, int pos) {
    if (w.stage16) reinterpret_cast<uint16_t*>(w.stage)[pos] = uint16_t(0xFFFFu);
    else w.stage[pos] = kEmptyId;
}
__device__ __forceinline__ int32_t stage_get(const EncodeWork& w, int pos) {
    if (w.stage16) {
        const uint32_t x = reinterpret_cast<const uint16_t*>(w.stage)[pos];
        return x == 0xFFFFu ? kEmptyId : int32_t(x);
    }
    return w.stage[pos];
}

// out[i] = i: the ragged dimension of a batch of plain strings (one string per row).
static __global__ __launch_bounds__(kBlockThreads) void iota_kernel(int n, int32_t* out) {
    const int stride = int(gridDim.x) * kBlockThreads;
    for (int i = int(blockIdx.x) * kBlockThreads + int(threadIdx.x); i < n; i += stride) out[i] = i;
}

// ---- row capacity: mul * sum over the row's strings of max(len, 1), with range validation.
__device__ __forceinline__ long long row_capacity(const RowsIn& in, int mul, long long row, RunStatus* status) {
    long long cap = 0;
    const int b = in.ragged_begins[row], e = in.ragged_ends[row];
    if (b < e && (b < 0 || e > in.n_strings)) {
        atomicOr(&status->flags, kFlagRange);
        return 0;
    }
    for (int col = b; col < e; ++col) {
        const long long sb = in.begins[col], se = in.ends[col];
        if (sb < 0 || se < sb || se > in.n_chars) {
            atomicOr(&status->flags, kFlagRange);
            return 0;
        }
        cap += (se - sb > 0 ? se - sb : 1) * mul;
    }
    return cap;
}

// True in every thread of the block that draws the last of `n_blocks` tickets; its loads then see what all other
// blocks stored before taking theirs.
// release = false: the block made nothing but device-scope atomics visible to the last block (an agent-scope release
// writes the XCD's dirty L2 lines back -- hundreds of blocks doing that at the end of a kernel cost tens of us).
__device__ __forceinline__ bool last_block_done(uint32_t* ticket, unsigned n_blocks, bool release = true) {
    __shared__ int is_last_s;
    // Every wave's own atomics (the per-tile and per-row counts are returnless adds) must have been performed before the
    // block's ticket is drawn: __syncthreads() is a workgroup-scope fence and does not wait for them, and a ticket that
    // overtakes them lets the last block scan counts that are a batch short (tools/soak.py: one encode in ~15 000 lost
    // the ids of one 64-piece batch in one tile).
    drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (release) publish_release();
        const bool last = atomicAdd(ticket, 1u) == n_blocks - 1u;
        if (last) publish_acquire();
        is_last_s = last ? 1 : 0;
    }
    __syncthreads();
    return is_last_s != 0;
}

// The same for the 2-D grids of the kernels that work through the deferred list (kShards x blocks per shard: merge_kernel,
// wordpiece_deferred_kernel): a ticket per shard first, and only the block that draws a shard's last one goes on to the one counter
// of the whole grid.  One address serves ~90 atomics per microsecond, and most blocks of such a grid find no batch and arrive at
// once: 2 048 tickets on one counter kept the block that actually had work waiting for 20 us (round 4; 1 024: 11 us).
__device__ __forceinline__ bool last_block_done_sharded(RunStatus* st, int shard, unsigned blocks_per_shard, bool release = true) {
    __shared__ int is_last_s2;
    drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (release) publish_release();
        bool last = atomicAdd(&st->done_ticket[shard * kCounterStride], 1u) == blocks_per_shard - 1u;
        if (last) {
            // what the shard's blocks published is seen here, and handed on with the grid's ticket
            if (release) {
                publish_acquire();
                publish_release();
            }
            last = atomicAdd(&st->ticket[0], 1u) == unsigned(kShards) - 1u;
            if (last) publish_acquire();
        }
        is_last_s2 = last ? 1 : 0;
    }
    __syncthreads();
    return is_last_s2 != 0;
}

// One launch with the lookup kernel's geometry: wave w sums the capacities of its rows; the last block scans.
static __global__ __launch_bounds__(kBlockThreads) void prep_rows_kernel(RowsIn in, int mul, EncodeWork w) {
    const int l = lane_id();
    // few blocks (every block draws a ticket from ONE counter): each of this launch's waves covers several of the
    // lookup launch's waves
    const int my_waves = int(gridDim.x) * kWavesPerBlock;
    for (int wave = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); wave < w.n_waves; wave += my_waves) {
        long long cap = 0;
        for (long long row = wave + (long long)l * w.n_waves; row < in.n_rows; row += (long long)kWave * w.n_waves)
            cap += row_capacity(in, mul, row, w.status);
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) cap += __shfl_xor(cap, d);
        if (l == 0) w.wave_off[wave] = cap;
    }
    if (!last_block_done(&w.status->ticket[0], gridDim.x)) return;
    const long long total = block_exclusive_scan<kBlockThreads, 16>(
        w.n_waves, [&](int i) -> long long { return w.wave_off[i]; }, [&](int i, long long off) { w.wave_off[i] = off; });
    if (threadIdx.x == 0) {
        w.wave_off[w.n_waves] = total;
        w.status->stage_need = total > INT32_MAX ? INT32_MAX : int32_t(total);
        if (total > (long long)w.stage_cap) atomicOr(&w.status->flags, kFlagStageOverflow);
    }
}

// Output offset of `row` from the tile offsets: tile offset + prefix of the counts of the rows before it in its tile.
// Wave-uniform call (row uniform); also returns the row's own count.
__device__ __forceinline__ long long row_output_offset(const EncodeWork& w, int n_rows, int row, int& cnt) {
    const int tile = row / kRowTile, first = tile * kRowTile;
    const int r = first + lane_id();
    const int c = r < n_rows ? w.row_cnt[r] : 0;
    const int incl = wave_incl_sum(c);
    const int j = row - first;
    cnt = wave_readlane(c, j);
    return w.tile_off[tile] + (wave_readlane(incl, j) - cnt);
}

// ---- memo probe -------------------------------------------------------------------------------
// Key of a piece of 1..15 bytes given its first 16 bytes (garbage beyond plen is masked off).
__device__ __forceinline__ void piece_key(uint64_t raw0, uint64_t raw1, int plen, uint64_t& k0, uint64_t& k1) {
    if (plen < 8) {
        k0 = raw0 & ((1ull << (8 * plen)) - 1ull);
        k1 = 0;
    } else {
        k0 = raw0;
        k1 = raw1 & ((1ull << (8 * (plen - 8))) - 1ull);
    }
    k1 |= uint64_t(plen) << 56;
}
// 16 bytes at byte offset `off` of a dword array in LDS: ONE ds_read_b128 at a byte address (gfx950 reads LDS at any alignment;
// `tools/lds_unaligned_probe.hip` checks that on the box).  Until r03 five aligned dword reads and four funnel shifts.
struct __attribute__((packed, aligned(1))) LdsBytes16 { uint64_t lo, hi; };
__device__ __forceinline__ void lds_bytes16(const uint32_t* words, int off, uint64_t& r0, uint64_t& r1) {
    const LdsBytes16 v = *reinterpret_cast<const LdsBytes16*>(reinterpret_cast<const uint8_t*>(words) + off);
    r0 = v.lo;
    r1 = v.hi;
}
// The piece's bytes from global memory (pre-split pieces: any alignment, may end at the buffer's end).
__device__ __forceinline__ void global_bytes16(const uint8_t* p, int plen, uint64_t& r0, uint64_t& r1) {
    r0 = 0;
    r1 = 0;
    for (int i = 0; i < plen && i < 8; ++i) r0 |= uint64_t(p[i]) << (8 * i);
    for (int i = 8; i < plen; ++i) r1 |= uint64_t(p[i]) << (8 * (i - 8));
}
// Returns the id count (0..kPieceMaxIds) and fills tok, or -1 when the piece is not in the memo.  The piece's one candidate
// entry (direct-mapped table, tables.hpp piece_h) is fetched with two independent 16-byte loads issued together: written with
// `if (key matches) take the payload`, the compiler sinks the payload load behind the key compare and a lookup costs two
// dependent round trips instead of one.
struct MemoFetch {
    uint4 k, p;  // key {k0, k1} and payload {tok[3], tag} of the candidate
    uint32_t mix;
};
__device__ __forceinline__ MemoFetch memo_fetch(const PieceTableDev& P, uint64_t k0, uint64_t k1) {
    const uint32_t mix = piece_mix(k0, k1);
    const uint4* e = reinterpret_cast<const uint4*>(P.slots + piece_h(mix, P.shift));
    MemoFetch f;
    f.mix = mix;
    f.k = e[0];
    f.p = e[1];
#ifndef OVTK_SIMT_EMULATOR
    // both loads are in flight before anything looks at a result
    asm volatile("" : "+v"(f.p.x), "+v"(f.p.y), "+v"(f.p.z), "+v"(f.p.w));
#endif
    return f;
}
// (P.packed6 tables -- the word memo of the fused WordPiece path -- hold up to six u16 ids per entry; the kernels that come through
// here take the entries of up to three and leave the longer ones to the deferred path: lookup_span_kernel reads all six.)
__device__ __forceinline__ int memo_resolve(const MemoFetch& f, uint64_t k0, uint64_t k1, int32_t (&tok)[kPieceMaxIds], bool packed6 = false) {
    const uint32_t a = uint32_t(k0), b = uint32_t(k0 >> 32), c = uint32_t(k1), d = uint32_t(k1 >> 32);
    // the payload's tag with the expected one folded out: the id count (0..3) when the payload is this key's
    const uint32_t c0 = f.p.w ^ piece_tag(f.mix, 0);
    // (differences OR-ed into one word and ONE compare: written as four `==` joined by `&&` the compiler turned every
    // compare into a 0/1 value and combined them with 16-bit shifts and ors -- 15 vector instructions instead of 5)
    uint32_t x = (f.k.x ^ a) | (f.k.y ^ b) | (f.k.z ^ c) | (f.k.w ^ d);
#ifndef OVTK_SIMT_EMULATOR
    asm volatile("" : "+v"(x));   // (or the optimiser turns `(x ^ a | ...) == 0` back into the four compares)
#endif
    tok[0] = packed6 ? int32_t(f.p.x & 0xFFFFu) : int32_t(f.p.x);
    tok[1] = packed6 ? int32_t(f.p.x >> 16) : int32_t(f.p.y);
    tok[2] = packed6 ? int32_t(f.p.y & 0xFFFFu) : int32_t(f.p.z);
    return (x == 0u && c0 <= uint32_t(kPieceMaxIds)) ? int(c0) : -1;
}
// merge_kernel's side of the memo -- the reference's piece cache (bpe_tokenizer.cpp:197-205, 331-338: a piece's ids are
// kept the first time it is seen, while the cache has room -- cache_capacity entries there, PieceTableDev::room here --; nothing is ever evicted).
// This lane's piece (1..15 bytes, `cnt` <= kPieceMaxIds ids) goes into its slot if that slot is free: no entry ever moves
// or changes, so a concurrent reader sees a slot either free, or claimed (kPieceBusy never equals a key), or complete
// (payload checked by its tag).  A piece whose slot is taken stays a miss.  Returns false when nothing was added.
// tok: the payload's three dwords as the table wants them -- three i32 ids, or six u16 ids (P.packed6).
__device__ __forceinline__ bool memo_insert(const PieceTableDev& P, uint64_t k0, uint64_t k1, const int32_t (&tok)[kPieceMaxIds], int cnt) {
    // (the lookup kernel has just missed this piece; nothing is read first -- one CAS.  A slot that is being written, or holds
    // this very piece -- filed by another wave a moment ago --, or any other piece ends the attempt.)
    const uint32_t mix = piece_mix(k0, k1);
    const uint32_t d3 = uint32_t(k1 >> 32);
    uint32_t* slot = reinterpret_cast<uint32_t*>(const_cast<PieceEntry*>(P.slots) + piece_h(mix, P.shift));
    if (atomicCAS(slot + 3, 0u, kPieceBusy) != 0u) return false;
    *reinterpret_cast<uint4*>(slot + 4) = uint4{uint32_t(tok[0]), uint32_t(tok[1]), uint32_t(tok[2]), piece_tag(mix, cnt)};
    // The payload has COMPLETED at the L2 before the key that makes it reachable is sent (a workgroup-scope release: s_waitcnt
    // vmcnt(0)).  Both stores land in the same line of the same L2 and a line leaves an L2 whole, so no reader -- on this XCD or
    // another -- can meet the key without the payload; and one that did would see a tag that does not match (a zeroed payload lacks
    // the valid bit) and take the miss path, with the same result.  An agent-scope release here also wrote the XCD's dirty lines
    // back: tens of microseconds in every launch that learns (first sight of a text).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *reinterpret_cast<uint4*>(slot) = uint4{uint32_t(k0), uint32_t(k0 >> 32), uint32_t(k1), d3};
    return true;
}
__device__ __forceinline__ int memo_lookup(const PieceTableDev& P, uint64_t k0, uint64_t k1, int32_t (&tok)[kPieceMaxIds]) {
    return memo_resolve(memo_fetch(P, k0, k1), k0, k1, tok, P.packed6 != 0);
}

// ---- the piece store (tables.hpp): the memo's second level, merge_kernel's own ------------------------------------
// Key of a piece of 1..15 bytes from its first-level key halves (k1 carries the length in its top byte).
__device__ __forceinline__ void store_key_short(uint64_t k0, uint64_t k1, int len, uint32_t (&key)[8]) {
    key[0] = uint32_t(k0);
    key[1] = uint32_t(k0 >> 32);
    key[2] = uint32_t(k1);
    key[3] = uint32_t(k1 >> 32) & 0x00FFFFFFu;
    key[4] = key[5] = key[6] = 0;
    key[7] = uint32_t(len) << 24;
}
// Key of a piece of 16..31 bytes: its bytes come from the text (a deferred entry carries 15).
__device__ __forceinline__ void store_key_long(const uint8_t* p, int len, uint32_t (&key)[8]) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = 4 * w + b;
            if (i < kStoreKeyBytes && i < len) v |= uint32_t(p[i]) << (8 * b);
        }
        key[w] = v;
    }
    key[7] |= uint32_t(len) << 24;
}
__device__ __forceinline__ bool store_key_eq(const uint4& a, const uint4& b, const uint32_t (&key)[8]) {
    // (the compare chain stays here: the one-compare form of memo_resolve was tried and bought nothing measurable -- this runs once per
    // deferred piece in kernels that wait for memory, not once per piece in a kernel bound by instruction issue)
    return a.x == key[0] && a.y == key[1] && a.z == key[2] && a.w == key[3] && b.x == key[4] && b.y == key[5] && b.z == key[6] &&
           b.w == key[7];
}
// -> id count and the payload, or -1.  All kStoreWays candidates are fetched together (independent 16-byte loads).
template <bool NARROW>
__device__ __forceinline__ int store_lookup(const PieceStoreDev& S, const uint32_t (&key)[8], uint32_t (&pay)[8]) {
    const uint32_t mix = store_mix(key);
    uint4 k0[kStoreWays], k1[kStoreWays], p0[kStoreWays], p1[kStoreWays];
#pragma unroll
    for (int c = 0; c < kStoreWays; ++c) {
        const uint4* e = reinterpret_cast<const uint4*>(S.slots + store_h(mix, c, S.shift));
        k0[c] = e[0];
        k1[c] = e[1];
        p0[c] = e[2];
        p1[c] = e[3];
    }
    bool any = false;
    uint4 q0 = p0[0], q1 = p1[0];
#pragma unroll
    for (int c = 0; c < kStoreWays; ++c) {
        const bool m = store_key_eq(k0[c], k1[c], key);
        if (m) {
            q0 = p0[c];
            q1 = p1[c];
        }
        any = any || m;
    }
    // the lines behind a full one (tables.hpp kStoreOverflow): rare, one more round trip each, for the lanes concerned
    bool full = !any;
#pragma unroll
    for (int c = 0; c < kStoreWays; ++c) full = full && k1[c].w != 0u;
    for (int step = 1; step <= kStoreOverflow && full; ++step) {   // (a lane's own loop: callers sit in divergent code)
#pragma unroll
        for (int c = 0; c < kStoreWays; ++c) {
            const uint4* e = reinterpret_cast<const uint4*>(S.slots + store_h_next(store_h(mix, c, S.shift), step, S.shift));
            const uint4 a = e[0], b = e[1];
            uint4 c2 = e[2], c3 = e[3];
#ifndef OVTK_SIMT_EMULATOR
            // (all four loads in flight before anything looks at a result: written as "if the key matches, take the payload" the
            // payload is a second round trip -- memo_fetch)
            asm volatile("" : "+v"(c2.x), "+v"(c2.y), "+v"(c2.z), "+v"(c2.w), "+v"(c3.x), "+v"(c3.y), "+v"(c3.z), "+v"(c3.w));
#endif
            if (!any && store_key_eq(a, b, key)) {
                any = true;
                q0 = c2;
                q1 = c3;
            }
            full = full && b.w != 0u;
        }
        full = full && !any;
    }
    pay[0] = q0.x; pay[1] = q0.y; pay[2] = q0.z; pay[3] = q0.w;
    pay[4] = q1.x; pay[5] = q1.y; pay[6] = q1.z; pay[7] = q1.w;
    if (!any) return -1;
    if (NARROW) {
        const uint32_t t = pay[7] >> 16;
        const int cnt = int((t >> 11) & 15u);
        return t == store_tag16(pay, cnt) ? cnt : -1;
    }
    const int cnt = int((pay[7] >> 24) & 0x7Fu);
    return (pay[7] == store_tag32(pay, cnt) && cnt <= kStoreIds32) ? cnt : -1;
}
template <bool NARROW>
__device__ __forceinline__ int32_t store_id(const uint32_t (&pay)[8], int k) {  // k: compile-time after unrolling
    return NARROW ? int32_t((pay[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) : int32_t(pay[k]);
}
// Files a piece under its key in the first of its kStoreWays slots that is free; `pay` carries the ids, the tag is added
// here.  The caller has just looked the piece up and missed, so nothing is read first: one CAS per way tried.  (The first
// version read every candidate with agent-scope loads before claiming one and kept an exact count of the room with a
// returning atomic: five dependent round trips behind every merge chain -- with a store that still had room merge_kernel
// took 78 us instead of 57.)  A slot that is being written, or holds this very piece (filed by another wave a moment ago: its
// last key dword says whether that is worth checking), ends the attempt.
template <bool NARROW>
__device__ __forceinline__ bool store_insert(const PieceStoreDev& S, const uint32_t (&key)[8], uint32_t (&pay)[8], int cnt) {
    if (NARROW) pay[7] = (pay[7] & 0xFFFFu) | (store_tag16(pay, cnt) << 16);
    else pay[7] = store_tag32(pay, cnt);
    const uint32_t mix = store_mix(key);
    uint32_t* slot = nullptr;
    for (int step = 0; step <= kStoreOverflow && !slot; ++step) {   // the piece's line, then the lines behind it while they are full
#pragma unroll
        for (int c = 0; c < kStoreWays; ++c) {
            if (slot) break;
            uint32_t* cand = reinterpret_cast<uint32_t*>(S.slots + store_h_next(store_h(mix, c, S.shift), step, S.shift));
            const uint32_t old = atomicCAS(cand + 7, 0u, kPieceBusy);
            if (old == 0u) {
                slot = cand;
            } else if (old == kPieceBusy) {
                return false;
            } else if (old == key[7]) {   // same length (and tail): this very piece?  (only then are its other dwords read)
                bool same = true;
#pragma unroll
                for (int j = 0; j < 7; ++j) same = same && __hip_atomic_load(cand + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key[j];
                if (same) return false;
            }
        }
    }
    if (!slot) return false;
    *reinterpret_cast<uint4*>(slot + 8) = uint4{pay[0], pay[1], pay[2], pay[3]};
    *reinterpret_cast<uint4*>(slot + 12) = uint4{pay[4], pay[5], pay[6], pay[7]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the payload has completed at the L2 before the key that makes it reachable is sent (memo_insert says why not agent scope; the tag's checksum stays as the second line of defence)
    *reinterpret_cast<uint4*>(slot) = uint4{key[0], key[1], key[2], key[3]};
    *reinterpret_cast<uint4*>(slot + 4) = uint4{key[4], key[5], key[6], key[7]};  // (replaces kPieceBusy: the entry is complete)
    return true;
}

// ---- lookup kernel ------------------------------------------------------------------------------
struct RowState {
    int base;      // staging offset of the row
    int used;      // staging entries handed out so far
    int emitted;   // ids written so far (deferred pieces excluded)
    int row;
};

struct WaveMiss {
    DeferredPiece e[kMissBuf];
};

// Writes the first `n` (<= 64) buffered pieces to this block's shard of the deferred list, moves the rest up.
__device__ __forceinline__ void flush_misses(WaveMiss& mb, int& n_miss, int n, const EncodeWork& w) {
    const int l = lane_id();
    const int shard = w.small ? 0 : int(blockIdx.x) % kShards;  // (a small batch: one dense list for the one block that merges)
    int idx = 0;
    if (l == 0) idx = atomicAdd(&w.status->shard_count[shard * kCounterStride], n);
    idx = wave_readlane(idx, 0);
    if (l < n) {
        if (idx + l < w.shard_cap) w.deferred[(long long)shard * w.shard_cap + idx + l] = mb.e[l];
        else atomicOr(&w.status->flags, kFlagDeferOverflow);
    }
    const int rest = n_miss - n;
    // move the remaining entries up (through scalars: a struct temporary would live in scratch across wave_sync)
    uint64_t t0 = 0, t1 = 0;
    int t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    if (l < rest) {
        const DeferredPiece& src = mb.e[n + l];
        t0 = src.k0; t1 = src.k1; t2 = src.stage_pos; t3 = src.row; t4 = src.begin; t5 = src.len;
    }
    wave_sync();
    if (l < rest) mb.e[l] = DeferredPiece{t0, t1, t2, t3, t4, t5};
    wave_sync();
    n_miss = rest;
}

// One batch: lane l holds piece (raw bytes r0/r1 if plen <= 15, plen, abs_begin) when `valid`.
__device__ __forceinline__ void lookup_batch(const BpeDev& T, RowState& st, const EncodeWork& w, WaveMiss& mb, int& n_miss,
                                             bool valid, uint64_t r0, uint64_t r1, int plen, int abs_begin) {
    const int SL = T.suffix_len;
    int32_t tok[kPieceMaxIds] = {0, 0, 0};
    int cnt = -1;
    uint64_t k0 = 0, k1 = 0;
    if (valid && plen >= 1 && plen <= kPieceKeyBytes) {
        piece_key(r0, r1, plen, k0, k1);
        if (T.pieces.slots) cnt = memo_lookup(T.pieces, k0, k1, tok);
    }
    const bool hit = cnt >= 0;
    const int need = valid ? (hit ? cnt : plen + SL) : 0;
    const int incl = wave_incl_sum(need);
    const int pos = st.base + st.used + incl - need;
    if (hit) {
#pragma unroll
        for (int k = 0; k < kPieceMaxIds; ++k)
            if (k < cnt) stage_put(w, pos + k, tok[k]);   // (not a streaming store: compact_kernel reads it back within microseconds -- measured: 21.7 -> 26.5 us for compact with the hint)
    }
    const bool miss = valid && !hit;
    const unsigned long long mm = __ballot(miss);
    if (mm) {
        if (n_miss + __popcll(mm) > kMissBuf) flush_misses(mb, n_miss, n_miss, w);  // (rare: a batch with > 32 misses)
        if (miss) mb.e[n_miss + rank_below(mm)] = DeferredPiece{k0, k1, pos, st.row, abs_begin, plen};
        n_miss += __popcll(mm);
        wave_sync();
        if (n_miss >= kWave) flush_misses(mb, n_miss, kWave, w);
    }
    st.used += wave_readlane(incl, kWave - 1);
    // ids written by the hits: cnt is 0..3, summed as two bit-planes of ballots
    const int c = hit ? cnt : 0;
    st.emitted += __popcll(__ballot(c & 1)) + 2 * __popcll(__ballot(c & 2));
}

enum EncodeMode : int { kFused = 0, kPieces = 1, kFusedLlama3 = 2, kFusedSeq = 3 };  // kFusedLlama3: kFused compiled for the Llama-3 scanner;
                                                                               // kFusedSeq: for the literal matchers of span_fam.hpp's families

// Whole strings as pieces (kPieces mode, and skipped strings of the fused mode): cols [c_begin, c_end).
__device__ __forceinline__ void lookup_whole_strings(const BpeDev& T, RowState& st, const EncodeWork& w, WaveMiss& mb,
                                                     int& n_miss, const RowsIn& in, int c_begin, int c_end) {
    const int l = lane_id();
    for (int col = c_begin; col < c_end; col += kWave) {
        const int my = col + l;
        const bool valid = my < c_end;
        int sb = 0, plen = 0;
        uint64_t r0 = 0, r1 = 0;
        if (valid) {
            sb = in.begins[my];
            plen = in.ends[my] - sb;
            if (plen >= 1 && plen <= kPieceKeyBytes) global_bytes16(in.chars + sb, plen, r0, r1);
        }
        lookup_batch(T, st, w, mb, n_miss, valid, r0, r1, plen, sb);
    }
}

// Header of a row (scalar loads).  simple: exactly one string, not skipped, at most kChunk bytes long, offsets inside
// the chars tensor.  (A software pipeline that fetched the next row's header and text into registers while the
// current row was processed was measured: it cost 25 VGPRs = one wave per SIMD and was a net loss.)
struct RowHdr {
    int cb, ce, sb, slen;
    bool simple;
};
__device__ __forceinline__ RowHdr load_row_range(const RowsIn& in, int row) {
    RowHdr h{0, 0, 0, 0, false};
    if (row < in.n_rows) {
        h.cb = uniform_load(in.ragged_begins + row);
        h.ce = uniform_load(in.ragged_ends + row);
    }
    return h;
}
__device__ __forceinline__ RowHdr load_row_string(const RowsIn& in, RowHdr h) {
    if (h.ce == h.cb + 1 && h.cb >= 0 && h.cb < in.n_strings && !(in.skips && uniform_load(in.skips + h.cb))) {
        h.sb = uniform_load(in.begins + h.cb);
        h.slen = uniform_load(in.ends + h.cb) - h.sb;
        // (offsets that leave the chars tensor make the row "not simple": the generic path reports them)
        h.simple = h.slen > 0 && h.slen <= kChunk && h.sb >= 0 && (long long)h.sb + h.slen <= in.n_chars;
    }
    return h;
}

constexpr int kStageChunk = 4096;  // staging entries a wave takes from its allocator at a time
// Staging capacity of a row (mul * sum of max(len, 1)) with the offset validation prep_rows_kernel does otherwise;
// -1 (and kFlagRange) for offsets that leave their tensors.  Wave-uniform result.
__device__ __forceinline__ int row_capacity_checked(const RowsIn& in, const RowHdr& h, int mul, RunStatus* status) {
    const int l = lane_id();
    bool bad = h.cb < h.ce && (h.cb < 0 || h.ce > in.n_strings);
    long long cap = 0;
    if (!bad) {
        if (h.simple) {
            bad = h.sb < 0 || (long long)h.sb + h.slen > in.n_chars;
            cap = (long long)h.slen * mul;
        } else {
            for (int col = h.cb + l; col < h.ce; col += kWave) {
                const long long sb = in.begins[col], se = in.ends[col];
                if (sb < 0 || se < sb || se > in.n_chars) bad = true;
                else cap += (se - sb > 0 ? se - sb : 1) * mul;
            }
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) cap += __shfl_xor(cap, d);
        }
    }
    if (__ballot(bad) || cap > INT32_MAX / 2) {
        if (l == 0) atomicOr(&status->flags, kFlagRange);
        return -1;
    }
    return int(cap);
}

template <int MODE, bool TICKETS = false>
__device__ __forceinline__ void lookup_body(const RowsIn& in, const SplitDev& sp, const BpeDev& T, const EncodeWork& w) {
    __shared__ WaveScratch ws_all[kWavesPerBlock];
    __shared__ WaveMiss miss_all[kWavesPerBlock];
    if (w.status->flags & kFatalFlags) return;
    if (w.only_pending && w.status->n_pending == 0) return;  // the span / rows kernel took every row
    WaveScratch& ws = ws_all[wave_in_block()];
    WaveMiss& mb = miss_all[wave_in_block()];
    const int l = lane_id();
    int n_miss = 0;
    const int n_waves = w.n_waves;  // == gridDim.x * kWavesPerBlock: the geometry prep_rows_kernel summed over
    const int wave = wave_uniform(int(blockIdx.x) * kWavesPerBlock + wave_in_block());
    // rows of this wave are staged back to back: in its arena (prep_rows_kernel ran), or in chunks it takes from one
    // of kShards bump allocators (no prep launch; offsets are validated here)
    const bool alloc = w.wave_off == nullptr;
    int cursor = alloc ? 0 : int(w.wave_off[wave]), limit = alloc ? 0 : INT32_MAX;
    bool dead = false;  // allocator mode: staging exhausted or bad offsets -- the host reruns / reports
    // Which rows a wave takes.  Static (w.rows_per_ticket == 0): rows wave, wave + n_waves, ... -- the share
    // prep_rows_kernel sized its arena for, and the cheapest when the GPU is ours alone.  Dynamic (allocator mode
    // only): rows are handed out rows_per_ticket at a time -- a block that becomes resident late (another stream's
    // kernel, e.g. RCCL's all-gather, sits on its CU) then simply takes fewer, instead of running its whole share
    // after everybody else (+30 % kernel time measured with 16 CUs taken; the tickets cost +5 %).  The tickets
    // form kShards ranges with one counter each (a single device-scope counter saturates at ~90 atomics/us); a wave
    // works through its home range, then through the ranges not yet marked done.  The NEXT ticket is requested
    // before the current rows' header loads are issued, so most of its latency is paid together with theirs.
    const int rpt = TICKETS ? w.rows_per_ticket : 0;  // a template flag: the static kernel carries none of this
    const int n_tickets = rpt ? (in.n_rows + rpt - 1) / (rpt ? rpt : 1) : 0;
    const int per = (n_tickets + kShards - 1) / kShards;
    int tk_shard = wave % kShards, tk_pend = 0;
    uint32_t tk_seen = 0;  // ranges this wave found empty
    auto tk_issue = [&]() {
        if (l == 0) tk_pend = atomicAdd(&w.status->row_ticket[tk_shard * kCounterStride], 1);
    };
    auto tk_resolve = [&]() -> int {  // -> first row of the ticket, or -1
        for (;;) {
            const int idx = wave_readlane(tk_pend, 0);
            const int base = tk_shard * per;
            const int size = n_tickets - base < per ? n_tickets - base : per;
            if (idx < size) return (base + idx) * rpt;
            if (idx == size || idx == size + 1) {  // the first waves to find the range empty say so
                if (l == 0) atomicOr(&w.status->rows_done, 1u << tk_shard);
            }
            tk_seen |= 1u << tk_shard;
            uint32_t done = __hip_atomic_load(&w.status->rows_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | tk_seen;
            done = uint32_t(wave_uniform(int(done)));
            int next = -1;
            for (int k = 1; k < kShards && next < 0; ++k) {
                const int cand = (tk_shard + k) % kShards;
                if (!(done >> cand & 1u)) next = cand;
            }
            if (next < 0) return -1;
            tk_shard = next;
            tk_issue();
        }
    };
    int row = wave, chunk_end = 0;
    if (rpt) {
        tk_issue();
        row = tk_resolve();
        chunk_end = row + rpt;
    }
    // only_pending with a list: wave k takes entries k, k + n_waves, ... of it (a few thousand rows of a batch of 131 072: walking
    // every row's flag cost the Llama-3 configuration 27 us)
    const bool listed = !TICKETS && w.only_pending && w.pending_rows;
    const int n_listed = listed ? w.status->n_pending : 0;
    int list_at = wave;
    if (listed) row = list_at < n_listed ? uniform_load(w.pending_rows + list_at) : -1;
    while (row >= 0 && row < in.n_rows) {
        if (!TICKETS && !listed && w.only_pending && uniform_load(w.row_used + row) != kRowPending) {  // done by the span / rows kernel
            row += n_waves;
            continue;
        }
        if (rpt && row + rpt == chunk_end) tk_issue();  // first row of a ticket
        RowHdr h{0, 0, 0, 0, false};
        if (MODE != kPieces) h = load_row_string(in, load_row_range(in, row));
        else h = load_row_range(in, row);
        if (alloc) {
            const int cap = dead ? 0 : row_capacity_checked(in, h, T.suffix_len + 1, w.status);
            if (cap < 0) dead = true;
            if (!dead && cursor + cap > limit) {
                const int size = cap > kStageChunk ? cap : kStageChunk;
                const int shard = wave % kShards;
                int base = 0;
                if (l == 0) base = atomicAdd(&w.status->stage_top[shard * kCounterStride], size);
                base = wave_readlane(base, 0);
                if (base < 0 || base > w.stage_region - size) {
                    if (l == 0) atomicOr(&w.status->flags, kFlagStageOverflow);
                    dead = true;
                } else {
                    cursor = shard * w.stage_region + base;
                    limit = cursor + size;
                }
            }
        }
        RowState st{cursor, 0, 0, row};
        if (dead) {
            // nothing is staged for this row
        } else if (MODE == kPieces) {
            lookup_whole_strings(T, st, w, mb, n_miss, in, h.cb, h.ce);
        } else {
            for (int col = h.cb; col < h.ce; ++col) {
                if (!h.simple && in.skips && in.skips[col]) {  // regex_split.cpp:231-234: passes through unsplit
                    lookup_whole_strings(T, st, w, mb, n_miss, in, col, col + 1);
                    continue;
                }
                const int sb = h.simple ? h.sb : in.begins[col];
                const int slen = h.simple ? h.slen : in.ends[col] - sb;
                scan_string<(MODE == kFusedLlama3 ? kScanLlama3 : (MODE == kFusedSeq ? kScanFamLiteral : kScanGeneric))>(
                    ws, sp, in.chars + sb, slen, in.chars, in.chars + in.n_chars,
                    [&](int np, int c0, int w0, int skew) {
                        for (int jb = 0; jb < np; jb += kWave) {
                            const int j = jb + l;
                            bool valid = j < np;
                            int ps = 0, plen = 0;
                            uint64_t r0 = 0, r1 = 0;
                            if (valid) {
                                ps = c0 + int(ws.pstart[j] & kPiecePosMask);
                                plen = c0 + int(ws.pstart[j + 1] & kPiecePosMask) - ps;
                                valid = !(ws.pstart[j] & kPieceDropped);
                                if (valid && plen <= kPieceKeyBytes) lds_bytes16(ws.text_w, kTextPad + ps - w0 + skew, r0, r1);
                            }
                            lookup_batch(T, st, w, mb, n_miss, valid, r0, r1, plen, sb + ps);
                        }
                    },
                    [&](int b, int e, bool dropped) {  // a piece longer than the scan window: straight to the deferred list
                        lookup_batch(T, st, w, mb, n_miss, l == 0 && !dropped, 0, 0, e - b, sb + b);
                    });
            }
        }
        if (l == 0) {
            w.row_stage[row] = cursor;
            w.row_cnt[row] = st.emitted;
            if (w.row_emit) w.row_emit[row] = st.emitted;  // (an atomic into tile_cnt here would stall every row: vmcnt is in order)
            w.row_used[row] = st.used;
            if (w.span_sums && st.emitted) atomicAdd(&w.tile_cnt[row / kRowTile], st.emitted);   // (the short path: the few rows the span kernel left)
        }
        cursor += st.used;
        if (listed) {
            list_at += n_waves;
            row = list_at < n_listed ? uniform_load(w.pending_rows + list_at) : -1;
        } else if (!rpt) {
            row += n_waves;
        } else if (++row == chunk_end) {
            row = tk_resolve();
            chunk_end = row + rpt;
        }
    }
    if (n_miss > 0) flush_misses(mb, n_miss, n_miss, w);
}
template <int MODE, bool TICKETS = false>
static __global__ __launch_bounds__(kBlockThreads, MODE == kFusedLlama3 ? 4 : 5) void lookup_kernel(RowsIn in, SplitDev sp, BpeDev T, EncodeWork w) {
    lookup_body<MODE, TICKETS>(in, sp, T, w);
}


// ---- lookup_rows_kernel: one row per scan, the rows of a wave consecutive (round 3).  The Llama-3 family, the fused WordPiece path
// (BERT words) and GPT-2-family handles without a memo run it; GPT-2-family handles with a memo take lookup_span_kernel
// (span_kernel.hpp: several rows per scan).  A row that is ONE string that is ONE scan window is scanned with the packed-byte
// scanner of its pattern and looked up in 64-piece batches; any other row is marked kRowPending in row_used, listed in
// pending_rows, and left to lookup_kernel<kFused> (launched right behind with only_pending set; it returns at once when nothing
// was left).  A wave owns rows_per_wave CONSECUTIVE rows:
//  * their headers arrive with ONE vector load (lane i = row i of the range), two dependent round trips per wave instead of
//    two per row;
//  * the text of row i + 1 is requested while row i is scanned, straight into LDS (global_load_lds: no registers in between),
//    into the second of two windows -- by the time row i's batches are through it has landed.
// Staging chunks come from the bump allocators.
__device__ __forceinline__ void lds_fetch_words(const uint8_t* ga, int nwords, uint32_t* dst) {
    const int l = lane_id();
#if defined(OVTK_SIMT_EMULATOR)
    for (int k = l; k < nwords; k += kWave) dst[k] = *reinterpret_cast<const uint32_t*>(ga + 4 * k);
#else
    // one instruction per 64 dwords: lane l's dword goes to dst[64 c + l] (the LDS address is the wave-uniform base + 4 l)
    for (int c = 0; c * kWave < nwords; ++c)
        if (c * kWave + l < nwords)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(ga) + c * kWave + l,
                                             (__attribute__((address_space(3))) uint32_t*)(dst + c * kWave), 4, 0, /*aux: nt*/ 2);
#endif
}
// SCAN: which packed-byte scanner reads the window -- the GPT-2 rules, the same with every digit on its own, or the BERT words
// (white space dropped, every delimiter character a word: the fused WordPiece path; `T` then holds nothing but the word memo).
enum RowsScan : int { kRowsGpt2 = 0, kRowsGpt2Digits = 1, kRowsBertWords = 2, kRowsLlama3 = 3 };
template <int SCAN>
static __global__ __launch_bounds__(kBlockThreads, SCAN == kRowsLlama3 ? 4 : 6) void lookup_rows_kernel(RowsIn in, SplitDev sp, BpeDev T, EncodeWork w) {
    __shared__ uint32_t text_all[kWavesPerBlock][2][kWinBytes / 4];
    __shared__ uint16_t pstart_all[kWavesPerBlock][kChunk + 2];
    __shared__ WaveMiss miss_all[kWavesPerBlock];
    if (w.status->flags & kFatalFlags) return;
    WaveMiss& mb = miss_all[wave_in_block()];
    const int l = lane_id();
    const int wave = wave_uniform(int(blockIdx.x) * kWavesPerBlock + wave_in_block());
    const int R = w.rows_per_wave;  // <= kWave
    const int row0 = wave * R;
    if (row0 >= in.n_rows) return;
    const int nr = in.n_rows - row0 < R ? in.n_rows - row0 : R;
    const int mul = T.suffix_len + 1;
    // ---- the headers of all my rows: lane i = row row0 + i
    int h_sb = 0, h_len = 0;
    bool h_simple = false;
    if (l < nr) {
        const int cb = in.ragged_begins[row0 + l], ce = in.ragged_ends[row0 + l];
        if (ce == cb + 1 && cb >= 0 && cb < in.n_strings && !(in.skips && in.skips[cb])) {
            h_sb = in.begins[cb];
            h_len = in.ends[cb] - h_sb;
            h_simple = h_len > 0 && h_len <= kChunk && h_sb >= 0 && (long long)h_sb + h_len <= in.n_chars;
        }
    }
    const unsigned skew0 = unsigned(reinterpret_cast<uintptr_t>(in.chars) & 3u);
    // a window may be requested ahead when its aligned dwords lie inside the chars tensor (all but the tensor's last row or so)
    unsigned long long simple_m, ahead_m;
    {
        const int h_skew = int((skew0 + unsigned(h_sb)) & 3u);
        const int h_nwords = (h_skew + h_len + 3) >> 2;
        const bool h_ahead = h_simple && (long long)h_sb - h_skew + 4ll * h_nwords <= in.n_chars;
        simple_m = __ballot(h_simple);
        ahead_m = __ballot(h_ahead);
    }
    auto request = [&](int i) {  // row i's text -> window i & 1
        const int sb = wave_readlane(h_sb, i), slen = wave_readlane(h_len, i);
        const int skew = int((skew0 + unsigned(sb)) & 3u), nwords = (skew + slen + 3) >> 2;
        lds_fetch_words(in.chars + sb - skew, nwords, text_all[wave_in_block()][i & 1] + kTextPad / 4);
    };
    int n_miss = 0, n_pending = 0;
    int cursor = 0, limit = 0;
    bool dead = false;  // staging exhausted: the host grows the buffer and reruns
    // the rows' records (staging offset, ids, entries used) collect in lane i for row i and leave with ONE store per array at the
    // end: four 4-byte stores per row from lane 0 were four partly written 32-byte sectors per row
    int rec_stage = 0, rec_cnt = 0, rec_used = 0;
    if (ahead_m & 1ull) request(0);
    for (int i = 0; i < nr; ++i) {
        const int row = row0 + i;
        const int sb = wave_readlane(h_sb, i), slen = wave_readlane(h_len, i);
        const int skew = int((skew0 + unsigned(sb)) & 3u);
        WsView ws{text_all[wave_in_block()][i & 1], pstart_all[wave_in_block()]};
        bool fast = ((simple_m >> i) & 1ull) != 0 && !dead;
        int np = 0;
        if (fast) {
            // what was requested one row ago has had that row's batches to arrive; nothing else of this wave is in flight
            drain_vmem();
            wave_sync();
            if ((ahead_m >> i) & 1ull) {
                // the string's own bytes only: those in front of it and behind it in its first / last dword are staged as zeros
                if (l == 0) {
                    const int nwords = (skew + slen + 3) >> 2;
                    uint32_t* t = ws.text_w + kTextPad / 4;
                    ws.text_w[0] = 0;  // kTextPad
                    t[0] &= ~((1u << (8 * skew)) - 1u);
                    const int keep = skew + slen - 4 * (nwords - 1);  // bytes of the last dword that belong to the string: 1..4
                    if (keep < 4) t[nwords - 1] &= (1u << (8 * keep)) - 1u;
                }
            } else {
                stage_window(ws, in.chars + sb, slen, 0, slen, in.chars, in.chars + in.n_chars);
            }
            wave_sync();
        }
        // the next row's text, under this row's scan and batches (its window is free: row i - 1 is done with it)
        if (i + 1 < nr && ((ahead_m >> (i + 1)) & 1ull) && !dead) request(i + 1);
        if (fast) {
            if (SCAN == kRowsLlama3) {
                // the packed form of the Llama-3 rules decides the whole string, or says that it cannot (a window that needs the
                // ballot form or the literal matcher: a non-ASCII digit, nine digits in a row ...): the generic kernel's business
                int und = 0;
                bool seq = false;
                fast = slen <= 512 ? llama3_packed_starts<2>(ws, sp, skew, slen, 0, slen, true, np, und, seq)
                                   : llama3_packed_starts<3>(ws, sp, skew, slen, 0, slen, true, np, und, seq);
                if (seq || und < slen) fast = false;
            } else if (SCAN == kRowsBertWords)
                fast = slen <= 256 ? class_packed_starts<1>(ws, sp, skew, slen, 0, slen, np)
                                   : (slen <= 512 ? class_packed_starts<2>(ws, sp, skew, slen, 0, slen, np)
                                                  : class_packed_starts<kLaneDwords>(ws, sp, skew, slen, 0, slen, np));
            else
                fast = slen <= 64 * 4 * (kLaneDwords - 1)
                           ? gpt2_packed_starts<kLaneDwords - 1>(ws, skew, slen, SCAN == kRowsGpt2Digits, 0, slen, np)
                           : gpt2_packed_starts<kLaneDwords>(ws, skew, slen, SCAN == kRowsGpt2Digits, 0, slen, np);
        }
        if (fast) {
            const int cap = slen * mul;
            if (cursor + cap > limit) {
                const int size = cap > kStageChunk ? cap : kStageChunk;
                const int shard = wave % kShards;
                int base = 0;
                if (l == 0) base = atomicAdd(&w.status->stage_top[shard * kCounterStride], size);
                base = wave_readlane(base, 0);
                if (base < 0 || base > w.stage_region - size) {
                    if (l == 0) atomicOr(&w.status->flags, kFlagStageOverflow);
                    dead = true;
                    fast = false;
                } else {
                    cursor = shard * w.stage_region + base;
                    limit = cursor + size;
                }
            }
        }
        if (!fast) {
            if (l == i) rec_used = kRowPending;
            ++n_pending;
            continue;
        }
        if (l == 0) ws.pstart[np] = uint16_t(slen);
        wave_sync();
        RowState st{cursor, 0, 0, row};
        for (int jb = 0; jb < np; jb += kWave) {
            const int j = jb + l;
            bool valid = j < np;
            int ps = 0, plen = 0;
            uint64_t r0 = 0, r1 = 0;
            if (valid) {
                if (SCAN == kRowsBertWords || SCAN == kRowsLlama3) {   // (the class scanners flag the pieces RegexSplit drops: white space)
                    const uint32_t p0 = ws.pstart[j];
                    ps = int(p0 & kPiecePosMask);
                    plen = int(ws.pstart[j + 1] & kPiecePosMask) - ps;
                    valid = !(p0 & kPieceDropped);
                } else {
                    ps = int(ws.pstart[j]);
                    plen = int(ws.pstart[j + 1]) - ps;
                }
                if (valid && plen <= kPieceKeyBytes) lds_bytes16(ws.text_w, kTextPad + ps + skew, r0, r1);
            }
            lookup_batch(T, st, w, mb, n_miss, valid, r0, r1, plen, sb + ps);
        }
        if (l == i) {
            rec_stage = cursor;
            rec_cnt = st.emitted;
            rec_used = st.used;
        }
        cursor += st.used;
    }
    if (l < nr) {
        w.row_used[row0 + l] = rec_used;
        if (rec_used != kRowPending) {
            w.row_stage[row0 + l] = rec_stage;
            w.row_cnt[row0 + l] = rec_cnt;
            if (w.row_emit) w.row_emit[row0 + l] = rec_cnt;
        }
    }
    if (n_miss > 0) flush_misses(mb, n_miss, n_miss, w);
    if (n_pending) {   // (wave-uniform) the rows left to the generic kernel: counted, and listed for it
        int base = 0;
        if (l == 0) base = atomicAdd(&w.status->n_pending, n_pending);
        base = wave_readlane(base, 0);
        const unsigned long long pm = __ballot(l < nr && rec_used == kRowPending);
        if (w.pending_rows && ((pm >> l) & 1ull)) w.pending_rows[base + rank_below(pm)] = row0 + l;
    }
}

// ---- path X, one lane per piece.
__device__ __forceinline__ void exact_piece(const RowsIn& in, const BpeDev& T, const EncodeWork& w, const ExactPiece& p) {
    const int SL = T.suffix_len;
    const int ntext = p.len + SL;
    // (+ ntext i32 ids behind the working arrays when the staging entries are u16)
    const uint32_t bytes = ((bpe_exact_scratch_bytes(uint32_t(ntext)) + 15u) & ~15u) + (w.stage16 ? 4u * ((uint32_t(ntext) + 3u) & ~3u) : 0u);
    const uint32_t off = atomicAdd(&w.status->scratch_used, bytes);
    if (off > w.scratch_cap || bytes > w.scratch_cap - off) {
        atomicOr(&w.status->flags, kFlagScratchOverflow);
        return;
    }
    const uint8_t* text = in.chars + p.begin;
    // (bpe_exact_piece writes i32 ids: straight into the staging entries, or -- u16 staging -- into the words behind its scratch)
    int32_t* out = w.stage16 ? reinterpret_cast<int32_t*>(w.scratch + off + bytes - 4u * ((uint32_t(ntext) + 3u) & ~3u)) : w.stage + p.stage_pos;
    const int cnt = bpe_exact_piece(
        T, [&](int k) -> uint32_t { return k < p.len ? text[k] : T.suffix[k - p.len]; }, ntext, w.scratch + off, out);
    if (w.stage16) {
        for (int k = 0; k < cnt; ++k) stage_put(w, p.stage_pos + k, out[k]);
    }
    for (int k = cnt; k < ntext; ++k) stage_clear(w, p.stage_pos + k);
    if (cnt) {
        atomicAdd(&w.row_cnt[p.row], cnt);
        if (w.tile_cnt) atomicAdd(&w.tile_cnt[p.row / kRowTile], cnt);
    }
}

__device__ __forceinline__ void exact_one(const RowsIn& in, const BpeDev& T, const EncodeWork& w, int i) { exact_piece(in, T, w, w.exact[i]); }

// Final offsets by ONE block (the folded tail of merge_kernel) -- count_scan_kernel without its launch: the per-tile
// sums were accumulated in w.tile_cnt while the ids were produced, so only their scan is left.
__device__ __forceinline__ void scan_tiles_one_block(int n_rows, const EncodeWork& w, long long out_cap) {
    const int n_tiles = (n_rows + kRowTile - 1) / kRowTile;
    const long long total = block_exclusive_scan<kBlockThreads, 16>(
        n_tiles, [&](int i) -> long long { return w.tile_cnt[i]; }, [&](int i, long long off) { w.tile_off[i] = off; });
    if (threadIdx.x == 0) {
        w.status->n_out = total > INT32_MAX ? INT32_MAX : int32_t(total);
        if (total > out_cap) atomicOr(&w.status->flags, kFlagOutCapacity);
    }
}

// Start of a kernel that folds the row scan into its end (merge_kernel, wordpiece_deferred_kernel): tile sums of what the
// lookup kernel emitted, one wave per tile, spread over the whole grid (2-D grids: x fastest).
__device__ __forceinline__ void fold_emitted_tile_sums(const EncodeWork& w, int tail_rows, bool solo = false) {
    const int n_tiles = (tail_rows + kRowTile - 1) / kRowTile;
    const int wave = solo ? wave_in_block() : (int(blockIdx.y) * int(gridDim.x) + int(blockIdx.x)) * kWavesPerBlock + wave_in_block();
    const int n_waves = solo ? kWavesPerBlock : int(gridDim.x) * int(gridDim.y) * kWavesPerBlock;
    for (int tile = wave; tile < n_tiles; tile += n_waves) {
        const int row = tile * kRowTile + lane_id();
        const int s0 = wave_sum(row < tail_rows ? w.row_emit[row] : 0);
        if (lane_id() == 0 && s0) atomicAdd(&w.tile_cnt[tile], s0);
    }
}

// ---- merge kernel: dense batches of deferred pieces ----------------------------------------------
// tail_rows > 0: the block that finishes last also runs the exact pieces (when few) and the scan of the row counts, so
// that exact_kernel and count_scan_kernel need no launches of their own (tail_rows = n_rows, out_cap as for count_scan).
// NARROW: every id < 65536 -- path F keeps ids and merged ids as u16 (8 KB of LDS per wave instead of 12: 4 resident
// blocks per CU instead of 3, and the kernel's time follows its occupancy).
// A wave's LDS: path F  key u32[16*64] | id IdT[16*64] | nid IdT[16*64];  path W  key u64[512] | id u32[512].
// solo: this block is the only one at work (the last block of encode_small_kernel): every shard is its, no ticket.
template <bool NARROW>
__device__ __forceinline__ void merge_body(const RowsIn& in, const BpeDev& T, const EncodeWork& w, int tail_rows, long long out_cap,
                                           bool solo = false) {
    using IdT = typename std::conditional<NARROW, uint16_t, uint32_t>::type;
    constexpr int kWaveLdsBytes = kFastSyms * kWave * (4 + 2 * int(sizeof(IdT)));
    static_assert(kWaveLdsBytes >= kChunkSyms * 12, "path W's key u64[] + id u32[] must fit the wave's LDS");
    __shared__ uint64_t lds_all[kWavesPerBlock][kWaveLdsBytes / 8];
    static_assert(kLongSyms * (kWave / 2) == kFastSyms * kWave, "path L reuses path F's LDS arrays: 32 lanes x 32 symbols");
    __shared__ I2 root_lds[256];
    __shared__ uint8_t long_src_all[kWavesPerBlock][kWave / 2];  // path L: source lane of the piece lane t works on
    __shared__ int pushed_exact;  // this block stored exact-list entries (plain stores the tail block must see)
    // Everything the kernel has to know before it can start, asked for together (as wordpiece_deferred_kernel does): most of its
    // blocks find no batch, and the ones that do are one chain of memory round trips from here to the kernel's end -- the tables'
    // roots, the flags, the shard's count and the store's room used to be one wait each.
    const uint32_t flags0 = __hip_atomic_load(&w.status->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int shard0 = solo ? 0 : int(blockIdx.x);
    const int count0 = __hip_atomic_load(&w.status->shard_count[shard0 * kCounterStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int store_room0 = T.store.slots ? __hip_atomic_load(T.store.room, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    for (int i = int(threadIdx.x); i < 256; i += kBlockThreads) root_lds[i] = T.trie.root[i];
    if (threadIdx.x == 0) pushed_exact = 0;
    __syncthreads();
    PROBE(0);
    if (flags0 & (kFatalFlags | kFlagDeferOverflow)) return;
    if (tail_rows > 0 && !w.span_sums) fold_emitted_tile_sums(w, tail_rows, solo);   // (span_sums: the lookup kernels have summed their rows themselves)
    PROBE(1);
    uint64_t* key = lds_all[wave_in_block()];                                 // path W
    uint32_t* id = reinterpret_cast<uint32_t*>(key + kChunkSyms);             // path W
    uint32_t* fkey = reinterpret_cast<uint32_t*>(key);                        // path F
    IdT* fid = reinterpret_cast<IdT*>(fkey + kFastSyms * kWave);              // path F
    IdT* fnid = fid + kFastSyms * kWave;                                      // path F
    const int l = lane_id();
    const int SL = T.suffix_len;
    // The piece store takes entries while its room counter is positive.  Every wave reads the counter ONCE (a look per batch is a
    // round trip on every batch's chain) -- and until round 5 a wave that found it positive filed for its whole life: a launch whose
    // ten million pieces are all new (uniform random text) left a store made for 200 000 entries with every one of its million slots
    // taken, and every later lookup of a new piece walked the overflow lines of a full table (stress.uniform_text 67.6 -> 59.4 GB/s).
    // Now the room a wave finds is shared out: a wave files at most its share of it, so a LAUNCH files at most the room it found,
    // whatever its text (launches that overlap on several streams may each find the same room: a small multiple at worst).
    int store_budget = 0;
    if (T.store.slots) {
        const int room = wave_uniform(store_room0);
        const int n_waves_grid = (solo ? 1 : int(gridDim.x) * int(gridDim.y)) * kWavesPerBlock;
        store_budget = room > 0 ? (room + n_waves_grid - 1) / n_waves_grid : 0;
    }
    {  // (x is the fastest-varying block index: blocks that become resident late are spread over all shards; solo: the
       // small batch's blocks filed everything under shard 0)
    const int shard = shard0;
    int count = wave_uniform(count0);
    if (count > w.shard_cap) count = w.shard_cap;
    const DeferredPiece* list = w.deferred + (long long)shard * w.shard_cap;
    // 64-piece batches of the shard: strided over its waves, or (w.rows_per_ticket != 0, see lookup_kernel: late blocks
    // take fewer) handed out by a per-shard ticket, the next one requested before this batch's entries are loaded.
    const bool dyn = w.rows_per_ticket != 0;
    const int stride = (solo ? 1 : int(gridDim.y)) * kBlockThreads;
    int bt_pend = 0, base = ((solo ? 0 : int(blockIdx.y)) * kWavesPerBlock + wave_in_block()) * kWave;
    auto bt_issue = [&]() {
        if (l == 0) bt_pend = atomicAdd(&w.status->batch_ticket[shard * kCounterStride], 1);
    };
    if (dyn) bt_issue();
    for (;; base += stride) {
        if (dyn) {
            base = wave_readlane(bt_pend, 0) * kWave;
            if (base < count) bt_issue();
        }
        if (base >= count) break;
        const bool valid = base + l < count;
        DeferredPiece e{};
        if (valid) e = list[base + l];
        const int need = e.len + SL;
#ifdef OVTK_PROBE
        unsigned long long pt_ = wall_clock64();   // phase clock of this batch
#define PHASE(slot)                                                                                     \
    do {                                                                                                \
        const unsigned long long now_ = wall_clock64();                                                 \
        const int pw_ = (int(blockIdx.y) * int(gridDim.x) + int(blockIdx.x)) * kWavesPerBlock + wave_in_block(); \
        if (lane_id() == 0 && pw_ < 8000) g_ts[pw_][slot] += now_ - pt_;                                \
        pt_ = now_;                                                                                     \
    } while (0)
#else
#define PHASE(slot)
#endif
        int f_cnt = 0;
        // The piece store first (tables.hpp): a piece it holds is one round trip, not a merge chain.
        constexpr int kStoreIds = NARROW ? kStoreIds16 : kStoreIds32;
        const bool keyed = T.store.slots && valid && e.len >= 1 && e.len <= kStoreKeyBytes;
        bool stored = false;
        uint32_t skey[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (keyed) {
            if (e.len <= kPieceKeyBytes) store_key_short(e.k0, e.k1, e.len, skey);
            else store_key_long(in.chars + e.begin, e.len, skey);
            uint32_t pay[8];
            const int c = store_lookup<NARROW>(T.store, skey, pay);
            if (c >= 0) {
                stored = true;
                f_cnt = c;
#pragma unroll
                for (int k = 0; k < kStoreIds; ++k)
                    if (k < c) stage_put(w, e.stage_pos + k, store_id<NARROW>(pay, k));
                for (int k = c; k < need; ++k) stage_clear(w, e.stage_pos + k);
            }
        }
        // A SAMPLE of the waves says what the store did for this call (it decides whether the next calls ask it): one wave in 64.
        // Every wave adding to the same two words was 8 000 adds to one address per launch -- ~90 of those complete per
        // microsecond, and the step went from 0.128 to 0.189 ms.
        if (T.store.slots && wave_in_block() == 0 && ((blockIdx.x + blockIdx.y) & 15) == 0) {
            const unsigned long long pm = __ballot(keyed), hm = __ballot(stored);
            if (l == 0 && pm) {
                atomicAdd(&w.status->n_store_probe, int(__popcll(pm)));
                if (hm) atomicAdd(&w.status->n_store_hit, int(__popcll(hm)));
            }
        }
        // What the batch had to merge is offered to the store while it has room (`store_open`, read once per wave at the
        // kernel's start: the count is kept with returnless adds, nothing here waits for an atomic's answer).
        auto store_offer = [&](bool want, const uint32_t (&key)[8], uint32_t (&pay)[8], int cnt) {
            if (store_budget <= 0 || !__ballot(want)) return;
            const bool added = want && store_insert<NARROW>(T.store, key, pay, cnt);
            const int n_added = __popcll(__ballot(added));
            if (l == 0 && n_added) atomicAdd(T.store.room, -n_added);
            store_budget -= n_added;
        };
        const bool is_f = valid && !stored && e.len >= 1 && e.len <= kPieceKeyBytes && need <= kFastSyms;
        const bool is_l = valid && !stored && !is_f && e.len >= 1 && need <= kLongSyms;
        const bool is_w = valid && !stored && !is_f && !is_l && need <= kChunkSyms;
        bool is_x = valid && !stored && !is_f && !is_l && !is_w;
        bool keep = false;  // path F result short enough for a memo entry
        wave_sync();  // the previous batch is done with the LDS arrays
        if (is_f) {
            const uint64_t k0 = e.k0, k1 = e.k1;
            const int plen = e.len;
            const int n = bpe_symbolize(
                T, root_lds,
                [&](int i) -> uint32_t {
                    if (i >= plen) return T.suffix[i - plen];
                    return uint32_t((i < 8 ? k0 >> (8 * i) : k1 >> (8 * (i - 8))) & 0xFF);
                },
                need, [&](int k, int tok) { fid[k * kWave + l] = IdT(tok); });
            PHASE(2);
            const int res = bpe_merge_lane<IdT>(T, fid, fkey, fnid, n);
            PHASE(3);
            if (res < 0) {
                is_x = true;
            } else {
                for (int k = 0; k < res; ++k) stage_put(w, e.stage_pos + k, int32_t(fid[k * kWave + l]));
                for (int k = res; k < need; ++k) stage_clear(w, e.stage_pos + k);
                f_cnt = res;
                keep = res <= (T.pieces.packed6 ? kPieceMaxIds6 : kPieceMaxIds);
            }
        }
        if (T.store.slots) {
            const bool want = is_f && !is_x && keyed && f_cnt <= kStoreIds;
            uint32_t pay[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (want) {
#pragma unroll
                for (int k = 0; k < kStoreIds; ++k) {
                    const uint32_t v = k < f_cnt ? uint32_t(fid[k * kWave + l]) : 0u;
                    if (NARROW) pay[k >> 1] |= v << (16 * (k & 1));
                    else pay[k] = v;
                }
            }
            store_offer(want, skey, pay, f_cnt);
        }
        // the memo learns the batch's short results while it has room (one atomic per wave takes the room)
        if (T.pieces.room) {
            int32_t* my_room = T.pieces.room + ((blockIdx.x + 5u * blockIdx.y + uint32_t(wave_in_block())) & T.pieces.room_mask) * kRoomStride;
            const unsigned long long km = __ballot(keep);
            int room_now = 0;
            if (l == 0 && km) room_now = __hip_atomic_load(my_room, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wave_readlane(room_now, 0) > 0) {
                int left = 0;
                if (l == 0) left = atomicAdd(my_room, -int(__popcll(km)));
                left = wave_readlane(left, 0);
                const int rank = rank_below(km);
                bool added = false;
                if (keep && rank < left) {
                    int32_t t3[kPieceMaxIds] = {0, 0, 0};
                    if (T.pieces.packed6) {   // (six u16 ids: id 2k in the low half of tok[k])
#pragma unroll
                        for (int k = 0; k < kPieceMaxIds6; ++k)
                            if (k < f_cnt) t3[k >> 1] |= int32_t((uint32_t(fid[k * kWave + l]) & 0xFFFFu) << (16 * (k & 1)));
                    } else {
#pragma unroll
                        for (int k = 0; k < kPieceMaxIds; ++k) t3[k] = k < f_cnt ? int32_t(fid[k * kWave + l]) : 0;
                    }
                    added = memo_insert(T.pieces, e.k0, e.k1, t3, f_cnt);
                }
                const int unused = __popcll(km) - __popcll(__ballot(added));  // room taken but not filled goes back
                if (l == 0 && unused) atomicAdd(my_room, unused);
            }
        }
        // ids of the lane-per-piece results go to row_cnt: the 64 entries of a batch were flushed by ONE wave in
        // row order, so equal rows are contiguous -- one atomic per run of equal rows instead of one per piece
        {
            const int incl = wave_incl_sum(f_cnt);
            const int my_row = valid ? e.row : -1;
            const int prev_row = __shfl_up(my_row, 1), next_row = __shfl_down(my_row, 1);
            const bool head = l == 0 || prev_row != my_row, tail = l == kWave - 1 || next_row != my_row;
            const int seg_base = wave_incl_max(head ? incl - f_cnt : 0);  // prefix before this lane's run (monotone)
            if (valid && tail && incl - seg_base > 0) atomicAdd(&w.row_cnt[e.row], incl - seg_base);
            if (w.tile_cnt) {
                // the same once more per run of equal TILES: with consecutive rows per lookup wave (lookup_rows_kernel) a batch's
                // rows share a tile, and a dozen adds to ONE address in one instruction are served one after the other
                // (merge_kernel 58 -> 69 us alone with the per-row adds)
                const int my_tile = valid ? e.row / kRowTile : -1;
                const int prev_tile = __shfl_up(my_tile, 1), next_tile = __shfl_down(my_tile, 1);
                const bool t_head = l == 0 || prev_tile != my_tile, t_tail = l == kWave - 1 || next_tile != my_tile;
                const int t_base = wave_incl_max(t_head ? incl - f_cnt : 0);
                if (valid && t_tail && incl - t_base > 0) atomicAdd(&w.tile_cnt[my_tile], incl - t_base);
            }
        }
        // Path L: the batch's pieces of 17..32 symbols, lane per piece again -- 32 of them at a time: 32 lanes x 32
        // symbols are the bytes of path F's 64 x 16 arrays.  (On path W each of them took the whole wave for about one
        // global round trip per merge: with mixed scripts a fifth of the deferred pieces, and 70 % of the kernel.)
        unsigned long long lm = __ballot(is_l);
        while (lm) {
            uint8_t* long_src = long_src_all[wave_in_block()];
            const int rank = rank_below(lm);
            const int cnt = __popcll(lm) < kWave / 2 ? __popcll(lm) : kWave / 2;
            wave_sync();  // the previous user of the LDS arrays (path F / the previous group) is done
            if (((lm >> l) & 1ull) && rank < kWave / 2) long_src[rank] = uint8_t(l);
            wave_sync();
            const int src = l < cnt ? int(long_src[l]) : 0;
            const int s_begin = __shfl(e.begin, src), s_len = __shfl(e.len, src), s_pos = __shfl(e.stage_pos, src);
            const int s_row = __shfl(e.row, src);
            int res = 0;
            if (l < cnt) {
                const uint8_t* text = in.chars + s_begin;
                const int s_need = s_len + SL;
                const int n = bpe_symbolize(
                    T, root_lds, [&](int i) -> uint32_t { return i < s_len ? text[i] : T.suffix[i - s_len]; }, s_need,
                    [&](int k, int tok) { fid[k * (kWave / 2) + l] = IdT(tok); });
                res = bpe_merge_lane<IdT, kLongSyms, kWave / 2>(T, fid, fkey, fnid, n);
                if (res >= 0) {
                    for (int k = 0; k < res; ++k) stage_put(w, s_pos + k, int32_t(fid[k * (kWave / 2) + l]));
                    for (int k = res; k < s_need; ++k) stage_clear(w, s_pos + k);
                    if (res) {
                        atomicAdd(&w.row_cnt[s_row], res);
                        if (w.tile_cnt) atomicAdd(&w.tile_cnt[s_row / kRowTile], res);
                    }
                }
            }
            if (T.store.slots) {  // the long pieces' results: the store keys up to 31 bytes
                const bool want = l < cnt && res >= 0 && res <= kStoreIds && s_len <= kStoreKeyBytes;
                uint32_t lkey[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pay[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (want) {
                    store_key_long(in.chars + s_begin, s_len, lkey);
#pragma unroll
                    for (int k = 0; k < kStoreIds; ++k) {
                        const uint32_t v = k < res ? uint32_t(fid[k * (kWave / 2) + l]) : 0u;
                        if (NARROW) pay[k >> 1] |= v << (16 * (k & 1));
                        else pay[k] = v;
                    }
                }
                store_offer(want, lkey, pay, res);
            }
            // a non-unique minimum goes to the exact path: tell the piece's own lane
            const unsigned long long failed = __ballot(l < cnt && res < 0);
            if (((lm >> l) & 1ull) && rank < kWave / 2 && ((failed >> rank) & 1ull)) is_x = true;
            // drop the pieces of this group from the mask
            const unsigned long long done = __ballot(((lm >> l) & 1ull) && rank < kWave / 2);
            lm &= ~done;
        }
        PHASE(10);
        unsigned long long wm = __ballot(is_w);
        while (wm) {
            const int src = __ffsll(wm) - 1;
            wm &= wm - 1;
            const int s_begin = __shfl(e.begin, src), s_len = __shfl(e.len, src), s_pos = __shfl(e.stage_pos, src);
            const int s_need = s_len + SL;
            wave_sync();
            int n = 0;
            if (l == src) {
                const uint8_t* text = in.chars + s_begin;
                n = bpe_symbolize(
                    T, root_lds, [&](int i) -> uint32_t { return i < s_len ? text[i] : T.suffix[i - s_len]; }, s_need,
                    [&](int k, int tok) { id[k] = uint32_t(tok); });
            }
            n = __shfl(n, src);
            wave_sync();
            const int res = bpe_merge_wave(T, id, key, n);
            if (res < 0) {
                if (l == src) is_x = true;
            } else {
                for (int k = l; k < s_need; k += kWave) {
                    if (k < res) stage_put(w, s_pos + k, int32_t(id[k]));
                    else stage_clear(w, s_pos + k);
                }
                if (l == src && res) {
                    atomicAdd(&w.row_cnt[e.row], res);
                    if (w.tile_cnt) atomicAdd(&w.tile_cnt[e.row / kRowTile], res);
                }
            }
        }
        PHASE(11);
        const unsigned long long xm = __ballot(is_x);
        if (xm && w.tile_sums && tail_rows > 0 && !solo) {
            // no tail block to leave them to: the lane works its piece out here (rare: a tie in the merge heap, a piece of more than
            // kChunkSyms symbols -- the tail ran them a thread each after everybody else was through, which was no faster)
            if (is_x) exact_piece(in, T, w, ExactPiece{e.begin, e.len, e.stage_pos, e.row});
        } else if (xm) {
            int idx = 0;
            if (l == 0) idx = atomicAdd(&w.status->n_exact, __popcll(xm));
            idx = __shfl(idx, 0) + rank_below(xm);
            if (is_x) {
                if (idx < w.exact_cap) w.exact[idx] = ExactPiece{e.begin, e.len, e.stage_pos, e.row};
                else atomicOr(&w.status->flags, kFlagExactOverflow);
            }
            if (l == 0) pushed_exact = 1;
        }
    }
    }
    if (tail_rows <= 0) return;
    if (w.tile_sums && !solo) return;   // (compact_kernel sums tile_cnt itself: no ticket, no scan -- three dependent round trips less at the end of every call)
    // ---- folded tail: every block takes a ticket when its batches are done; the last one is alone on the data
    __syncthreads();
    PROBE(4);
    if (solo) {
        if (threadIdx.x == 0) publish_release();
        __syncthreads();
        publish_acquire();
    } else if (!last_block_done_sharded(w.status, int(blockIdx.x), gridDim.y, pushed_exact != 0)) {   // (grid: kShards x blocks per shard)
        return;
    }
    const int n_exact = w.status->n_exact;
    if (n_exact > kBlockThreads || (w.status->flags & kFlagExactOverflow)) {  // too many for one block: separate launches
        if (threadIdx.x == 0) atomicOr(&w.status->flags, kFlagTailPending);
        return;
    }
    if (n_exact > 0) {
        if (int(threadIdx.x) < n_exact) exact_one(in, T, w, int(threadIdx.x));
        publish_release();  // the exact pieces' count atomics before the scan below
        __syncthreads();
        publish_acquire();
        if (w.status->flags & kFlagScratchOverflow) return;
    }
    scan_tiles_one_block(tail_rows, w, out_cap);
}
template <bool NARROW>
static __global__ __launch_bounds__(kBlockThreads) void merge_kernel(RowsIn in, BpeDev T, EncodeWork w, int tail_rows,
                                                                     long long out_cap) {
    merge_body<NARROW>(in, T, w, tail_rows, out_cap);
}

// ---- path X as its own launch.
static __global__ __launch_bounds__(kBlockThreads) void exact_kernel(RowsIn in, BpeDev T, EncodeWork w) {
    if (w.status->flags & (kFatalFlags | kFlagDeferOverflow | kFlagExactOverflow)) return;
    const int n = w.status->n_exact;
    const int stride = int(gridDim.x) * kBlockThreads;
    for (int i = int(blockIdx.x) * kBlockThreads + int(threadIdx.x); i < n; i += stride) exact_one(in, T, w, i);
}

// ---- the tile scan of the final offsets: one wave per tile of kRowTile rows sums the row counts, the last block
// scans the tile sums.
static __global__ __launch_bounds__(kBlockThreads) void count_scan_kernel(int n_rows, EncodeWork w, long long out_cap) {
    const int l = lane_id();
    const int n_tiles = (n_rows + kRowTile - 1) / kRowTile;
    const int my_waves = int(gridDim.x) * kWavesPerBlock;
    for (int tile = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); tile < n_tiles; tile += my_waves) {
        const int row = tile * kRowTile + l;
        const int s = wave_sum(row < n_rows ? w.row_cnt[row] : 0);
        if (l == 0) w.tile_off[tile] = s;
    }
    if (!last_block_done(&w.status->ticket[1], gridDim.x)) return;
    const long long total = block_exclusive_scan<kBlockThreads, 16>(
        n_tiles, [&](int i) -> long long { return w.tile_off[i]; }, [&](int i, long long off) { w.tile_off[i] = off; });
    if (threadIdx.x == 0) {
        w.status->n_out = total > INT32_MAX ? INT32_MAX : int32_t(total);
        if (total > out_cap) atomicOr(&w.status->flags, kFlagOutCapacity);
    }
}

// ---- staging -> caller's buffer + the rows' begins/ends.  A wave takes kCompactRows consecutive rows of one tile: the
// tile's 64 row records (count, staging offset, entries used) arrive with three coalesced loads and ONE prefix sum
// serves all of them, and the staged ids of all its rows are requested before the first is stored -- a wave pays a few
// memory round trips per item instead of three dependent ones per row (the kernel was latency-bound: 30 -> 23 us at config 2).
// Unused staging entries (rows that had deferred pieces) are squeezed out by ballot compaction.
constexpr int kCompactRows = 4;    // rows per work item (8: no better)
constexpr int kCompactChunks = 3;  // 64-entry chunks of a row held in registers; longer rows finish in a loop
// Where compact_body puts the result.  RaggedSink: the caller's begins / ends / ids (the ops' contract).  WireSink: the
// row-shard exchange's wire (ops_kernels.hpp "row-shard exchange": header | i32 ends[max_rows] | pad_ids ids of 2 or 4
// bytes) -- the ids leave the encode already narrowed and in place for the all-gather, no i32 copy, no pack kernel; ids
// beyond the wire's pad are dropped (the header's n_ids says so: the receivers ask for a larger pad).
struct RaggedSink {
    int32_t *ids, *begins, *ends;
    __device__ __forceinline__ void start(const RunStatus*) {}
    __device__ __forceinline__ void row(int r, int b, int e) const {
        begins[r] = b;
        ends[r] = e;
    }
    // id number j of row r (which has cnt of them), position pos of the ragged output
    __device__ __forceinline__ void id(int, int, int, int pos, int32_t v) const { ids[pos] = v; }
    __device__ __forceinline__ void row_done(int, int) const {}   // (every lane of the wave)
    __device__ __forceinline__ void finish(int, const RunStatus*) const {}
};
// DenseSink: the graph's tail in the encode's last pass (ovtk_encode_dense_*) -- Truncate (src/truncate.cpp:37-150, one input) ->
// CombineSegments with constant segments in front and behind (src/combine_segments.cpp:36-134: the post-processor's BOS / EOS)
// -> RaggedToDense twice (src/ragged_to_dense.cpp:70-174): input_ids [n_rows, T] and attention_mask [n_rows, T].  No ragged ids
// tensor exists.  T = RunStatus::width (row_width_kernel, launched in front).
constexpr int kDenseAffix = 4;   // constant ids in front / behind, at most
struct DenseSink {
    int32_t* ids;        // [n_rows * T]
    uint8_t* mask;       // [n_rows * T] or nullptr
    int64_t capacity;    // cells
    int32_t max_length;  // Truncate: ids of a row that stay
    int32_t trunc_left;  // 0: the first max_length stay, 1: the last
    int32_t pad_right, pad_value;
    int32_t target_dim;  // < 0: the longest row
    int32_t n_pre, n_suf;
    int32_t pre[kDenseAffix], suf[kDenseAffix];
    int32_t T;           // (set by start)
    __device__ __forceinline__ void start(const RunStatus* st) { T = st->width; }
    // (constant subscripts only: indexed by a lane's value the arrays -- and with them the whole struct -- went to scratch memory, and
    // every field was fetched from there again wherever it was used: 164 scratch loads in compact_kernel<DenseSink>)
    __device__ __forceinline__ int32_t pre_at(int k) const { return k == 0 ? pre[0] : (k == 1 ? pre[1] : (k == 2 ? pre[2] : pre[3])); }
    __device__ __forceinline__ int32_t suf_at(int k) const { return k == 0 ? suf[0] : (k == 1 ? suf[1] : (k == 2 ? suf[2] : suf[3])); }
    __device__ __forceinline__ void row(int, int, int) const {}
    __device__ __forceinline__ void id(int r, int cnt, int j, int, int32_t v) const {
        const int keep = cnt < max_length ? cnt : max_length, first = trunc_left ? cnt - keep : 0;
        if (j < first || j >= first + keep) return;
        const int len = keep + n_pre + n_suf;
        const int col = (pad_right ? 0 : (T > len ? T - len : 0)) + n_pre + (j - first);
        if (col < T) ids[(long long)r * T + col] = v;   // (a target_dim below a row's length cuts the row, as RaggedToDense does: :120-135)
    }
    __device__ __forceinline__ void row_done(int r, int cnt) const {   // every lane: the cells that are not the row's own ids
        const int keep = cnt < max_length ? cnt : max_length;
        const int len = keep + n_pre + n_suf;
        const int col0 = pad_right ? 0 : (T > len ? T - len : 0);
        for (int c = lane_id(); c < T; c += kWave) {
            const int k = c - col0;
            const bool inside = k >= 0 && k < len;
            if (mask) mask[(long long)r * T + c] = inside ? uint8_t(1) : uint8_t(0);
            if (!inside) ids[(long long)r * T + c] = pad_value;
            else if (k < n_pre) ids[(long long)r * T + c] = pre_at(k);
            else if (k >= n_pre + keep) ids[(long long)r * T + c] = suf_at(k - n_pre - keep);
        }
    }
    __device__ __forceinline__ void finish(int, const RunStatus*) const {}
};
// T of a dense output: target_dim, or the longest row after Truncate and the constant segments; more cells than the caller has room
// for: kFlagOutCapacity (compact_kernel then writes nothing).
static __global__ __launch_bounds__(kBlockThreads) void row_width_kernel(int n_rows, EncodeWork w, DenseSink d) {
    int m = 0;
    if (d.target_dim < 0) {
        for (int r = int(blockIdx.x) * kBlockThreads + int(threadIdx.x); r < n_rows; r += int(gridDim.x) * kBlockThreads) {
            const int c = w.row_cnt[r];
            const int len = (c < d.max_length ? c : d.max_length) + d.n_pre + d.n_suf;
            m = len > m ? len : m;
        }
        m = wave_max(m);
        if (lane_id() == 0 && m > 0) atomicMax(&w.status->width, m);
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.status->width = d.target_dim;
    }
    // the last block to arrive checks the room
    if (!last_block_done(&w.status->width_ticket, gridDim.x, false)) return;
    if (threadIdx.x == 0) {
        const int T = __hip_atomic_load(&w.status->width, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((long long)T * n_rows > d.capacity) atomicOr(&w.status->flags, kFlagOutCapacity);
    }
}
struct WireSink {
    int32_t* hdr;      // i32 n_ids, i32 n_rows, 0, 0
    int32_t* ends;     // [max_rows]
    void* ids;         // [pad_ids] u16 or i32
    int32_t pad_ids, max_rows, id_bytes;
    __device__ __forceinline__ void start(const RunStatus*) {}
    __device__ __forceinline__ void row(int r, int, int e) const { ends[r] = e; }
    __device__ __forceinline__ void row_done(int, int) const {}
    __device__ __forceinline__ void id(int, int, int, int pos, int32_t v) const {
        if (pos >= pad_ids) return;
        if (id_bytes == 2) static_cast<uint16_t*>(ids)[pos] = uint16_t(v);
        else static_cast<int32_t*>(ids)[pos] = v;
    }
    __device__ __forceinline__ void finish(int n_rows, const RunStatus* st) const {  // the first block: header, unused row slots
        if (blockIdx.x != 0) return;
        if (threadIdx.x == 0) {
            hdr[0] = st->n_out;
            hdr[1] = n_rows;
            hdr[2] = hdr[3] = 0;
        }
        for (int r = n_rows + int(threadIdx.x); r < max_rows; r += kBlockThreads) ends[r] = 0;
    }
};

// S16: the staging entries are u16 (w.stage16) -- a template flag, so that the loads of a work item still leave together.
template <bool S16>
__device__ __forceinline__ int32_t stage_get_t(const EncodeWork& w, int pos) {
    if (S16) {
        const uint32_t x = reinterpret_cast<const uint16_t*>(w.stage)[pos];
        return x == 0xFFFFu ? kEmptyId : int32_t(x);
    }
    return w.stage[pos];
}
// The flat form of a work item (round 5; RaggedSink): the item's four rows nearly always lie back to back in the staging buffer (one
// wave of the lookup kernel staged them), and their ids back to back in the output.  So the item is ONE stretch of staging entries,
// worked through 512 (u16) or 256 (i32) at a time: every lane takes eight (four) of them with one 16-byte load, the valid ones are
// squeezed through an LDS buffer (a prefix sum over the lanes' counts), and leave 16 bytes per lane and store.  A row at a time -- a lane per entry, three
// 2-byte loads, a ballot and a 4-byte store per lane for a row of 112 ids -- cost 60 instructions per row; this costs about as much
// per FOUR rows (compact_kernel 7.0 M -> see DESIGN 6 quad-cycles of a config-2 step's 45 M).
constexpr int kFlatIds = kWave * 8;    // ids the LDS buffer holds: one 16-byte load per lane of 2-byte entries
struct __attribute__((packed, aligned(1))) FlatBytes16 { uint32_t d[4]; };
struct __attribute__((packed, aligned(1))) FlatBytes8 { uint32_t d[2]; };
// part / split: EncodeWork::compact_split waves share an item (few, long rows); the copy's steps are dealt out among them, anything else
// is part 0's.
// A stretch of `total_used` staging entries from `stage_bytes` on -> i32 ids at `out`: a widening copy when no entry of it is unused
// (`plain`), else the valid entries squeezed through `buf` (kFlatIds ids of LDS, the wave's own).  compact_flat's copy, and the tail of
// lookup_span_kernel's one-pass form (the wave's whole stretch).
template <bool S16>
__device__ __forceinline__ void flat_copy(const uint8_t* stage_bytes, int total_used, bool plain, int32_t* out, int32_t* buf, int part, int split) {
    constexpr int E = S16 ? 8 : 4;   // entries per 16 bytes
    const int l = lane_id();
    // the stretch in steps of 64 x 16 bytes (rows of any length: a row of 8 KB is a dozen steps); the next step's load leaves before
    // this step's entries are looked at
    auto fetch = [&](int off) -> FlatBytes16 {
        const int idx = off + l * E;
        if (idx < total_used) return *reinterpret_cast<const FlatBytes16*>(stage_bytes + idx * (S16 ? 2 : 4));   // (may read up to 14 bytes behind
                                                                                                                // the stretch: the buffer has the slack)
        return FlatBytes16{{0u, 0u, 0u, 0u}};
    };
    if (plain) {
        // No unused entry in the stretch (no piece of these rows was deferred, or each came to exactly the entries it had reserved -- with
        // the six-id memo entries of round 5 that is nearly every item): a widening copy.  Every group of four entries stands on its own --
        // no prefix sum, no LDS, a step's loads all in flight before its first store -- where the squeeze below is a chain of one memory
        // round trip per step (rows of 8 KB: 14 steps per item).  A lane takes FOUR entries per load (8 bytes of u16 / 16 bytes of i32) and
        // stores them as 16 bytes: every load and every store of the wave is one contiguous stretch (eight entries per lane and two
        // stores 32 bytes apart were measured first: compact_kernel 19.9 -> 22.6 us).
        constexpr int U = 2;   // groups of 4 x 64 entries in flight (512 ids a step: config 2's items are ~450; four cost every item the
                               // address arithmetic of two groups it does not have)
        for (int off0 = part * (U * kWave * 4); off0 < total_used; off0 += split * (U * kWave * 4)) {
            uint32_t x[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = off0 + (u * kWave + l) * 4;
                x[u][0] = x[u][1] = x[u][2] = x[u][3] = 0u;
                if (idx < total_used) {   // (may read up to 12 bytes behind the stretch: the buffer has the slack)
                    if (S16) {
                        const FlatBytes8 r = *reinterpret_cast<const FlatBytes8*>(stage_bytes + idx * 2);
                        x[u][0] = r.d[0] & 0xFFFFu; x[u][1] = r.d[0] >> 16; x[u][2] = r.d[1] & 0xFFFFu; x[u][3] = r.d[1] >> 16;
                    } else {
                        const FlatBytes16 r = *reinterpret_cast<const FlatBytes16*>(stage_bytes + idx * 4);
                        x[u][0] = r.d[0]; x[u][1] = r.d[1]; x[u][2] = r.d[2]; x[u][3] = r.d[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = off0 + (u * kWave + l) * 4;
                const int n_here = total_used - idx;
                if (n_here >= 4) {
                    *reinterpret_cast<FlatBytes16*>(out + idx) = FlatBytes16{{x[u][0], x[u][1], x[u][2], x[u][3]}};
                } else {
                    if (n_here > 0) out[idx] = int32_t(x[u][0]);
                    if (n_here > 1) out[idx + 1] = int32_t(x[u][1]);
                    if (n_here > 2) out[idx + 2] = int32_t(x[u][2]);
                }
            }
        }
        return;
    }
    FlatBytes16 v = fetch(0);
    for (int off = 0; off < total_used; off += kWave * E) {
        const FlatBytes16 nxt = fetch(off + kWave * E);
        const int n_here = total_used - (off + l * E);   // entries of the lane that exist (<= 0: none)
        int32_t e[E];
        uint32_t valid = 0;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            if (S16) {
                const uint32_t x = (v.d[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                e[i] = int32_t(x);
                if (i < n_here && x != 0xFFFFu) valid |= 1u << i;
            } else {
                e[i] = int32_t(v.d[i]);
                if (i < n_here && e[i] != kEmptyId) valid |= 1u << i;
            }
        }
        const int c = __popc(valid);
        const int incl = wave_incl_sum(c);
        int at = incl - c;
#pragma unroll
        for (int i = 0; i < E; ++i)
            if ((valid >> i) & 1u) buf[at++] = e[i];
        const int total = wave_readlane(incl, kWave - 1);
        wave_sync();
        for (int t = 4 * l; t < total; t += 4 * kWave) {
            if (t + 4 <= total) {
                *reinterpret_cast<FlatBytes16*>(out + t) = *reinterpret_cast<const FlatBytes16*>(buf + t);
            } else {
                for (int i = t; i < total; ++i) out[i] = buf[i];
            }
        }
        out += total;
        v = nxt;
        wave_sync();   // (the buffer is the next step's)
    }
}
template <bool S16>
__device__ __forceinline__ bool compact_flat(const EncodeWork& w, const RaggedSink& sink, int32_t* buf, const int (&cnt)[4], const int (&o)[4],
                                             const int (&base)[4], const int (&used)[4], int row0, int part, int split) {
    const int l = lane_id();
    const int total_used = used[0] + used[1] + used[2] + used[3];
    if (used[0] < 0 || used[1] < 0 || used[2] < 0 || used[3] < 0 || base[1] != base[0] + used[0] || base[2] != base[1] + used[1] ||
        base[3] != base[2] + used[2])
        return false;
    const bool plain = total_used == cnt[0] + cnt[1] + cnt[2] + cnt[3];
    if (part != 0 && !plain) return true;   // (the squeeze is one wave's)
    if (l < 4 && part == 0) {
        const int oo = l == 0 ? o[0] : (l == 1 ? o[1] : (l == 2 ? o[2] : o[3])), cc = l == 0 ? cnt[0] : (l == 1 ? cnt[1] : (l == 2 ? cnt[2] : cnt[3]));
        sink.row(row0 + l, oo, oo + cc);
    }
    flat_copy<S16>(reinterpret_cast<const uint8_t*>(w.stage) + (long long)base[0] * (S16 ? 2 : 4), total_used, plain, sink.ids + o[0], buf, part, split);
    return true;
}

template <class Sink, bool S16 = false>
__device__ __forceinline__ void compact_body(int n_rows, const EncodeWork& w, Sink sink, bool solo = false, int32_t* flat_buf = nullptr,
                                             uint32_t more_flags = 0u) {
    // kFlagTailPending: merge_kernel's folded tail left the exact pieces and the tile scan to a second attempt -- tile_off
    // and parts of the staging buffer hold whatever the previous call left there
    if ((w.status->flags | more_flags) & (kFatalFlags | kFlagOutCapacity | kFlagDeferOverflow | kFlagExactOverflow | kFlagScratchOverflow |
                                         kFlagTailPending))
        return;
    // the short path: rows or pieces were left to a kernel that was not launched (the host launches it, and this kernel again).
    // Unresolved pieces = entries of the deferred list: with span_sums nothing else is filed there (a counter of its own, one more add
    // per flush on ONE address, cost text that misses everywhere half its rate: 135 000 flushes per batch of uniform-random text).
    if ((w.skip_mask & kSkipPending) && w.status->n_pending != 0) return;
    if (w.skip_mask & kSkipMerge) {
        const int mine = lane_id() < kShards ? __hip_atomic_load(&w.status->shard_count[lane_id() * kCounterStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        if (__ballot(mine != 0)) return;
    }
    sink.start(w.status);
    const int l = lane_id();
    const int n_waves = (solo ? 1 : int(gridDim.x)) * kWavesPerBlock;
    constexpr int kSubs = kRowTile / kCompactRows;
    const int n_items = ((n_rows + kRowTile - 1) / kRowTile) * kSubs;
    const int split = std::is_same<Sink, RaggedSink>::value && !solo && w.compact_split > 1 ? w.compact_split : 1;   // (a power of two)
    for (int unit = (solo ? 0 : int(blockIdx.x)) * kWavesPerBlock + wave_in_block(); unit < n_items * split; unit += n_waves) {
        const int item = unit / split, part = unit % split;
        const int tile = item / kSubs, sub = item % kSubs;
        const int rj = tile * kRowTile + l;
        const bool have = rj < n_rows;
        const int c = have ? w.row_cnt[rj] : 0;
        const int sv = have ? w.row_stage[rj] : 0;
        const int uv = have ? w.row_used[rj] : 0;
        long long toff;
        if constexpr (std::is_same<Sink, DenseSink>::value) {
            toff = 0;   // (a dense cell's place follows from its row and column alone)
        } else if (w.tile_sums && !solo) {
            // the tile's offset = the counts of the tiles in front of it, summed here (tile_cnt: 4 bytes per 64 rows, hot in every L2;
            // the array has the slack for 16-byte reads).  What merge_kernel's last block did for everybody, at the price of a ticket
            // and a scan on every call's chain.  A lane takes sixteen tiles per step, its loads leave together with the row records'
            // above -- before anything waits for those.
            // (whole groups of sixteen in front of the tile unmasked -- a lane's group counts or it does not --, the up to fifteen tiles
            // between the last whole group and the tile one to a lane: 22 vector instructions a step where masking every value took 64)
            int acc = 0;
            const int whole = tile >> 4, rest = tile & 15;   // (wave-uniform)
            for (int g0 = 0; g0 < whole; g0 += kWave) {
                const int g = g0 + l;
                int4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = g < whole ? *reinterpret_cast<const int4*>(w.tile_cnt + 16 * g + 4 * u) : int4{0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            }
            acc += l < rest ? w.tile_cnt[16 * whole + l] : 0;
            toff = wave_sum(acc);
        } else {
            toff = w.tile_off[tile];
        }
        const int incl = wave_incl_sum(c);
        int cnt[kCompactRows], o[kCompactRows], base[kCompactRows], used[kCompactRows];
        int32_t v[kCompactRows][kCompactChunks];
#pragma unroll
        for (int q = 0; q < kCompactRows; ++q) {
            const int j = sub * kCompactRows + q;
            cnt[q] = wave_readlane(c, j);
            o[q] = int(toff + (wave_readlane(incl, j) - cnt[q]));
            base[q] = wave_readlane(sv, j);
            used[q] = wave_readlane(uv, j);
        }
        if constexpr (std::is_same<Sink, RaggedSink>::value) {
            // (tile_sums: the total is block 0's to find out while everybody writes -- an item that would leave the caller's buffer
            // writes nothing, the call ends with OVTK_E_CAPACITY either way)
            if (w.tile_sums && !solo && (long long)o[kCompactRows - 1] + cnt[kCompactRows - 1] > w.out_cap) continue;
        }
        if constexpr (std::is_same<Sink, RaggedSink>::value && kCompactRows == 4) {
            const int row0 = tile * kRowTile + sub * kCompactRows;
            if (flat_buf && row0 + kCompactRows <= n_rows && compact_flat<S16>(w, sink, flat_buf, cnt, o, base, used, row0, part, split)) continue;
        }
        if (part != 0) continue;   // (row by row: one wave's)
#pragma unroll
        for (int q = 0; q < kCompactRows; ++q)
#pragma unroll
            for (int k = 0; k < kCompactChunks; ++k)
                v[q][k] = (k * kWave + l < used[q]) ? stage_get_t<S16>(w, base[q] + k * kWave + l) : kEmptyId;
#pragma unroll
        for (int q = 0; q < kCompactRows; ++q) {
            const int row = tile * kRowTile + sub * kCompactRows + q;
            if (row >= n_rows) break;
            if (l == 0) sink.row(row, o[q], o[q] + cnt[q]);
            int run = 0;
#pragma unroll
            for (int k = 0; k < kCompactChunks; ++k) {
                if (k * kWave < used[q]) {
                    const unsigned long long m = __ballot(v[q][k] != kEmptyId);
                    if (v[q][k] != kEmptyId) {
                        const int j = run + rank_below(m);
                        sink.id(row, cnt[q], j, o[q] + j, v[q][k]);
                    }
                    run += __popcll(m);
                }
            }
            for (int b = kCompactChunks * kWave; b < used[q]; b += kWave) {
                const int x = (b + l < used[q]) ? stage_get_t<S16>(w, base[q] + b + l) : kEmptyId;
                const unsigned long long m = __ballot(x != kEmptyId);
                if (x != kEmptyId) {
                    const int j = run + rank_below(m);
                    sink.id(row, cnt[q], j, o[q] + j, x);
                }
                run += __popcll(m);
            }
            sink.row_done(row, cnt[q]);
        }
    }
    sink.finish(n_rows, w.status);
}

template <class Sink, bool S16 = false>
static __global__ __launch_bounds__(kBlockThreads) void compact_kernel(int n_rows, EncodeWork w, Sink sink) {
    // The call's last kernel reports: every kernel before it is through (stream order), so the status block holds its final
    // values -- block 0 writes what the host reads (the scalar fields, the shards' deferred counts) straight into the caller's
    // pinned block and zeroes the workspace's other status block for its next call.  Two dispatches less per call (status
    // memset, status copy: +-2.7 % of config 2's step, measured by adding two).  Whatever the flags say: before the early return.
    if (w.next_status && blockIdx.x == 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(w.status);
        uint32_t* dst = reinterpret_cast<uint32_t*>(w.host_status);
        constexpr int kHead = int(offsetof(RunStatus, shard_count) / 4);
        constexpr int kShard0 = kHead;
        for (int i = int(threadIdx.x); i < kHead + kShards; i += kBlockThreads) {
            const int at = i < kHead ? i : kShard0 + (i - kHead) * kCounterStride;
            if (w.tile_sums && (at == int(offsetof(RunStatus, n_out) / 4) || at == int(offsetof(RunStatus, flags) / 4))) continue;   // (thread 0, below)
            __hip_atomic_store(dst + at, __hip_atomic_load(src + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        uint32_t* nxt = reinterpret_cast<uint32_t*>(w.next_status);
        for (int i = int(threadIdx.x); i < w.status_words; i += kBlockThreads) nxt[i] = 0u;
    }
    __shared__ int32_t flat_all[std::is_same<Sink, RaggedSink>::value ? kWavesPerBlock * kFlatIds : 1];
    __shared__ uint32_t over_s;   // block 0: kFlagOutCapacity as this launch found it (the block's threads must agree on it)
    if (threadIdx.x == 0) over_s = 0u;
    if (w.tile_sums && blockIdx.x == 0) {
        // the call's total and its capacity check (merge_kernel's folded tail did both): the sum of every tile's count
        __shared__ long long total_s[kWavesPerBlock];
        const int n_tiles = (n_rows + kRowTile - 1) / kRowTile;
        long long acc = 0;
        for (int i = int(threadIdx.x); i < n_tiles; i += kBlockThreads) acc += w.tile_cnt[i];
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane_id() == 0) total_s[wave_in_block()] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long total = 0;
            for (int k = 0; k < kWavesPerBlock; ++k) total += total_s[k];
            const int32_t n_out = total > INT32_MAX ? INT32_MAX : int32_t(total);
            const uint32_t over = total > w.out_cap ? kFlagOutCapacity : 0u;
            w.status->n_out = n_out;   // (WireSink::finish reads it: the same thread)
            // the two words of the caller's block that the copy above left out
            if (w.next_status) {
                __hip_atomic_store(&w.host_status->n_out, n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&w.host_status->flags, __hip_atomic_load(&w.status->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | over,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (over) atomicOr(&w.status->flags, over);
            over_s = over;
        }
    }
    __syncthreads();
    compact_body<Sink, S16>(n_rows, w, sink, false, std::is_same<Sink, RaggedSink>::value ? flat_all + wave_in_block() * kFlatIds : nullptr, over_s);
}

// ---- a small batch in ONE launch (BASELINE config 1: 32 x 128-byte strings; any batch of a few hundred short rows).
// A call of the pipeline above is six dispatches (status memset, two lookup kernels, merge, compact, status copy), each
// waiting for the one before: tens of microseconds of launch latency for microseconds of work.  Here every block runs
// the lookup body on its rows (a wave per row, as ever); the block that finishes LAST ("last block done": agent-scope
// release, ticket, acquire) goes on alone with the merge body (every shard, the exact pieces, the tile scan), the compact
// body, writes the status block straight into the caller's pinned host block and leaves the device-side status zeroed for
// the next call -- no memset, no copy.  Same bodies, same results.
template <int MODE, bool NARROW>
static __global__ __launch_bounds__(kBlockThreads) void encode_small_kernel(RowsIn in, SplitDev sp, BpeDev T, EncodeWork w) {
    lookup_body<MODE, false>(in, sp, T, w);
    if (!last_block_done(&w.status->ticket[1], gridDim.x)) return;
    merge_body<NARROW>(in, T, w, in.n_rows, w.out_cap, /*solo=*/true);
    __syncthreads();
    if (threadIdx.x == 0) publish_release();
    __syncthreads();
    publish_acquire();
    if (w.stage16) compact_body<RaggedSink, true>(in.n_rows, w, RaggedSink{w.out_ids, w.out_begins, w.out_ends}, /*solo=*/true);
    else compact_body<RaggedSink, false>(in.n_rows, w, RaggedSink{w.out_ids, w.out_begins, w.out_ends}, /*solo=*/true);
    __syncthreads();
    uint32_t* src = reinterpret_cast<uint32_t*>(w.status);
    uint32_t* dst = reinterpret_cast<uint32_t*>(w.host_status);
    constexpr int kHead = int(offsetof(RunStatus, shard_count) / 4);  // the scalar fields; of the counter arrays the host reads
    for (int i = int(threadIdx.x); i < w.status_words; i += kBlockThreads) {  // shard_count[0] only (the one shard in use)
        const uint32_t v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i <= kHead) __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // each one a write over PCIe
        src[i] = 0;  // clean for the next call of this workspace
    }
}

// ---- RegexSplit as its own op: count pass, then write pass (the scan is cheap enough to run twice).
// mode 0: row_cnt[row] = number of pieces.  mode 1: write begins/ends/skips at row_out[row].
// mode 2 (round 6, the op in one pass): write at w.row_stage[row] -- the row's region in buffers of the reference's capacity, a slot per byte and
// one per string, the offsets a scan of those bounds -- and file the count in row_cnt[row]; the caller scans the counts and gathers.
template <int WRITE, bool LLAMA3 = false>
static __global__ __launch_bounds__(kBlockThreads) void split_kernel(RowsIn in, SplitDev sp, int max_splits, EncodeWork w,
                                                                     int32_t* out_rb, int32_t* out_re,
                                                                     int32_t* out_begins, int32_t* out_ends,
                                                                     uint8_t* out_skips) {
    __shared__ WaveScratch ws_all[kWavesPerBlock];
    if (w.status->flags & (kFlagRange | kFlagOutCapacity)) return;
    WaveScratch& ws = ws_all[wave_in_block()];
    const int l = lane_id();
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int row = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); row < in.n_rows; row += n_waves) {
        int count = 0;
        int total = 0;
        const int o = WRITE == 2 ? w.row_stage[row] : (WRITE ? int(row_output_offset(w, in.n_rows, row, total)) : 0);
        if (WRITE == 1 && l == 0) {
            out_rb[row] = o;
            out_re[row] = o + total;
        }
        for (int col = in.ragged_begins[row]; col < in.ragged_ends[row]; ++col) {
            const int sb = in.begins[col], se = in.ends[col];
            if (in.skips && in.skips[col]) {
                if (WRITE && l == 0) {
                    out_begins[o + count] = sb;
                    out_ends[o + count] = se;
                    if (out_skips) out_skips[o + count] = 1;
                }
                ++count;
                continue;
            }
            int in_string = 0;  // pieces of this string so far (num_splits of regex_split.cpp:241)
            auto emit = [&](int idx, int b, int e) {  // lane-local: idx-th kept piece of the string
                if (!WRITE) return;
                out_begins[o + count + idx] = sb + b;
                // regex_split.cpp:278-280: the piece whose index equals max_splits is stretched to the end
                out_ends[o + count + idx] = (idx == max_splits) ? se : sb + e;
                if (out_skips) out_skips[o + count + idx] = 0;
            };
            scan_string<(LLAMA3 ? kScanLlama3 : kScanGeneric)>(
                ws, sp, in.chars + sb, se - sb, in.chars, in.chars + in.n_chars,
                [&](int np, int c0, int, int) {
                    for (int jb = 0; jb < np; jb += kWave) {
                        const int j = jb + l;
                        const bool keep = j < np && !(ws.pstart[j] & kPieceDropped);
                        const unsigned long long km = __ballot(keep);
                        if (keep)
                            emit(in_string + rank_below(km), c0 + int(ws.pstart[j] & kPiecePosMask),
                                 c0 + int(ws.pstart[j + 1] & kPiecePosMask));
                        in_string += __popcll(km);
                    }
                },
                [&](int b, int e, bool dropped) {
                    if (dropped) return;
                    if (l == 0) emit(in_string, b, e);
                    in_string += 1;
                });
            count += in_string;
        }
        if (WRITE != 1 && l == 0) w.row_cnt[row] = count;
    }
}

// ---- SpecialTokensSplit: one LANE per row, count pass then write pass.
template <int WRITE>
static __global__ __launch_bounds__(kBlockThreads) void special_split_kernel(RowsIn in, SpecialDev T, EncodeWork w,
                                                                             int32_t* out_rb, int32_t* out_re,
                                                                             int32_t* out_begins, int32_t* out_ends,
                                                                             uint8_t* out_skips) {
    if (w.status->flags & (kFlagRange | kFlagOutCapacity)) return;
    const int row = int(blockIdx.x) * kBlockThreads + int(threadIdx.x);
    const bool valid = row < in.n_rows;
    int o = 0;
    if (WRITE) {
        const int cnt = valid ? w.row_cnt[row] : 0;
        const int incl = wave_incl_sum(cnt);
        o = int(w.tile_off[row / kRowTile < (in.n_rows + kRowTile - 1) / kRowTile ? row / kRowTile : 0]) + incl - cnt;
        if (valid) {
            out_rb[row] = o;
            out_re[row] = o + cnt;
        }
    }
    if (!valid) return;
    int count = 0;
    auto put = [&](int b, int e, int skip) {
        if (WRITE) {
            out_begins[o + count] = b;
            out_ends[o + count] = e;
            out_skips[o + count] = uint8_t(skip);
        }
        ++count;
    };
    for (int col = in.ragged_begins[row]; col < in.ragged_ends[row]; ++col) {
        const int sb = in.begins[col], se = in.ends[col];
        if (in.skips && in.skips[col]) {  // special_tokens_split.cpp:110-113
            put(sb, se, 1);
            continue;
        }
        const uint8_t* s = in.chars + sb;
        const int slen = se - sb;
        int start = 0, mb = 0, gb = 0, ge = 0, me = 0;
        if (!string_may_hold_token(T, s, slen)) {   // no token's first byte in it: the string passes as it is
            if (slen > 0) put(sb, se, 0);
            continue;
        }
        while (start < slen && special_next_match(T, special_tabs(T), s, slen, start, mb, gb, ge, me)) {  // :127-141
            if (start < mb) put(sb + start, sb + mb, 0);
            put(sb + gb, sb + ge, 1);
            start = me;
        }
        if (start < slen) put(sb + start, sb + slen, 0);  // :143-147
    }
    if (!WRITE) w.row_cnt[row] = count;
}

// ---- SpecialTokensSplit inside the fused encode (ovtk_encode_special_* / ovtk_encode_dense_*): ONE pass, a wave per 64 rows.
// The op's contract is dense outputs behind a count pass (two launches of the kernel above, an offset scan between them, one lane
// walking each row).  The encode kernels only dereference offsets, so here -- as in regex_sparse_kernel -- every wave's strings go
// to a region of their own inside buffers of the reference's capacity (one atomic per wave takes it) and the rows' ragged begins /
// ends point there.  And nearly every row holds no special token at all: when the wave's 64 rows are one stretch of text (one
// string per row, each continuing the one before: what StringTensorUnpack produces), the wave first sweeps that stretch with
// coalesced 16-byte loads for the tokens' first bytes; only the rows in which one turns up are walked by their lane.
static __global__ __launch_bounds__(kBlockThreads) void special_sparse_kernel(RowsIn in, SpecialDev T_in, RunStatus* status, long long cap,
                                                                              int32_t* out_rb, int32_t* out_re, int32_t* out_begins,
                                                                              int32_t* out_ends, uint8_t* out_skips) {
    // The token lists in LDS when they are small (they nearly always are: a handful of special tokens): a lane that walks a row reads
    // them one dependent load after the other -- group flags, token range, offsets, first byte -- and from global memory that chain
    // was most of the kernel (52 us for config 2's batch with one row in a hundred holding a token).
    constexpr int kLdsTokens = 64, kLdsChars = 2048, kLdsGroups = 8;
    __shared__ int32_t tb_s[kLdsTokens], te_s[kLdsTokens], gf_s[kLdsGroups + 1];
    __shared__ uint8_t gfl_s[kLdsGroups];
    __shared__ __attribute__((aligned(16))) uint8_t tc_s[kLdsChars];
    if (T_in.n_groups <= kLdsGroups && T_in.n_tokens <= kLdsTokens && T_in.n_tok_chars <= kLdsChars) {
        for (int i = int(threadIdx.x); i < T_in.n_tokens; i += kBlockThreads) {
            tb_s[i] = T_in.tok_begins[i];
            te_s[i] = T_in.tok_ends[i];
        }
        for (int i = int(threadIdx.x); i <= T_in.n_groups; i += kBlockThreads) gf_s[i] = T_in.group_first[i];
        for (int i = int(threadIdx.x); i < T_in.n_groups; i += kBlockThreads) gfl_s[i] = T_in.group_flags[i];
        for (int i = int(threadIdx.x); i < T_in.n_tok_chars; i += kBlockThreads) tc_s[i] = T_in.tok_chars[i];
        __syncthreads();
    }
    const SpecialDev& T = T_in;
    SpecialTabs tabs = special_tabs(T_in);
    if (T_in.n_groups <= kLdsGroups && T_in.n_tokens <= kLdsTokens && T_in.n_tok_chars <= kLdsChars)
        tabs = SpecialTabs{tb_s, te_s, tc_s, gf_s, gfl_s};
    // A BLOCK per 64 rows: its four waves read the rows' headers alike (lane i = row i) and sweep a quarter of the stretch each --
    // one wave per 64 rows was one wave per SIMD with nothing to hide its memory round trips behind (49 us for config 2's batch
    // without a single token in it) --, then the first wave goes on alone with the rows' lanes.
    __shared__ uint32_t cand_s[2];
    __shared__ int32_t wlo_s[kWave], whi_s[kWave];
    if (threadIdx.x < 2) cand_s[threadIdx.x] = 0u;
    if (threadIdx.x < kWave) {
        wlo_s[threadIdx.x] = 0x7FFFFFFF;
        whi_s[threadIdx.x] = 0;
    }
    __syncthreads();
    const int l = lane_id();
    const int wq = wave_in_block();
    const int row = int(blockIdx.x) * kWave + l;
    const bool valid = row < in.n_rows;
    int cb = 0, ce = 0;
    bool ok = valid;
    if (valid) {
        cb = in.ragged_begins[row];
        ce = in.ragged_ends[row];
        if (cb < ce && (cb < 0 || ce > in.n_strings)) ok = false;
    }
    const bool one = ok && ce == cb + 1 && !(in.skips && in.skips[cb]);
    int sb = 0, se = 0;
    if (one) {
        sb = in.begins[cb];
        se = in.ends[cb];
        if (sb < 0 || se < sb || (long long)se > in.n_chars) ok = false;
    }
    if (valid && !ok && wq == 0) atomicOr(&status->flags, kFlagRange);
    // one stretch of text?  (every row of the wave one string, each beginning where the one before ends)
    const int prev_se = int(lane_prev(uint32_t(se)));
    const bool chain = __ballot(valid && !(one && ok && (l == 0 || sb == prev_se))) == 0ull;
    bool cand = true;   // the row may hold a token
    bool swept = false;   // ... and the sweep said where: [w_lo, w_hi) of its string
    int w_lo = 0x7FFFFFFF, w_hi = 0;
    if (chain && T.n_tok_first > 0) {
        const unsigned long long vm = __ballot(valid);
        const int last = 63 - __clzll(vm);
        const long long t0 = wave_readlane(sb, 0), t1 = wave_readlane(se, last);
        unsigned long long cand_rows = 0;
        constexpr int U = 4;   // 16-byte loads in flight per lane (one after the other they were 32 memory round trips per wave)
        for (long long base = t0 + (long long)wq * U * kWave * 16; base < t1; base += (long long)kWavesPerBlock * U * kWave * 16) {
            SeqBytes16 v[U];
            bool in_text[U], whole[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long at = base + (u * kWave + l) * 16;
                in_text[u] = at < t1;
                whole[u] = in_text[u] && at + 16 <= in.n_chars;
                v[u] = SeqBytes16{{0u, 0u, 0u, 0u}};
                if (whole[u]) v[u] = *reinterpret_cast<const SeqBytes16*>(in.chars + at);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t any = 0;
                for (int i = 0; i < T.n_tok_first; ++i) {
                    const uint32_t c = T.tok_first[i] * 0x01010101u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t t = v[u].d[j] ^ c;
                        any |= (t - 0x01010101u) & ~t;   // (bit 7 of some byte is set iff some byte of t is zero)
                    }
                }
                // (bytes behind t1 may hit too: a row walked for nothing; the tensor's last bytes: their row is walked)
                const bool hit = in_text[u] && (!whole[u] || (any & 0x80808080u) != 0u);
                unsigned long long hm = __ballot(hit);
                while (hm) {
                    const int src = __ffsll(hm) - 1;
                    hm &= hm - 1ull;
                    const long long lo = base + (u * kWave + src) * 16, hi = lo + 16;
                    const bool mine = valid && (long long)sb < hi && (long long)se > lo;
                    if (mine) {   // the window of my row in which the tokens' first bytes stand
                        const int a = int(lo - sb) > 0 ? int(lo - sb) : 0, b2 = int(hi - sb);
                        w_lo = a < w_lo ? a : w_lo;
                        w_hi = b2 > w_hi ? b2 : w_hi;
                    }
                    cand_rows |= __ballot(mine);
                }
            }
        }
        // the four waves' findings together
        if (l == 0 && cand_rows) {
            atomicOr(&cand_s[0], uint32_t(cand_rows));
            atomicOr(&cand_s[1], uint32_t(cand_rows >> 32));
        }
        if ((cand_rows >> l) & 1ull) {
            atomicMin(&wlo_s[l], w_lo);
            atomicMax(&whi_s[l], w_hi);
        }
        __syncthreads();
        if (wq != 0) return;
        cand = ((l < 32 ? cand_s[0] >> l : cand_s[1] >> (l - 32)) & 1u) != 0u;
        w_lo = wlo_s[l];
        w_hi = whi_s[l];
        swept = true;
    } else if (wq != 0) {
        return;
    }
    // the row's strings: `emit(begin, end, skip)` for each (special_tokens_split.cpp:104-147)
    auto walk = [&](auto&& emit) {
        if (!ok) return;
        if (one && !cand) {
            if (se > sb) emit(sb, se, 0);
            return;
        }
        for (int col = cb; col < ce; ++col) {
            const int b = in.begins[col], e = in.ends[col];
            if (b < 0 || e < b || (long long)e > in.n_chars) {
                atomicOr(&status->flags, kFlagRange);
                return;
            }
            if (in.skips && in.skips[col]) {   // :110-113
                emit(b, e, 1);
                continue;
            }
            const uint8_t* s = in.chars + b;
            const int slen = e - b;
            if (!swept && !string_may_hold_token(T, s, slen)) {
                if (slen > 0) emit(b, e, 0);
                continue;
            }
            int start = 0, mb = 0, gb = 0, ge = 0, me = 0;
            while (start < slen && special_next_match(T, tabs, s, slen, start, mb, gb, ge, me, swept ? w_lo : 0, swept ? w_hi : 0x7FFFFFFF)) {   // :127-141
                if (start < mb) emit(b + start, b + mb, 0);
                emit(b + gb, b + ge, 1);
                start = me;
            }
            if (start < slen) emit(b + start, b + slen, 0);   // :143-147
        }
    };
    // Where the strings go: a row of at most one string (nearly every row) has entry `row` of the buffers -- no counter, and the
    // encode kernels' guess "row i is string i" holds --; the strings of the other rows follow behind the n_rows entries, taken with
    // one atomic per block that has any.
    int count = 0;
    walk([&](int, int, int) { ++count; });
    const int extra = count > 1 ? count : 0;
    const int incl = wave_incl_sum(extra);
    const int total = wave_readlane(incl, kWave - 1);
    int base = 0;
    if (l == 0 && total > 0) base = atomicAdd(&status->n_out, total);
    base = wave_readlane(base, 0);
    if ((long long)base + total > cap) {
        if (l == 0) atomicOr(&status->flags, kFlagOutCapacity);
        if (valid) out_rb[row] = out_re[row] = 0;
        return;
    }
    int o = count > 1 ? in.n_rows + base + incl - extra : row;
    if (valid) {
        out_rb[row] = o;
        out_re[row] = o + count;
    }
    walk([&](int b, int e, int skip) {
        out_begins[o] = b;
        out_ends[o] = e;
        out_skips[o] = uint8_t(skip);
        ++o;
    });
}

}  // namespace ovtk
