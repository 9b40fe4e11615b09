// ops_kernels.hpp -- gfx950 kernels of WordpieceTokenizer, VocabEncoder, RaggedToDense, VocabDecoder,
// ByteFallback, FuzeRagged and the fused detokenizer.  All HBM-bound gather / scan / copy work.
#pragma once

#include "../../include/ovtk_amd.h"
#include "bpe_device.hpp"
#include "device_common.hpp"
#include "encode_kernels.hpp"

namespace ovtk {

// =============================================================================================
// WordpieceTokenizer (src/wordpiece_tokenizer.cpp:94-130): one wave per row, one lane per word.
// Word w owns staging entries [bytepos, bytepos + max(len,1)) of its row; unused ones hold kEmptyId and are
// squeezed out by compact_kernel.
// =============================================================================================
struct WordpieceDev {
    TrieDev root, sub;
    int32_t max_bytes;
    PieceStoreDev store;  // word -> ids of the words wordpiece_deferred_kernel had to walk the tries for (tables.hpp "piece store")
    PieceTableDev memo;   // the fused path's first-level word memo (lookup_span_kernel probes it): wordpiece_deferred_kernel files
                          // the short words it resolves there while memo.room lasts, so that they stop being deferred
};

// WordPiece of one word (wordpiece_tokenizer.cpp:100-126) into slot[0..): returns the id count (>= 1).
// The staging entries of one word / piece: i32, or u16 (S16) when the call stages two bytes per id (EncodeWork::stage16) -- a
// template flag: with a run-time branch in every put wordpiece_deferred_kernel took 54 us instead of 50.
template <bool S16>
struct StageSlot {
    const EncodeWork& w;
    int pos;
    __device__ __forceinline__ void put(int k, int32_t id) const {
        if (S16) reinterpret_cast<uint16_t*>(w.stage)[pos + k] = uint16_t(id);
        else w.stage[pos + k] = id;
    }
    __device__ __forceinline__ void clear(int k) const {
        if (S16) reinterpret_cast<uint16_t*>(w.stage)[pos + k] = uint16_t(0xFFFFu);
        else w.stage[pos + k] = kEmptyId;
    }
    __device__ __forceinline__ int32_t get(int k) const {
        if (S16) {
            const uint32_t x = reinterpret_cast<const uint16_t*>(w.stage)[pos + k];
            return x == 0xFFFFu ? kEmptyId : int32_t(x);
        }
        return w.stage[pos + k];
    }
};
// *is_unk (optional): the word has no segmentation and came out as unk_token_id -- as opposed to a word whose one token happens to be that id.
template <class GetByte, class Slot>
__device__ __forceinline__ int wordpiece_word(const WordpieceDev& T, const I2* root_lds, const I2* sub_lds, GetByte&& getb,
                                              int len, int32_t unk_id, const Slot& slot, bool* is_unk = nullptr) {
    if (is_unk) *is_unk = true;
    if (len > T.max_bytes || len <= 0) {  // strict > (:100-103); an empty word is undefined in the reference
        slot.put(0, unk_id);
        return 1;
    }
    int idx = 0, cnt = 0;
    int tok = trie_longest(T.root, root_lds, getb, len, idx);
    if (tok == -1) {
        slot.put(0, unk_id);
        return 1;
    }
    slot.put(cnt++, tok);
    while (idx < len) {
        tok = trie_longest(T.sub, sub_lds, getb, len, idx);
        if (tok == -1) {  // :118-123 the whole word becomes one unk
            slot.put(0, unk_id);
            return 1;
        }
        slot.put(cnt++, tok);
    }
    if (is_unk) *is_unk = false;
    return cnt;
}

static __global__ __launch_bounds__(kBlockThreads) void wordpiece_kernel(RowsIn in, WordpieceDev T, int32_t unk_id, EncodeWork w) {
    __shared__ I2 root_lds[256];
    __shared__ I2 sub_lds[256];
    for (int i = int(threadIdx.x); i < 256; i += kBlockThreads) {
        root_lds[i] = T.root.root[i];
        sub_lds[i] = T.sub.root[i];
    }
    __syncthreads();
    if (w.status->flags & (kFlagRange | kFlagStageOverflow)) return;
    const int l = lane_id();
    const int n_waves = w.n_waves;  // the geometry prep_rows_kernel summed the staging arenas over
    const int wave = int(blockIdx.x) * kWavesPerBlock + wave_in_block();
    int cursor = int(w.wave_off[wave]);
    for (int row = wave; row < in.n_rows; row += n_waves) {
        const int base = cursor;
        const int cb = in.ragged_begins[row], ce = in.ragged_ends[row];
        int bytepos = 0, emitted = 0;
        for (int c0 = cb; c0 < ce; c0 += kWave) {
            const int col = c0 + l;
            const bool valid = col < ce;
            int sb = 0, len = 0;
            if (valid) { sb = in.begins[col]; len = in.ends[col] - sb; }
            const int units = valid ? (len > 0 ? len : 1) : 0;
            const int incl = wave_incl_sum(units);
            const StageSlot<false> slot{w, base + bytepos + incl - units};   // (the op alone stages i32)
            int cnt = 0;
            if (valid) {
                const uint8_t* s = in.chars + sb;
                cnt = wordpiece_word(T, root_lds, sub_lds, [&](int i) -> uint32_t { return s[i]; }, len, unk_id, slot);
                for (int k = cnt; k < units; ++k) slot.clear(k);
            }
            emitted += wave_sum(cnt);
            bytepos += __shfl(incl, kWave - 1);
        }
        if (l == 0) {
            w.row_stage[row] = base;
            w.row_cnt[row] = emitted;
            w.row_used[row] = bytepos;
        }
        cursor += bytepos;
    }
}

// The words the fused BERT path could not resolve through the memo (encode_kernels.hpp lookup_kernel with the
// kSplitBertWords scanner): dense batches of 64 deferred words, one lane per word.
// tail_rows > 0: like merge_kernel, the block that finishes last also scans the per-tile id counts (no count_scan launch).
template <bool S16>
static __global__ __launch_bounds__(kBlockThreads) void wordpiece_deferred_kernel(RowsIn in, WordpieceDev T, int32_t unk_id,
                                                                                 EncodeWork w, int tail_rows, long long out_cap) {
    __shared__ I2 root_lds[256];
    __shared__ I2 sub_lds[256];
    // Everything the kernel has to know before it can start, asked for together: most of its blocks find no batch, and the ones
    // that do are one chain of memory round trips from here to their ticket -- five of them used to stand in front of the first
    // word (the tables' roots, the flags, the shard's count, the store's room, the memo's room: one wait each).
    const int l = lane_id();
    const int shard = int(blockIdx.y);
    PROBE(0);
    const uint32_t flags0 = __hip_atomic_load(&w.status->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int count = __hip_atomic_load(&w.status->shard_count[shard * kCounterStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int store_room = T.store.slots ? __hip_atomic_load(T.store.room, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int32_t* my_room = nullptr;   // the word memo's room, looked at once (a look per batch was a memory round trip per batch)
    int memo_room = 0;
    if (T.memo.room) {
        my_room = T.memo.room + ((blockIdx.x + 5u * blockIdx.y + uint32_t(wave_in_block())) & T.memo.room_mask) * kRoomStride;
        memo_room = __hip_atomic_load(my_room, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = int(threadIdx.x); i < 256; i += kBlockThreads) {
        root_lds[i] = T.root.root[i];
        sub_lds[i] = T.sub.root[i];
    }
    __syncthreads();
    PROBE(1);
    if (flags0 & (kFatalFlags | kFlagDeferOverflow)) return;
    if (tail_rows > 0 && !w.span_sums) fold_emitted_tile_sums(w, tail_rows);   // (span_sums: the lookup kernels have summed their rows themselves)
    PROBE(2);
    if (count > w.shard_cap) count = w.shard_cap;
    const DeferredPiece* list = w.deferred + (long long)shard * w.shard_cap;
    const int stride = int(gridDim.x) * kBlockThreads;
    // (a wave files at most its share of the room it found: merge_body says why)
    int store_budget = 0;
    {
        const int room = wave_uniform(store_room);
        const int n_waves_grid = int(gridDim.x) * int(gridDim.y) * kWavesPerBlock;
        store_budget = room > 0 ? (room + n_waves_grid - 1) / n_waves_grid : 0;
    }
    if (wave_uniform(memo_room) <= 0) my_room = nullptr;
    for (int base = (int(blockIdx.x) * kWavesPerBlock + wave_in_block()) * kWave; base < count; base += stride) {
        const bool valid = base + l < count;
        DeferredPiece e{};
        if (valid) e = list[base + l];
        int cnt = 0;
        // The word store first (the piece store of tables.hpp, keyed by the word): a word it holds is one probe instead of a
        // trie walk of one dependent load per byte.  What is filed there never depends on unk_token_id -- a word without a
        // segmentation is filed as one id kStoreUnk16 / kStoreUnk32 (no vocabulary index) and comes back as THIS call's input 8 --,
        // so the table stays valid whatever input 8 says on the next call.
        PROBE(3);   // (the batch's entries are here)
        const StageSlot<S16> out{w, e.stage_pos};
        uint32_t skey[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool keyed = T.store.slots && valid && e.len >= 1 && e.len <= kStoreKeyBytes;
        bool stored = false;
        if (keyed) {
            if (e.len <= kPieceKeyBytes) store_key_short(e.k0, e.k1, e.len, skey);
            else store_key_long(in.chars + e.begin, e.len, skey);
            uint32_t pay[8];
            const int c = T.store.narrow ? store_lookup<true>(T.store, skey, pay) : store_lookup<false>(T.store, skey, pay);
#ifdef OVTK_PROBE
            if (c >= 1 || c < 1) PROBE(8);   // (the lookup's loads are back: lanes in divergent code, lane 0 of the wave stamps if it is here)
#endif
            if (c >= 1) {
                stored = true;
                cnt = c;
                if (T.store.narrow) {
#pragma unroll
                    for (int k = 0; k < kStoreIds16; ++k)
                        if (k < c) out.put(k, k == 0 && store_id<true>(pay, 0) == kStoreUnk16 ? unk_id : store_id<true>(pay, k));
                } else {
#pragma unroll
                    for (int k = 0; k < kStoreIds32; ++k)
                        if (k < c) out.put(k, k == 0 && store_id<false>(pay, 0) == kStoreUnk32 ? unk_id : store_id<false>(pay, k));
                }
                for (int k = c; k < e.len; ++k) out.clear(k);
            }
        }
        PROBE(4);   // (the store has answered)
        bool is_unk = false;   // the word has no segmentation (as opposed to: its one token is the id input 8 names)
        if (valid && !stored) {
            if (e.len >= 1 && e.len <= kPieceKeyBytes) {
                const uint64_t k0 = e.k0, k1 = e.k1;
                cnt = wordpiece_word(
                    T, root_lds, sub_lds,
                    [&](int i) -> uint32_t { return uint32_t((i < 8 ? k0 >> (8 * i) : k1 >> (8 * (i - 8))) & 0xFF); }, e.len,
                    unk_id, out, &is_unk);
            } else {
                const uint8_t* s = in.chars + e.begin;
                cnt = wordpiece_word(T, root_lds, sub_lds, [&](int i) -> uint32_t { return s[i]; }, e.len, unk_id, out, &is_unk);
            }
            for (int k = cnt; k < e.len; ++k) out.clear(k);
        }
        if (store_budget > 0) {  // file what was walked: its ids come back from the lane's own staging entries
            const int max_ids = T.store.narrow ? kStoreIds16 : kStoreIds32;
            const bool want = keyed && !stored && cnt <= max_ids;
            if (__ballot(want)) {
                bool added = false;
                if (want) {
                    uint32_t pay[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < kStoreIds16; ++k) {
                        uint32_t v = k < cnt ? uint32_t(out.get(k)) : 0u;
                        if (k == 0 && is_unk) v = T.store.narrow ? uint32_t(kStoreUnk16) : uint32_t(kStoreUnk32);
                        if (T.store.narrow) pay[k >> 1] |= v << (16 * (k & 1));
                        else if (k < kStoreIds32) pay[k] = v;
                    }
                    added = T.store.narrow ? store_insert<true>(T.store, skey, pay, cnt) : store_insert<false>(T.store, skey, pay, cnt);
                }
                const int n_added = __popcll(__ballot(added));
                if (l == 0 && n_added) atomicAdd(T.store.room, -n_added);
                store_budget -= n_added;
            }
        }
        // The word memo learns them too (encode_kernels.hpp memo_insert, as merge_kernel does for the BPE pieces): a word of up to 15
        // bytes and three ids that is not unk is a hit of the lookup kernel from the next call
        // on, and no longer crosses the deferred list at all.  One atomic per wave takes the room.
        if (my_room) {
            // (not the words the store knew: they were offered to the memo when they were walked, and found their slot taken)
            const int memo_ids = T.memo.packed6 ? kPieceMaxIds6 : kPieceMaxIds;
            const bool keep = valid && !stored && e.len >= 1 && e.len <= kPieceKeyBytes && cnt >= 1 && cnt <= memo_ids && !is_unk;   // (the memo has no "input 8" entry)
            const unsigned long long km = __ballot(keep);
            if (km) {
                int left = 0;
                if (l == 0) left = atomicAdd(my_room, -int(__popcll(km)));
                left = wave_readlane(left, 0);
                bool added = false;
                if (keep && rank_below(km) < left) {
                    int32_t t3[kPieceMaxIds] = {0, 0, 0};
                    if (T.memo.packed6) {
#pragma unroll
                        for (int k = 0; k < kPieceMaxIds6; ++k)
                            if (k < cnt) t3[k >> 1] |= int32_t(uint32_t(out.get(k) & 0xFFFF) << (16 * (k & 1)));
                    } else {
#pragma unroll
                        for (int k = 0; k < kPieceMaxIds; ++k) t3[k] = k < cnt ? out.get(k) : 0;
                    }
                    added = memo_insert(T.memo, e.k0, e.k1, t3, cnt);
                }
                const int unused = __popcll(km) - __popcll(__ballot(added));  // room taken but not filled goes back
                if (l == 0 && unused) atomicAdd(my_room, unused);
            }
        }
        const int incl = wave_incl_sum(cnt);
        const int my_row = valid ? e.row : -1;
        const int prev_row = __shfl_up(my_row, 1), next_row = __shfl_down(my_row, 1);
        const bool head = l == 0 || prev_row != my_row, tail = l == kWave - 1 || next_row != my_row;
        const int seg_base = wave_incl_max(head ? incl - cnt : 0);
        if (valid && tail && incl - seg_base > 0) {
            atomicAdd(&w.row_cnt[e.row], incl - seg_base);
            if (w.tile_cnt) atomicAdd(&w.tile_cnt[e.row / kRowTile], incl - seg_base);
        }
    }
    PROBE(5);   // (out of the batch loop)
    if (tail_rows <= 0 || w.tile_sums) return;   // (tile_sums: compact_kernel sums tile_cnt itself, merge_body)
    __syncthreads();
    const bool last_ = last_block_done_sharded(w.status, int(blockIdx.y), gridDim.x, /*release=*/false);  // only atomics to hand over (grid: blocks per shard x kShards)
    PROBE(6);   // (ticket drawn)
    if (!last_) return;
    scan_tiles_one_block(tail_rows, w, out_cap);
    PROBE(7);
}

// =============================================================================================
// VocabEncoder (src/vocab_encoder.cpp:88-91): one lane per element, open addressing.  A string of up to 16 bytes (nearly
// every word) is fetched as the aligned dwords that hold it -- five loads instead of one per byte -- and hashed / compared
// from registers; the key it is compared with is fetched the same way.  Longer strings take the byte loops.
// =============================================================================================
// Bytes [p, p + len) (len <= 16) as four little-endian words, zero-padded; false when an aligned dword that holds them
// would reach outside [buf, buf_end) (the caller then reads bytes).
__device__ __forceinline__ bool load_words16(const uint8_t* p, int len, const uint8_t* buf, const uint8_t* buf_end, uint32_t (&w)[4]) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    const uint8_t* a = p - (addr & 3);
    const int sh = int(addr & 3) * 8;
    const int nw = (int(addr & 3) + len + 3) >> 2;  // 1..5 dwords
    if (a < buf || a + 4 * nw > buf_end) return false;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(a);
    uint32_t r[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) r[j] = j < nw ? q[j] : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t x = funnel_shr(r[j], r[j + 1], sh);
        const int have = len - 4 * j;
        w[j] = have >= 4 ? x : (have <= 0 ? 0u : (x & ((1u << (8 * have)) - 1u)));
    }
    return true;
}
template <typename T>
static __global__ __launch_bounds__(kBlockThreads) void vocab_encoder_kernel(const int32_t* begins, const int32_t* ends,
                                                                      const uint8_t* chars, long long n_chars, int n,
                                                                      StringMapDev M, long long n_key_chars, T dflt, T* out,
                                                                      RunStatus* status) {
    const int stride = int(gridDim.x) * kBlockThreads;
    const T* values = static_cast<const T*>(M.values);
    for (int i = int(blockIdx.x) * kBlockThreads + int(threadIdx.x); i < n; i += stride) {
        const long long b = begins[i], e = ends[i];
        if (b < 0 || e < b || e > n_chars) {
            atomicOr(&status->flags, kFlagRange);
            continue;
        }
        const uint8_t* s = chars + b;
        const int len = int(e - b);
        uint32_t w[4] = {0, 0, 0, 0};
        const bool packed = len <= 16 && load_words16(s, len, chars, chars + n_chars, w);
        uint32_t h;
        if (packed) {
            h = 2166136261u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * j < len) h = hash_word_step(h, w[j]);
            h = hash_finish(h, len);
        } else {
            h = hash_bytes(s, len);
        }
        uint32_t p = h & M.mask;
        T val = dflt;
        for (;;) {
            const uint64_t slot = M.slots[p];
            if (slot == kEmptySlot) break;
            if (uint32_t(slot >> 32) == h) {
                const int k = int(uint32_t(slot));
                const int kb = M.key_begins[k];
                if (M.key_ends[k] - kb == len) {
                    bool same = true;
                    uint32_t kw[4];
                    if (packed && load_words16(M.key_chars + kb, len, M.key_chars, M.key_chars + n_key_chars, kw))
                        same = kw[0] == w[0] && kw[1] == w[1] && kw[2] == w[2] && kw[3] == w[3];
                    else
                        for (int j = 0; j < len && same; ++j) same = M.key_chars[kb + j] == s[j];
                    if (same) { val = values[k]; break; }
                }
            }
            p = (p + 1) & M.mask;
        }
        out[i] = val;
    }
}

// =============================================================================================
// RaggedToDense (src/ragged_to_dense.cpp:129-167): one wave per row, byte-granular cells.
// =============================================================================================
struct DenseArgs {
    const int32_t* begins;
    const int32_t* ends;
    int32_t n_rows;
    const uint8_t* data;
    long long n_data;    // ragged elements
    int32_t cell;        // bytes per ragged element (elem_size * inner)
    int32_t elem_size;
    int32_t inner;
    int32_t target;
    int32_t pad_right;
    int32_t pad_max_length;
    uint8_t dflt[16];
    uint8_t* out;
    uint8_t* mask;       // may be nullptr
    RunStatus* status;
};

static __global__ __launch_bounds__(kBlockThreads) void ragged_to_dense_kernel(DenseArgs a) {
    const int l = lane_id();
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int row = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); row < a.n_rows; row += n_waves) {
        const long long b = a.begins[row];
        const long long len = (long long)a.ends[row] - b;
        // :132-133: with pad_max_length the copy is `target` long whatever the row holds
        // size_t(len) in the reference: a negative length behaves like a huge one
        const long long take = (a.pad_max_length || len < 0 || len > a.target) ? a.target : len;
        if (b < 0 || b + take > a.n_data) {
            if (l == 0) atomicOr(&a.status->flags, kFlagRange);
            continue;
        }
        const long long pad = a.target - take;
        const long long first = a.pad_right ? 0 : pad;  // index of the first copied element in the output row
        uint8_t* orow = a.out + (long long)row * a.target * a.cell;
        uint8_t* mrow = a.mask ? a.mask + (long long)row * a.target * a.inner : nullptr;
        const uint8_t* src = a.data + b * a.cell;
        if (a.elem_size == 4 && a.inner == 1) {  // the common case (i32 ids): one dword per lane
            uint32_t d;
            d = uint32_t(a.dflt[0]) | uint32_t(a.dflt[1]) << 8 | uint32_t(a.dflt[2]) << 16 | uint32_t(a.dflt[3]) << 24;
            for (long long k = l; k < a.target; k += kWave) {
                const bool data = k >= first && k < first + take;
                reinterpret_cast<uint32_t*>(orow)[k] = data ? reinterpret_cast<const uint32_t*>(src)[k - first] : d;
                if (mrow) mrow[k] = data ? 1 : 0;
            }
        } else {
            const long long row_bytes = (long long)a.target * a.cell;
            for (long long byte = l; byte < row_bytes; byte += kWave) {
                const long long k = byte / a.cell;
                const bool data = k >= first && k < first + take;
                orow[byte] = data ? src[byte - first * a.cell] : a.dflt[(byte % a.cell) % a.elem_size];
            }
            if (mrow)
                for (long long m = l; m < (long long)a.target * a.inner; m += kWave) {
                    const long long k = m / a.inner;
                    mrow[m] = (k >= first && k < first + take) ? 1 : 0;
                }
        }
    }
}

// The common case -- 4-byte elements (token ids), the whole output below 2^31 elements and 16-byte aligned -- as a flat
// streaming kernel: a thread produces FOUR consecutive output elements (one 16-byte store of ids, one 4-byte store of mask
// bytes) wherever they fall in the [rows, target] grid; a group of four may straddle two rows.  Every store instruction of
// a wave is a full 1 KB (ids) / 256 B (mask) of consecutive bytes; the wave-per-row form above stores 4 bytes and ONE
// mask byte per lane and leaves a third of its lanes idle on the last 64 columns of a 150-wide row (42 -> 2x us at
// config-2 size).
static __global__ __launch_bounds__(kBlockThreads) void ragged_to_dense_flat4_kernel(DenseArgs a) {
    const unsigned total = unsigned(a.n_rows) * unsigned(a.target);
    const unsigned T = unsigned(a.target);
    const uint32_t dflt = uint32_t(a.dflt[0]) | uint32_t(a.dflt[1]) << 8 | uint32_t(a.dflt[2]) << 16 | uint32_t(a.dflt[3]) << 24;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.data);
    const unsigned stride = gridDim.x * kBlockThreads * 4u;
    for (unsigned idx = (blockIdx.x * kBlockThreads + threadIdx.x) * 4u; idx < total; idx += stride) {
        unsigned row = idx / T, k = idx - row * T;
        uint32_t v[4];
        uint32_t m = 0;
        long long b = 0, take = 0, first = 0;
        bool have = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (idx + unsigned(j) < total) {
                if (!have) {  // the row's range (again after the group crossed into the next row)
                    b = a.begins[row];
                    const long long len = (long long)a.ends[row] - b;
                    take = (a.pad_max_length || len < 0 || len > a.target) ? a.target : len;
                    first = a.pad_right ? 0 : a.target - take;
                    if (b < 0 || b + take > a.n_data) {
                        atomicOr(&a.status->flags, kFlagRange);
                        take = 0;
                    }
                    have = true;
                }
                const bool data = (long long)k >= first && (long long)k < first + take;
                v[j] = data ? src[b + k - first] : dflt;
                m |= (data ? 1u : 0u) << (8 * j);
                if (++k == T) {
                    k = 0;
                    ++row;
                    have = false;
                }
            } else {
                v[j] = 0;
            }
        }
        if (idx + 4u <= total) {
            *reinterpret_cast<uint4*>(a.out + (size_t)idx * 4) = make_uint4(v[0], v[1], v[2], v[3]);
            if (a.mask) *reinterpret_cast<uint32_t*>(a.mask + idx) = m;
        } else {
            for (unsigned j = 0; idx + j < total; ++j) {
                reinterpret_cast<uint32_t*>(a.out)[idx + j] = v[j];
                if (a.mask) a.mask[idx + j] = uint8_t(m >> (8 * j));
            }
        }
    }
}

// Finalisation of the char-offset scans of VocabDecoder / ByteFallback: total -> status->n_out + capacity flag.
struct CharsFin {
    RunStatus* status;
    long long cap;
    __device__ void operator()(long long total) const {
        status->n_out = total > INT32_MAX ? INT32_MAX : int32_t(total);
        if (total > cap) atomicOr(&status->flags, kFlagOutCapacity);
    }
};

// =============================================================================================
// VocabDecoder / ByteFallback / detokenize element functors
// =============================================================================================
// Per-vocabulary decode tables (built at create): v_len[id] = bytes token id contributes, v_pack[id] = its first 16
// bytes.  Two flavours per handle: plain VocabDecoder, and VocabDecoder followed by ByteFallback ("<0xHH>" -> one byte).
struct alignas(16) TokenPack { uint32_t w[4]; };
struct DecodeDev {
    const int32_t* ids;          // [batch * seq]
    const int32_t* v_begins;
    const uint8_t* v_chars;
    const uint16_t* v_len;       // output bytes of the token (after ByteFallback in that flavour; 0 when it is skipped)
    const TokenPack* v_pack;     // its first min(len, 15) output bytes, zero padded; byte 15 = len (0xFF: 16 or more, see v_len)
    int32_t len_in_pack;         // 0: a call's own skip list is in force, lengths come from v_len only
    int32_t vocab_size;
};

// Length in bytes of a token's text (vocab_decoder.cpp:70-81): ids outside [0, V) or in the skip list give "".
__device__ __forceinline__ int decode_len(const DecodeDev& d, int32_t id) {
    const bool in_vocab = uint32_t(id) < uint32_t(d.vocab_size);  // `token_id < vocab_size` compares as size_t (:71)
    const int len = d.v_len[in_vocab ? uint32_t(id) : 0u];        // ONE lookup per token: skipped tokens have length 0 in the table
    return in_vocab ? len : 0;
}

// VocabDecoder / fused detokenizer as two wave-per-segment passes around one scan: a segment = up to kSegTokens
// consecutive tokens of one row.  Pass 1 sums the segment's output bytes; the scan (scan_kernels.hpp) turns them
// into offsets (and, for the fused form, into the row's begin / end = what FuzeRagged picks, fuze.cpp:35-38);
// pass 2 re-reads the ids (16 B per lane), ranks the token lengths with a wave prefix sum, gathers each token's
// packed bytes with ONE 16-byte load, assembles the segment's text in LDS and writes it out with coalesced dword stores.
constexpr int kSegTokens = 512;
constexpr int kSegLdsBytes = 4096;  // segments whose text is longer are copied token by token

// The 4 ids of lane l in the group of 256 tokens starting at token g of the row (ids beyond t1 read as -1 = "no text").
__device__ __forceinline__ void load_ids4(const DecodeDev& d, long long row_base, int g, int t1, int32_t (&id)[4]) {
    const int t = g + 4 * lane_id();
    const int32_t* p = d.ids + row_base + t;
    if (t + 4 <= t1 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const int4 v = *reinterpret_cast<const int4*>(p);
        id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) id[j] = t + j < t1 ? p[j] : -1;
    }
}

static __global__ __launch_bounds__(kBlockThreads) void decode_count_kernel(DecodeDev d, int seq, int n_seg, long long n_units,
                                                                            long long* unit_bytes) {
    const int l = lane_id();
    const long long my_waves = (long long)gridDim.x * kWavesPerBlock;
    for (long long u = (long long)blockIdx.x * kWavesPerBlock + wave_in_block(); u < n_units; u += my_waves) {
        const long long row = u / n_seg;
        const int seg = int(u - row * n_seg);
        const int t0 = seg * kSegTokens, t1 = t0 + kSegTokens < seq ? t0 + kSegTokens : seq;
        // a segment is two groups of 256 tokens: both id loads are issued before the first length lookup, and all eight
        // lookups of a lane before the first sum (the kernel was bound by dependent round trips, not by bandwidth)
        static_assert(kSegTokens == 8 * kWave, "two groups of 4 ids per lane");
        int32_t id[2][4];
        load_ids4(d, row * seq, t0, t1, id[0]);
        if (t0 + 4 * kWave < t1) {
            load_ids4(d, row * seq, t0 + 4 * kWave, t1, id[1]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) id[1][j] = -1;
        }
        int n[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) n[h][j] = decode_len(d, id[h][j]);
        int s = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += n[h][j];
        s = wave_sum(s);
        if (l == 0) unit_bytes[u] = s;
    }
}

// The same pass for vocabularies of at most 65 536 tokens: every block first copies the length table (2 bytes per token)
// into LDS, so the per-token lookup is an LDS read instead of an address-divergent global gather (the texture
// addresser, not HBM, bounded the kernel above: 64 cache lines per wave-instruction).  One 1024-thread block per CU.
constexpr int kLenLdsTokens = 65536;
constexpr int kCountLdsThreads = 1024;
static __global__ __launch_bounds__(kCountLdsThreads) void decode_count_lds_kernel(DecodeDev d, int seq, int n_seg, long long n_units,
                                                                                   long long* unit_bytes) {
    __shared__ uint16_t len_lds[kLenLdsTokens];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(d.v_len);  // tables are padded to an even token count
        uint32_t* dst = reinterpret_cast<uint32_t*>(len_lds);
        for (int i = int(threadIdx.x); i < (d.vocab_size + 1) / 2; i += kCountLdsThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int l = lane_id();
    const long long my_waves = (long long)gridDim.x * (kCountLdsThreads / kWave);
    for (long long u = (long long)blockIdx.x * (kCountLdsThreads / kWave) + wave_in_block(); u < n_units; u += my_waves) {
        const long long row = u / n_seg;
        const int seg = int(u - row * n_seg);
        const int t0 = seg * kSegTokens, t1 = t0 + kSegTokens < seq ? t0 + kSegTokens : seq;
        int32_t id[2][4];
        load_ids4(d, row * seq, t0, t1, id[0]);
        if (t0 + 4 * kWave < t1) {
            load_ids4(d, row * seq, t0 + 4 * kWave, t1, id[1]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) id[1][j] = -1;
        }
        int s = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s += uint32_t(id[h][j]) < uint32_t(d.vocab_size) ? int(len_lds[id[h][j]]) : 0;
        s = wave_sum(s);
        if (l == 0) unit_bytes[u] = s;
    }
}

struct UnitLen {
    const long long* unit_bytes;
    __device__ long long operator()(long long u) const { return unit_bytes[u]; }
};
// unit_off[u] = offset; fused form (row_begins != nullptr): first / last segment of a row give its begin / end.
struct UnitApply {
    long long* unit_off;
    int n_seg;
    int32_t* row_begins;
    int32_t* row_ends;
    __device__ void operator()(long long u, long long off, long long len) const {
        unit_off[u] = off;
        if (!row_begins) return;
        const long long row = u / n_seg;
        const int seg = int(u - row * n_seg);
        if (seg == 0) row_begins[row] = int32_t(off);
        if (seg == n_seg - 1) row_ends[row] = int32_t(off + len);
    }
};

static __global__ __launch_bounds__(kBlockThreads) void decode_write_kernel(DecodeDev d, int seq, int n_seg, long long n_units,
                                                                            const long long* unit_off,
                                                                            const long long* unit_bytes, int32_t* tok_begins,
                                                                            int32_t* tok_ends, uint8_t* out_chars,
                                                                            const RunStatus* status) {
    __shared__ uint32_t seg_all[kWavesPerBlock][kSegLdsBytes / 4 + 2];
    if (status->flags & (kFlagOutCapacity | kFlagRange)) return;
    const int l = lane_id();
    uint8_t* buf = reinterpret_cast<uint8_t*>(seg_all[wave_in_block()]);
    const long long my_waves = (long long)gridDim.x * kWavesPerBlock;
    for (long long u = (long long)blockIdx.x * kWavesPerBlock + wave_in_block(); u < n_units; u += my_waves) {
        const long long row = u / n_seg;
        const int seg = int(u - row * n_seg);
        const int t0 = seg * kSegTokens, t1 = t0 + kSegTokens < seq ? t0 + kSegTokens : seq;
        const long long base = unit_off[u];
        const int total = int(unit_bytes[u]);
        const int skew = int((reinterpret_cast<uintptr_t>(out_chars) + base) & 3);  // LDS byte skew + k <-> output byte base + k: dwords line up
        const bool staged = total + skew <= kSegLdsBytes;
        int run = 0;
        uint32_t* wbuf = seg_all[wave_in_block()];
        wave_sync();  // the previous segment's flush is done with the buffer
        if (staged) {
            for (int q = l; q < ((skew + total + 3) >> 2) + 1; q += kWave) wbuf[q] = 0;
            wave_sync();
        }
        for (int g = t0; g < t1; g += 4 * kWave) {
            int32_t id[4];
            load_ids4(d, row * seq, g, t1, id);
            int n[4];
            TokenPack pk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // ONE address-divergent load per token (the pack carries the length): the pass is bound by the texture
                // addresser (TA busy 80 %), not by bandwidth
                const bool in_vocab = uint32_t(id[j]) < uint32_t(d.vocab_size);
                pk[j] = in_vocab ? d.v_pack[id[j]] : TokenPack{{0, 0, 0, 0}};
                const int code = int(pk[j].w[3] >> 24);
                pk[j].w[3] &= 0x00FFFFFFu;
                if (d.len_in_pack) n[j] = code == 0xFF ? int(d.v_len[id[j]]) : code;
                else n[j] = decode_len(d, id[j]);
            }
            const int mine = n[0] + n[1] + n[2] + n[3];
            const int incl = wave_incl_sum(mine);
            int off = run + incl - mine;
            if (tok_begins) {
                const int t = g + 4 * l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (t + j < t1) {
                        tok_begins[row * seq + t + j] = int32_t(base + off + (j > 0 ? n[0] : 0) + (j > 1 ? n[1] : 0) + (j > 2 ? n[2] : 0));
                        tok_ends[row * seq + t + j] = int32_t(base + off + n[0] + (j > 0 ? n[1] : 0) + (j > 1 ? n[2] : 0) + (j > 2 ? n[3] : 0));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n[j] > 0) {
                    if (staged) {
                        // the token's (zero padded) first 16 bytes, shifted to its byte position, are OR-ed dword by
                        // dword into the zeroed buffer: ~2 LDS operations per token instead of one per byte
                        const int o = skew + off, sh = (o & 3) * 8, dw = o >> 2;
                        const int m = n[j] < 15 ? n[j] : 15;
                        const int nd = (((o & 3) + m) + 3) >> 2;  // dwords touched, 1..5
                        uint32_t prev = 0;
#pragma unroll
                        for (int q = 0; q < 5; ++q) {
                            const uint32_t cur = q < 4 ? pk[j].w[q] : 0u;
                            const uint32_t v = sh ? ((cur << sh) | (prev >> (32 - sh))) : cur;
                            if (q < nd && v) atomicOr(&wbuf[dw + q], v);
                            prev = cur;
                        }
                        if (n[j] > 15) {
                            uint8_t* dst = buf + o;
                            const uint8_t* src = d.v_chars + d.v_begins[id[j]];
                            for (int k = 15; k < n[j]; ++k) dst[k] = src[k];
                        }
                    } else {  // oversized segment: straight to the output, byte by byte
                        const uint8_t* src = d.v_chars + d.v_begins[id[j]];
                        uint8_t* dst = out_chars + base + off;
                        if (n[j] == 1) dst[0] = uint8_t(pk[j].w[0]);  // (also the ByteFallback byte)
                        else for (int k = 0; k < n[j]; ++k) dst[k] = src[k];
                    }
                    off += n[j];
                }
            }
            run += wave_readlane(incl, kWave - 1);
        }
        if (!staged) continue;
        wave_sync();
        // flush: output bytes [base, base + total) = buffer bytes [skew, skew + total); whole dwords inside that range
        // go out as dwords (the buffer's dword grid is the output's), the ragged ends byte by byte
        uint8_t* gout = out_chars + base - skew;
        const int lo = skew, hi = skew + total;
        const int d0 = (lo + 3) >> 2, d1 = hi >> 2;  // dwords [d0, d1) lie inside
        for (int k = d0 + l; k < d1; k += kWave) reinterpret_cast<uint32_t*>(gout)[k] = wbuf[k];
        if (d0 <= d1) {
            if (l < 4 * d0 - lo) gout[lo + l] = buf[lo + l];                   // head: bytes [lo, 4*d0)
            if (l < hi - 4 * d1) gout[4 * d1 + l] = buf[4 * d1 + l];           // tail: bytes [4*d1, hi)
        } else if (l < total) {                                                // the whole text sits inside one dword
            gout[lo + l] = buf[lo + l];
        }
    }
}

// ByteFallback (src/byte_fallback.cpp:33-46): a 6-byte token whose only '<' is at 0 and which ends in '>' becomes ONE
// byte: PieceToByte's value for the spellings "<0x%02X>" (upper-case hex, sentence_piece.cpp:27-46), otherwise
// PieceToByte's -1 stored into a uint8_t = 0xFF.  Returns -1: copy verbatim; 0..255: the byte.
__device__ __forceinline__ int byte_fallback_value(const uint8_t* s, int len) {
    if (len != 6 || s[0] != '<' || s[5] != '>') return -1;
    for (int k = 1; k < 6; ++k)
        if (s[k] == '<') return -1;  // rfind('<') must be 0
    if (s[1] != '0' || s[2] != 'x') return 255;
    int v = 0;
    for (int k = 3; k < 5; ++k) {
        const uint8_t c = s[k];
        int h;
        if (c >= '0' && c <= '9') h = c - '0';
        else if (c >= 'A' && c <= 'F') h = c - 'A' + 10;
        else return 255;
        v = v * 16 + h;
    }
    return v;
}

struct FallbackLen {
    const int32_t* begins;
    const int32_t* ends;
    const uint8_t* chars;
    long long n_chars;
    __device__ long long operator()(long long i) const {
        const long long b = begins[i], e = ends[i];
        if (b < 0 || e < b || e > n_chars) return 0;  // flagged by check_strings_kernel
        return byte_fallback_value(chars + b, int(e - b)) >= 0 ? 1 : e - b;
    }
};
struct FallbackApply {
    const int32_t* begins;
    const int32_t* ends;
    const uint8_t* chars;
    int32_t* out_begins;
    int32_t* out_ends;
    uint8_t* out_chars;
    __device__ void operator()(long long i, long long off, long long len) const {
        out_begins[i] = int32_t(off);
        out_ends[i] = int32_t(off + len);
        if (len == 0) return;
        const uint8_t* s = chars + begins[i];
        const int v = byte_fallback_value(s, ends[i] - begins[i]);
        if (v >= 0) { out_chars[off] = uint8_t(v); return; }
        for (long long k = 0; k < len; ++k) out_chars[off + k] = s[k];
    }
};

static __global__ __launch_bounds__(kBlockThreads) void check_strings_kernel(const int32_t* begins, const int32_t* ends,
                                                                      long long n, long long n_chars, RunStatus* status) {
    const long long stride = (long long)gridDim.x * kBlockThreads;
    for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += stride) {
        const long long b = begins[i], e = ends[i];
        if (b < 0 || e < b || e > n_chars) atomicOr(&status->flags, kFlagRange);
    }
}

// Row offsets of the decoder: ragged_begins[b] = b * S', ragged_ends[b] = (b + 1) * S' (vocab_decoder.cpp:58-59).
static __global__ __launch_bounds__(kBlockThreads) void decoder_rows_kernel(int batch, int sp, int32_t* rb, int32_t* re) {
    const int stride = int(gridDim.x) * kBlockThreads;
    for (int b = int(blockIdx.x) * kBlockThreads + int(threadIdx.x); b < batch; b += stride) {
        rb[b] = b * sp;
        re[b] = (b + 1) * sp;
    }
}

// FuzeRagged (src/fuze.cpp:35-38).
static __global__ __launch_bounds__(kBlockThreads) void fuze_kernel(const int32_t* rb, const int32_t* re, int n_rows,
                                                             const int32_t* begins, const int32_t* ends, int n,
                                                             int32_t* out_begins, int32_t* out_ends, RunStatus* status) {
    const int stride = int(gridDim.x) * kBlockThreads;
    for (int r = int(blockIdx.x) * kBlockThreads + int(threadIdx.x); r < n_rows; r += stride) {
        const int bi = rb[r], ei = re[r] > rb[r] ? re[r] - 1 : re[r];
        if (bi < 0 || bi >= n || ei < 0 || ei >= n) {  // the reference reads past the tensor here (undefined)
            atomicOr(&status->flags, kFlagRange);
            continue;
        }
        out_begins[r] = begins[bi];
        out_ends[r] = ends[ei];
    }
}

// ------------------------------------------------------------------------------- TrieTokenizer
// src/trie_tokenizer.cpp:66-78: per string the greedy chain of longest matches (Trie::find_longest, src/utils.cpp:517-538).
// A chain is a run of dependent table reads -- 1.4 us per step on this chip whatever the occupancy (a lane per row: 1.14 ms for a
// config-2 batch; the same rows spread over eight times the waves: 0.97) --, so the row is cut into SEGMENTS of 64 bytes, a lane each
// (round 6):
//   * trie_segments_kernel: every segment's lane walks the chain that STARTS AT ITS FIRST BYTE -- a guess for every segment but a
//     string's first -- until a token ends behind the segment; the token that starts at byte p of the row goes to entry p of the row's
//     staging stretch (a token takes at least a byte), bit p % 64 of the segment's mask says so, and the position behind the segment's
//     last token is filed;
//   * trie_rows_kernel, a lane per row: the true chain enters segment k where segment k - 1's true chain left it.  Greedy chains that
//     meet stay together, and they meet soon (a guess that starts inside a word ends with that word): where the entry is a token start
//     of the segment's guess, the guess from there on IS the true chain (the mask's lower bits are dropped); where not, the lane walks
//     on from the entry until it steps on a token start of the guess or leaves the segment.  The row's count is the kept bits;
//   * the scan of the counts is the reference's running ragged_offset; trie_gather_kernel, a wave per row, moves the kept entries to
//     their places.
// Rows of several strings (the guess would have to know where the strings start) are walked whole by their lane of trie_rows_kernel,
// into the same masks.  Until round 6: one walk per row (1.14 ms), before that a counting walk, the scan, a writing walk (2.85 ms).
constexpr int kTrieSeg = 64;
constexpr int32_t kTrieBroken = INT32_MIN;   // in a segment's filed exit: no token matches at the position in the lower bits
// the chars tensor's last bytes, fewer than sixteen: byte by byte (out of line: a path of a few lanes per batch, kept out of the walk's loop)
__device__ __attribute__((noinline)) static uint4 trie_window_tail(const uint8_t* chars, long long n_chars, long long base) {
    uint64_t lo = 0, hi = 0;
    for (long long k = base; k < n_chars; ++k) {
        const uint64_t v = uint64_t(chars[k]) << (8 * ((k - base) & 7));
        if (k - base < 8) lo |= v;
        else hi |= v;
    }
    return uint4{uint32_t(lo), uint32_t(lo >> 32), uint32_t(hi), uint32_t(hi >> 32)};
}
struct __attribute__((packed, aligned(1))) TrieBytes16 { uint32_t d[4]; };
// Sixteen bytes of the chars tensor per lane, in LDS: a step reads its byte with one ds_read_u8 (the byte load from global memory in front of
// every trie step was a dependent load of its own; the window in registers, a 64-bit shift per byte).
struct TrieWindow {
    uint8_t* mine;       // the lane's sixteen bytes (LDS)
    long long at = -1;   // chars offset of the window (a multiple of 16), -1: none
};
struct TrieRows {
    const int32_t* ragged_begins;
    const int32_t* ragged_ends;
    const int32_t* begins;
    const int32_t* ends;
    const uint8_t* chars;
    long long n_strings, n_chars;
    TrieBucketsDev trie;
    RunStatus* status;

    // bytes of the row's strings, -1 for offsets outside their tensors
    __device__ long long row_bytes(long long row) const {
        const long long cb = ragged_begins[row], ce = ragged_ends[row];
        if (cb < 0 || ce < cb || ce > n_strings) return -1;
        long long sum = 0;
        for (long long col = cb; col < ce; ++col) {
            const long long b = begins[col], e = ends[col];
            if (b < 0 || e < b || e > n_chars) return -1;
            sum += e - b;
        }
        return sum;
    }

    // The greedy chain over the string chars[b, b + n) from position `from` (Trie::find_longest, src/utils.cpp:517-538, token after token),
    // as a FLAT loop of ONE 16-BYTE READ PER TURN, whatever the lane is at:
    //   * the sixteen bytes of text round the byte at hand, when the lane's window does not hold it (a turn in sixteen), or
    //   * half a bucket of the edge table: the edge (node, byte) is there -- on to the next byte --, or an entry of the half is free -- the
    //     node has no such edge: the token found so far ends (emit, back to the root: part of the same turn) --, or the half is full of
    //     other edges: the next turn reads the other half / the next bucket.
    // The root is a node of the bucket table like the others (kTrieRoot): no branch for a token's first byte.
    // Measured on a config-2 batch's 590 000 segments (profiles/r06/n_trie_*): the open-addressed 16-byte edges with the root table in LDS
    // and the text window in registers 444 us (103 vector + 102 scalar instructions per turn of a wave: both branches, the window's
    // refill, the probe loop); buckets read whole 470; buckets by halves, the second half and the refill out of line 343; this form 359
    // -- same time, a third of the code.  72 M vector + 61 M scalar instructions per launch either way: a wave's turn is ~160
    // instructions for one table read, most of them the bookkeeping of lanes that are at different points of their tokens.
    // The chain goes on while `more(position)` says so for the position behind a token (and the string has bytes left); returns that
    // position, or kTrieBroken | position where no token matches (the reference spins there forever, src/trie_tokenizer.cpp:72-75).
    template <class Emit, class More>
    __device__ __forceinline__ int32_t chain(long long b, int n, int from, TrieWindow& tw, Emit&& emit, More&& more) const {
        if (from >= n) return from;
        int idx = from;   // where the current token starts
        int i = from;     // the position of the byte at hand
        int cur = kTrieRoot, best = -1, best_end = 0;
        uint32_t key = 0, bk = 0, half = 0;
        bool text = true;   // this turn reads text
        // the byte at position i is in the window: its edge's key and bucket
        auto aim = [&]() {
            const uint32_t byte = tw.mine[(b + i) & 15];
            key = (uint32_t(cur) << 8) | byte;
#ifdef OVTK_SIMT_EMULATOR
            bk = trie_bucket_of(uint32_t(cur), byte, trie.bucket_mask);
#else
            const uint32_t h = __umul24(uint32_t(cur), 0x9E3779u) + __umul24(byte, 0x85EBCBu);   // trie_bucket_of()
            bk = ((h >> 9) ^ h) & trie.bucket_mask;
#endif
            half = 0;
        };
        if (((b + i) & ~15ll) == tw.at) {
            text = false;
            aim();
        }
        for (;;) {
            const long long base = (b + i) & ~15ll;
            uint4 v;
            if (text && base + 16 > n_chars) {   // the chars tensor's last bytes (a lane or two per batch)
                v = trie_window_tail(chars, n_chars, base);
            } else {
                const char* at = text ? reinterpret_cast<const char*>(chars) + base : reinterpret_cast<const char*>(trie.buckets) + ((size_t(bk) << 5) | (half << 4));
                const TrieBytes16 w = *reinterpret_cast<const TrieBytes16*>(at);   // (the text may start at any address: gfx950 reads 16 bytes at any alignment)
                v = uint4{w.d[0], w.d[1], w.d[2], w.d[3]};
            }
#ifndef OVTK_SIMT_EMULATOR
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // (one 16-byte read: left alone, the compiler fetches an entry's value in a second round trip)
#endif
            if (text) {
                *reinterpret_cast<uint4*>(tw.mine) = v;
                tw.at = base;
                text = false;
                aim();
                continue;
            }
            const bool m0 = (v.x & ~kTrieKids) == key, m1 = (v.z & ~kTrieKids) == key;
            bool stop = true;
            if (m0 | m1) {
                const int value = int(m0 ? v.y : v.w);
                cur = int(4u * bk + 2u * half + (m0 ? 0u : 1u));
                ++i;
                if (value != -1) {
                    best = value;
                    best_end = i;
                }
                stop = (((m0 ? v.x : v.z) & kTrieKids) == 0) || i >= n;
            } else if (v.x != kTrieFree && v.z != kTrieFree) {   // a full half: the edge may be behind it
                bk = (bk + half) & trie.bucket_mask;
                half ^= 1u;
                continue;
            }
            if (stop) {
                if (best == -1) return kTrieBroken | idx;
                emit(idx, best);
                idx = i = best_end;
                cur = kTrieRoot;
                best = -1;
                if (idx >= n || !more(idx)) return idx;
            }
            if (((b + i) & ~15ll) != tw.at) text = true;
            else aim();
        }
    }
};
// a row's staging stretch: its bytes rounded up to whole segments (so that stretch offset / 64 numbers the segments of the batch)
struct TrieRowStretch {
    TrieRows r;
    __device__ long long operator()(long long row) const {
        const long long n = r.row_bytes(row);
        if (n < 0) atomicOr(&r.status->flags, kFlagRange);
        return n < 0 ? 0 : (n + kTrieSeg - 1) / kTrieSeg * kTrieSeg;
    }
};
// ... its offset filed, and the row's number with each of its segments
struct TrieStageOffsets {
    long long* off;
    int32_t* seg_row;
    long long cap;
    __device__ void operator()(long long i, long long o, long long len) const {
        off[i] = o;
        if (o + len > cap) return;   // (TrieStageFin raises the flag: the host grows the buffers and runs the call again)
        for (long long k = 0; k < len / kTrieSeg; ++k) seg_row[o / kTrieSeg + k] = int32_t(i);
    }
};
// the staging entries the batch needs (rows may share strings: then more than the chars tensor has bytes) -> stage_need, and the flag when
// the buffer at hand is smaller: the host grows it and runs the call again
struct TrieStageFin {
    RunStatus* status;
    long long cap;
    __device__ void operator()(long long total) const {
        status->stage_need = total > INT32_MAX ? INT32_MAX : int32_t(total);
        if (total > cap) atomicOr(&status->flags, kFlagStageOverflow);
    }
};
// A lane per segment: the chain that starts at the segment's first byte.
static __global__ __launch_bounds__(kTileThreads) void trie_segments_kernel(TrieRows r, const long long* stage_off, const int32_t* seg_row, int32_t* stage,
                                                                            unsigned long long* seg_bits, int32_t* seg_exit) {
    __shared__ __attribute__((aligned(16))) uint8_t win_lds[kTileThreads][16];
    if (r.status->flags & (kFlagRange | kFlagStageOverflow)) return;
    const long long g = (long long)blockIdx.x * kTileThreads + threadIdx.x;
    if (g >= r.status->stage_need / kTrieSeg) return;
    const long long row = seg_row[g];
    const long long so = stage_off[row];
    const long long cb = r.ragged_begins[row];
    unsigned long long bits = 0;
    int32_t exit = 0;
    if (r.ragged_ends[row] - cb == 1) {
        const long long b = r.begins[cb];
        const int n = int(r.ends[cb] - b);
        const int base = int(g - so / kTrieSeg) * kTrieSeg;
        const int until = base + kTrieSeg;
        TrieWindow tw{win_lds[threadIdx.x]};
        int32_t* dst = stage + so;
        exit = r.chain(b, n, base, tw,
                       [&](int at, int tok) {
                           dst[at] = tok;
                           bits |= 1ull << (at - base);
                       },
                       [&](int at) { return at < until; });
    }
    seg_bits[g] = bits;
    seg_exit[g] = exit;
}
// A lane per row: the true chain through the row's segments (rows of one string), or the whole walk (rows of several); the count filed.
// (`spread`: only every spread-th lane has a row -- the lanes wait for one table read after the other, and 65 536 rows on every lane are one
// wave per SIMD with nothing to switch to)
static __global__ __launch_bounds__(kTileThreads) void trie_rows_kernel(long long n_rows, TrieRows r, const long long* stage_off, int32_t* stage,
                                                                        unsigned long long* seg_bits, const int32_t* seg_exit, int32_t* lens, int spread) {
    __shared__ __attribute__((aligned(16))) uint8_t win_lds[kTileThreads][16];
    if (r.status->flags & (kFlagRange | kFlagStageOverflow)) return;
    const long long t0 = (long long)blockIdx.x * kTileThreads + threadIdx.x;
    if (t0 % spread != 0) return;
    const long long row = t0 / spread;
    if (row >= n_rows) return;
    const long long so = stage_off[row];
    const long long g0 = so / kTrieSeg;
    int32_t* dst = stage + so;
    const long long cb = r.ragged_begins[row], ce = r.ragged_ends[row];
    TrieWindow tw{win_lds[threadIdx.x]};
    int32_t count = 0;
    if (ce - cb == 1) {
        const long long b = r.begins[cb];
        const int n = int(r.ends[cb] - b);
        const int n_seg = (n + kTrieSeg - 1) / kTrieSeg;
        int t = 0;   // where the true chain enters the segment at hand
        for (int k = 0; k < n_seg; ++k) {
            const int base = k * kTrieSeg;
            if (t >= base + kTrieSeg) {   // a token that covers the whole segment
                seg_bits[g0 + k] = 0;
                continue;
            }
            const unsigned long long guess = seg_bits[g0 + k];
            int32_t exit = seg_exit[g0 + k];
            unsigned long long keep;
            if ((guess >> (t - base)) & 1ull) {
                keep = guess & (~0ull << (t - base));
            } else {
                keep = 0;
                int met = -1;   // the guess's token start the walk stepped on
                const int32_t e = r.chain(b, n, t, tw,
                                          [&](int at, int tok) {
                                              dst[at] = tok;
                                              keep |= 1ull << (at - base);
                                          },
                                          [&](int at) {
                                              if (at >= base + kTrieSeg) return false;
                                              if ((guess >> (at - base)) & 1ull) {
                                                  met = at;
                                                  return false;
                                              }
                                              return true;
                                          });
                if (met >= 0) keep |= guess & (~0ull << (met - base));
                else exit = e;
            }
            seg_bits[g0 + k] = keep;
            count += __popcll(keep);
            if (exit < 0) {  // the reference spins here forever (src/trie_tokenizer.cpp:72-75)
                atomicOr(&r.status->flags, kFlagItemsOverflow);
                break;
            }
            t = exit;
        }
    } else if (cb >= 0 && ce >= cb && ce <= r.n_strings) {
        // several strings: every string's chain from its first byte, positions counted through the row's strings
        int at0 = 0;   // bytes of the strings before
        unsigned long long bits = 0;
        long long seg = 0;
        bool broken = false;
        for (long long col = cb; col < ce && !broken; ++col) {
            const long long b = r.begins[col];
            const int n = int(r.ends[col] - b);
            const int32_t e = r.chain(b, n, 0, tw,
                                      [&](int at, int tok) {
                                          const int p = at0 + at;
                                          if (p / kTrieSeg != seg) {
                                              seg_bits[g0 + seg] = bits;
                                              for (long long q = seg + 1; q < p / kTrieSeg; ++q) seg_bits[g0 + q] = 0;
                                              seg = p / kTrieSeg;
                                              bits = 0;
                                          }
                                          dst[p] = tok;
                                          bits |= 1ull << (p % kTrieSeg);
                                          ++count;
                                      },
                                      [&](int) { return true; });
            broken = e < 0;
            at0 += n;
        }
        const long long n_seg = (at0 + kTrieSeg - 1) / kTrieSeg;
        if (seg < n_seg) seg_bits[g0 + seg] = bits;
        for (long long q = seg + 1; q < n_seg; ++q) seg_bits[g0 + q] = 0;
        if (broken) atomicOr(&r.status->flags, kFlagItemsOverflow);
    }
    lens[row] = count;
}
struct FiledLen {
    const int32_t* lens;
    __device__ long long operator()(long long i) const { return lens[i]; }
};
struct RowOffsets {
    int32_t* out_begins;
    int32_t* out_ends;
    long long base;
    __device__ void operator()(long long i, long long off, long long len) const {
        out_begins[i] = int32_t(base + off);
        out_ends[i] = int32_t(base + off + len);
    }
};
// a wave per row: the kept entries of the row's stretch, segment by segment, to their places in the output
struct TrieGather {
    const long long* stage_off;
    const int32_t* stage;
    const unsigned long long* seg_bits;
    const int32_t* lens;
    const int32_t* out_begins;
    int32_t* out_ids;
    const RunStatus* status;
    long long n_rows;
    __device__ void operator()(long long row) const {
        const long long so = stage_off[row];
        const long long end = row + 1 < n_rows ? stage_off[row + 1] : status->stage_need;
        int32_t* dst = out_ids + out_begins[row];
        const int l = lane_id();
        const long long g0 = so / kTrieSeg, n_seg = (end - so) / kTrieSeg;
        int done = 0;
        // the masks of 64 segments at a time, a lane each; then eight segments' entries in flight at once
        for (long long s0 = 0; s0 < n_seg; s0 += kWave) {
            const unsigned long long mine = s0 + l < n_seg ? seg_bits[g0 + s0 + l] : 0ull;
            const int here = int(n_seg - s0 < kWave ? n_seg - s0 : kWave);
            for (int k0 = 0; k0 < here; k0 += 8) {
                int32_t v[8];
                unsigned long long bits[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    bits[k] = (unsigned long long)(uint32_t)wave_readlane(int(uint32_t(mine)), (k0 + k) & (kWave - 1)) |
                              ((unsigned long long)(uint32_t)wave_readlane(int(uint32_t(mine >> 32)), (k0 + k) & (kWave - 1)) << 32);
                    if (k0 + k >= here) bits[k] = 0;
                    v[k] = ((bits[k] >> l) & 1ull) ? stage[(g0 + s0 + k0 + k) * kTrieSeg + l] : 0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((bits[k] >> l) & 1ull) dst[done + __popcll(bits[k] & ((1ull << l) - 1))] = v[k];
                    done += __popcll(bits[k]);
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------- UTF8Validate
// src/utf8_validate.cpp:18-143 walked symbol by symbol instead of byte by byte: a lead byte promises `need`
// continuation bytes; when one is missing (or the string ends) the symbol is replaced once and the offending byte
// is read again as a lead (:93-104, :134-137); a complete but overlong symbol is replaced once per byte (:111-121).
// (Until round 5 a lane walked its string that way, three times over -- count, count again, write --, four strings to a thread: 2.1 ms for
// a config-2 batch.)
// ---- that walk as arithmetic on bit masks, a wave per string, 64 bytes per step (round 5).
// The walk only ever steps over continuation bytes (10xxxxxx) -- behind a lead, as many as the lead asks for and no more -- so every
// other byte starts a symbol wherever the walk stood before, and what becomes of a byte is decided by the three bytes on either side
// of it.  With C = the continuation bytes of a 64-byte window and L1 / L2 / L3 = the leads that ask for one / two / three of them
// (bit i = byte i; bytes behind the string's end are in no mask):
//   c_k = "the k bytes behind me are continuation bytes" = c_{k-1} & (C >> k), the next window's first three bits shifted in;
//   complete = L1 & c1 | L2 & c2 | L3 & c3;  overlong (a complete symbol below its length's smallest code point, :111-121) is a
//   property of the lead and the byte behind it: C0 / C1, E0 + a byte below A0, F0 + a byte below 90;
//   a valid lead (complete, not overlong) and its continuation bytes are COPIED; an overlong one brings a replacement per byte and
//   its continuation bytes nothing; an incomplete lead ONE replacement, and the continuation bytes it did find (c1, c2) nothing;
//   a continuation byte none of them claims, and a byte F8..FF, one replacement each.
// The masks live in scalar registers; a lane's share is its byte, two compares and -- in the write pass -- the counts below it.
struct Utf8Window {
    uint64_t copy, bad1, o1, o2, o3;   // bytes kept | positions worth one replacement | overlong leads of 1 / 2 / 3 continuation bytes
};
// f(w0, c, in, m): window [w0, w0 + 64) of the string, the lane's byte c (in: inside the string), the window's masks
template <class F>
__device__ __forceinline__ void utf8_windows(const uint8_t* s, int len, F&& f) {
    const int l = lane_id();
    bool in = l < len;
    uint32_t c = in ? s[l] : 0u;
    uint64_t carry_keep = 0, carry_drop = 0;
    for (int w0 = 0; w0 < len; w0 += kWave) {
        const int ni = w0 + kWave + l;
        const bool in_n = ni < len;
        const uint32_t c_n = in_n ? s[ni] : 0u;   // the next window's bytes: on their way while this one is worked out
        const uint64_t C = __ballot(in && (c & 0xC0u) == 0x80u), Cn = __ballot(in_n && (c_n & 0xC0u) == 0x80u);
        const uint64_t A = __ballot(in && c < 0x80u);
        Utf8Window m;
        if (wave_uniform(int(A == __ballot(in)))) {   // nothing but ASCII (and nothing carried into it: a carry needs a continuation byte here)
            m = Utf8Window{A, 0, 0, 0, 0};
            carry_keep = carry_drop = 0;
        } else {
            const bool is1 = in && (c >> 5) == 0x6u, is2 = in && (c >> 4) == 0xEu, is3 = in && (c >> 3) == 0x1Eu;
            const uint64_t L1 = __ballot(is1), L2 = __ballot(is2), L3 = __ballot(is3), X = __ballot(in && c >= 0xF8u);
            const uint32_t next0 = uint32_t(wave_readlane(int(c_n), 0));   // (the byte behind lane 63's: the next window's first, 0 behind the end)
            uint32_t n1 = uint32_t(__shfl_down(int(c), 1));
            if (l == kWave - 1) n1 = next0;
            const bool over = (is1 && (c & 0x1Eu) == 0u) || (is2 && c == 0xE0u && n1 < 0xA0u) || (is3 && c == 0xF0u && n1 < 0x90u);
            const uint64_t OV = __ballot(over);
            const uint64_t c1 = (C >> 1) | (Cn << 63), c2 = c1 & ((C >> 2) | (Cn << 62)), c3 = c2 & ((C >> 3) | (Cn << 61));
            const uint64_t complete = (L1 & c1) | (L2 & c2) | (L3 & c3);
            const uint64_t I = (L1 | L2 | L3) & ~complete, O = complete & OV, V = complete & ~OV;
            const uint64_t V23 = V & (L2 | L3), V3 = V & L3, O23 = O & (L2 | L3), O3 = O & L3, I1 = I & c1, I2 = I & c2;
            const uint64_t keep = (V << 1) | (V23 << 2) | (V3 << 3) | carry_keep;
            const uint64_t drop = (O << 1) | (O23 << 2) | (O3 << 3) | (I1 << 1) | (I2 << 2) | carry_drop;
            carry_keep = (V >> 63) | (V23 >> 62) | (V3 >> 61);
            carry_drop = (O >> 63) | (O23 >> 62) | (O3 >> 61) | (I1 >> 63) | (I2 >> 62);
            m.copy = A | V | keep;
            m.bad1 = X | I | (C & ~keep & ~drop);
            m.o1 = O & L1;
            m.o2 = O & L2;
            m.o3 = O3;
        }
        f(w0, c, in, m);
        c = c_n;
        in = in_n;
    }
}
__device__ __forceinline__ int utf8_window_bytes(const Utf8Window& m, int replace) {
    return __popcll(m.copy) + (replace ? 3 * (__popcll(m.bad1) + 2 * __popcll(m.o1) + 3 * __popcll(m.o2) + 4 * __popcll(m.o3)) : 0);
}
struct Utf8WaveCount {   // each_wave_kernel: the string's length after validation
    const int32_t* begins;
    const int32_t* ends;
    const uint8_t* chars;
    long long n_chars;
    int replace;
    int32_t* lens;
    __device__ void operator()(long long i) const {
        const long long b = begins[i], e = ends[i];
        int total = 0;
        if (!(b < 0 || e < b || e > n_chars))   // (else: flagged by check_strings_kernel)
            utf8_windows(chars + b, int(e - b), [&](int, uint32_t, bool, const Utf8Window& m) { total += utf8_window_bytes(m, replace); });
        if (lane_id() == 0) lens[i] = total;
    }
};
struct Utf8WaveWrite {   // each_wave_kernel, behind the scan: the string's bytes to out_chars + out_begins[i]
    const int32_t* begins;
    const int32_t* ends;
    const uint8_t* chars;
    const int32_t* out_begins;
    uint8_t* out_chars;
    int replace;
    __device__ void operator()(long long i) const {
        uint8_t* dst = out_chars + out_begins[i];
        const int rep = replace;
        utf8_windows(chars + begins[i], ends[i] - begins[i], [&](int, uint32_t c, bool, const Utf8Window& m) {
            const int l = lane_id();
            int at = rank_below(m.copy);
            int times = 0;
            if (rep) {
                at += 3 * (rank_below(m.bad1) + 2 * rank_below(m.o1) + 3 * rank_below(m.o2) + 4 * rank_below(m.o3));
                times = int((m.bad1 >> l) & 1ull) + 2 * int((m.o1 >> l) & 1ull) + 3 * int((m.o2 >> l) & 1ull) + 4 * int((m.o3 >> l) & 1ull);
            }
            if ((m.copy >> l) & 1ull) dst[at] = uint8_t(c);
            for (int t = 0; t < times; ++t) {   // U+FFFD
                dst[at + 3 * t] = 0xEF;
                dst[at + 3 * t + 1] = 0xBF;
                dst[at + 3 * t + 2] = 0xBD;
            }
            dst += utf8_window_bytes(m, rep);
        });
    }
};

// ------------------------------------------------------------------------------- Truncate
// src/truncate.cpp:37-150, one lane per row.  side: 0 right / 1 left; mode: 0 only_first, 1 only_second,
// 2 longest_first.  b1/e1 == nullptr -> single input (:42-60).
struct TruncateArgs {
    const int32_t *b0, *e0, *b1, *e1;
    int32_t *ob0, *oe0, *ob1, *oe1;
    long long n;
    int32_t max_length;
    int left, mode;
};

static __global__ __launch_bounds__(kBlockThreads) void truncate_kernel(TruncateArgs a) {
    const long long stride = (long long)gridDim.x * kBlockThreads;
    for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < a.n; i += stride) {
        int32_t fb = a.b0[i], fe = a.e0[i];
        if (!a.b1) {
            const int32_t len = fe - fb, t = len < a.max_length ? len : a.max_length;
            if (a.left) fb = fe - t; else fe = fb + t;
            a.ob0[i] = fb;
            a.oe0[i] = fe;
            continue;
        }
        int32_t sb = a.b1[i], se = a.e1[i];
        const int32_t fl = fe - fb, sl = se - sb, m = a.max_length;
        int32_t keep_f = fl, keep_s = sl;  // lengths kept
        if (fl + sl > m) {
            const int32_t half = m / 2, half_up = m / 2 + m % 2;
            if (a.mode == 0) keep_f = fl > m ? m : fl;
            else if (a.mode == 1) keep_s = sl > m ? m : sl;
            else if (fl >= half_up && sl <= half) keep_f = m - sl;
            else if (fl < half_up && sl > half) keep_s = m - fl;
            else {
                keep_f = half + (m % 2) * (fl >= sl);
                keep_s = half + (m % 2) * (fl < sl);
            }
        }
        if (a.left) { fb = fe - keep_f; sb = se - keep_s; } else { fe = fb + keep_f; se = sb + keep_s; }
        a.ob0[i] = fb; a.oe0[i] = fe;
        a.ob1[i] = sb; a.oe1[i] = se;
    }
}

// ------------------------------------------------------------------------------- CombineSegments
// src/combine_segments.cpp:36-134: row i of the result is the concatenation of row i of every input
// (a one-row input is broadcast, :110-116) with a parallel tensor naming the source input.  The
// reference's running `flat_out_size` becomes a device scan of the row totals; then one wave copies
// a row, segment after segment, lanes on consecutive elements (coalesced both sides).
constexpr int kMaxSegments = 16;
struct CombineDev {
    const int32_t* begins[kMaxSegments];
    const int32_t* ends[kMaxSegments];
    const int32_t* data[kMaxSegments];
    int32_t n_rows[kMaxSegments];
    int32_t n_data[kMaxSegments];
    int32_t ids[kMaxSegments];
    int n_segs;
};

struct CombineLen {
    CombineDev d;
    RunStatus* status;
    __device__ long long operator()(long long i) const {
        long long tot = 0;
        for (int j = 0; j < d.n_segs; ++j) {
            const long long r = d.n_rows[j] == 1 ? 0 : i;
            const int32_t b = d.begins[j][r], e = d.ends[j][r];
            if (b < 0 || e < b || e > d.n_data[j]) {
                atomicOr(&status->flags, kFlagRange);
                continue;
            }
            tot += e - b;
        }
        return tot;
    }
};
struct CombineApply {
    int32_t* out_begins;
    int32_t* out_ends;
    __device__ void operator()(long long i, long long off, long long len) const {
        out_begins[i] = int32_t(off);
        out_ends[i] = int32_t(off + len);
    }
};

static __global__ __launch_bounds__(kBlockThreads) void combine_copy_kernel(CombineDev d, long long n_rows,
                                                                            const int32_t* out_begins, int32_t* out_data,
                                                                            int32_t* out_ids, const RunStatus* status) {
    if (status->flags & (kFlagRange | kFlagOutCapacity)) return;
    const int l = lane_id();
    const long long stride = (long long)gridDim.x * kWavesPerBlock;
    for (long long i = (long long)blockIdx.x * kWavesPerBlock + wave_in_block(); i < n_rows; i += stride) {
        long long off = out_begins[i];
        for (int j = 0; j < d.n_segs; ++j) {
            const long long r = d.n_rows[j] == 1 ? 0 : i;
            const int32_t b = d.begins[j][r], len = d.ends[j][r] - b;
            const int32_t id = d.ids[j];
            const int32_t* src = d.data[j] + b;
            for (int32_t k = l; k < len; k += kWave) {
                out_data[off + k] = src[k];
                out_ids[off + k] = id;
            }
            off += len;
        }
    }
}

// ------------------------------------------------------------------------------- Truncate + CombineSegments + RaggedToDense
// The tail of every encode pipeline (tokenizer_pipeline.py TruncationStep -> CombineSegmentsStep -> PaddingStep) in one
// kernel: a wave owns a row, cuts the one or two truncated segments (src/truncate.cpp:37-150), lays the segments out
// back to back (src/combine_segments.cpp:36-134) and writes the padded rows of input_ids / attention_mask /
// token_type_ids directly (src/ragged_to_dense.cpp:70-174) -- the combined ragged tensor never exists.
struct TailDev {
    CombineDev c;
    int trunc_a, trunc_b;  // indices of the truncated segments (-1: none); trunc_b only for pair truncation
    int32_t max_length;
    int left, mode;        // as truncate_kernel
    int32_t target, pad_value, type_pad;
    int pad_right;
};

// Kept range [b, e) of segment j in row i after truncation; false when the offsets leave the data tensor.
__device__ __forceinline__ bool tail_segment(const TailDev& t, long long i, int j, int32_t& b, int32_t& e) {
    const CombineDev& c = t.c;
    const long long r = c.n_rows[j] == 1 ? 0 : i;
    b = c.begins[j][r];
    e = c.ends[j][r];
    if (b < 0 || e < b || e > c.n_data[j]) return false;
    if (j != t.trunc_a && j != t.trunc_b) return true;
    int32_t keep = e - b;
    const int32_t m = t.max_length;
    if (t.trunc_b < 0) {
        keep = keep < m ? keep : m;  // single input (:42-60)
    } else {
        const int o = j == t.trunc_a ? t.trunc_b : t.trunc_a;
        const long long ro = c.n_rows[o] == 1 ? 0 : i;
        const int32_t other = c.ends[o][ro] - c.begins[o][ro];
        const int32_t fl = j == t.trunc_a ? keep : other, sl = j == t.trunc_a ? other : keep;
        if (fl + sl > m) {
            const int32_t half = m / 2, half_up = m / 2 + m % 2;
            int32_t keep_f = fl, keep_s = sl;
            if (t.mode == 0) keep_f = fl > m ? m : fl;
            else if (t.mode == 1) keep_s = sl > m ? m : sl;
            else if (fl >= half_up && sl <= half) keep_f = m - sl;
            else if (fl < half_up && sl > half) keep_s = m - fl;
            else {
                keep_f = half + (m % 2) * (fl >= sl);
                keep_s = half + (m % 2) * (fl < sl);
            }
            keep = j == t.trunc_a ? keep_f : keep_s;
        }
    }
    if (t.left) b = e - keep; else e = b + keep;
    return true;
}

// Longest combined row (the PaddingStep's ReduceMax) -> status->n_out.
static __global__ __launch_bounds__(kBlockThreads) void tail_measure_kernel(TailDev t, long long n_rows, RunStatus* status) {
    const long long stride = (long long)gridDim.x * kBlockThreads;
    int longest = 0;
    for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n_rows; i += stride) {
        int tot = 0;
        for (int j = 0; j < t.c.n_segs; ++j) {
            int32_t b, e;
            if (!tail_segment(t, i, j, b, e)) atomicOr(&status->flags, kFlagRange);
            else tot += e - b;
        }
        longest = tot > longest ? tot : longest;
    }
    longest = wave_max(longest);
    if (lane_id() == 0 && longest) atomicMax(&status->n_out, longest);
}

static __global__ __launch_bounds__(kBlockThreads) void tail_dense_kernel(TailDev t, long long n_rows, int32_t* out_ids,
                                                                          uint8_t* out_mask, int32_t* out_types, RunStatus* status) {
    const int l = lane_id();
    const long long stride = (long long)gridDim.x * kWavesPerBlock;
    const int T = t.target;
    for (long long i = (long long)blockIdx.x * kWavesPerBlock + wave_in_block(); i < n_rows; i += stride) {
        int total = 0;
        bool ok = true;
        for (int j = 0; j < t.c.n_segs; ++j) {
            int32_t b, e;
            ok = tail_segment(t, i, j, b, e) && ok;
            total += ok ? e - b : 0;
        }
        if (!ok) {
            if (l == 0) atomicOr(&status->flags, kFlagRange);
            continue;
        }
        const int used = total < T ? total : T;                 // ragged_to_dense.cpp: a longer row is cut at the target
        const int first = t.pad_right ? 0 : T - used;            // where the row's elements start in the dense row
        int32_t* ids = out_ids + i * T;
        uint8_t* mask = out_mask ? out_mask + i * T : nullptr;
        int32_t* types = out_types ? out_types + i * T : nullptr;
        for (int p = l; p < T; p += kWave) {
            const bool in_row = p >= first && p < first + used;
            if (!in_row) {
                ids[p] = t.pad_value;
                if (mask) mask[p] = 0;
                if (types) types[p] = t.type_pad;
            }
        }
        int at = 0;
        for (int j = 0; j < t.c.n_segs && at < used; ++j) {
            int32_t b, e;
            tail_segment(t, i, j, b, e);
            int len = e - b;
            if (at + len > used) len = used - at;
            const int32_t* src = t.c.data[j] + b;
            const int32_t sid = t.c.ids[j];
            for (int k = l; k < len; k += kWave) {
                ids[first + at + k] = src[k];
                if (mask) mask[first + at + k] = 1;
                if (types) types[first + at + k] = sid;
            }
            at += len;
        }
    }
}

// ------------------------------------------------------------------------------- string tensor wire format
// Pack (the inverse of parse_packed_strings, src/utils.cpp:18-29): string i goes to bytes[off_i, off_i + len_i) and
// its end offset to header word 2 + i; off = exclusive scan of the lengths.
struct PackLen {
    const int32_t* begins;
    const int32_t* ends;
    long long n_chars;
    __device__ long long operator()(long long i) const {
        const long long b = begins[i], e = ends[i];
        return (b < 0 || e < b || e > n_chars) ? 0 : e - b;  // flagged by check_strings_kernel
    }
};
struct PackApply {   // the end offsets (the scan's apply pass); the bytes: PackCopy, a wave per string
    int32_t* header;  // [n, begin_0, end_0 .. end_{n-1}]
    __device__ void operator()(long long i, long long off, long long len) const { header[2 + i] = int32_t(off + len); }
};
struct __attribute__((packed, aligned(1))) CopyBytes16 { uint32_t d[4]; };
// len bytes from src to dst by the 64 lanes of a wave: 16 bytes per lane and step at any alignment, the last bytes one by one
__device__ __forceinline__ void wave_copy_bytes(const uint8_t* src, uint8_t* dst, long long len) {
    const int l = lane_id();
    long long k = 16ll * l;
    for (; k + 16 <= len; k += 16 * kWave) *reinterpret_cast<CopyBytes16*>(dst + k) = *reinterpret_cast<const CopyBytes16*>(src + k);
    if (k < len)   // (one lane: the first whose 16 bytes do not fit)
        for (long long t = k; t < len; ++t) dst[t] = src[t];
}
struct PackCopy {
    const int32_t* begins;
    const uint8_t* chars;
    const int32_t* header;
    uint8_t* bytes;
    __device__ void operator()(long long i) const {
        const long long off = header[1 + i], end = header[2 + i];   // (header[1] = 0: the first string's begin)
        wave_copy_bytes(chars + begins[i], bytes + off, end - off);
    }
};
static __global__ __launch_bounds__(kWave) void pack_header_kernel(int32_t* header, int32_t n) {
    if (threadIdx.x == 0) {
        header[0] = n;
        header[1] = 0;
    }
}

// ------------------------------------------------------------------------------- row-shard exchange
// Wire format of one rank's ragged ids for the RCCL all-gather (no reference counterpart).  Every rank sends the same
// number of bytes because RCCL has no all-gather-v:
//     i32 n_ids, i32 n_rows, 2 x i32 0 | i32 ends[max_rows] (the shard's own end offsets) | pad_ids ids of 2 or 4 bytes
// The receiver needs no scan and no counts exchange: rank r's ids go to the sum of the n_ids of the ranks before it, its
// rows to the sum of their n_rows (any contiguous partition of the rows: balanced by count or by bytes), and a row's
// global offsets are the id base plus its local ones.  One kernel packs, one kernel unpacks.
struct ShardGeom {
    long long n_rows;    // global rows
    long long max_rows;  // ends slots per wire
    long long pad_ids;   // id slots per wire
    long long stride;    // bytes per wire
    int world, id_bytes;
};

static __global__ __launch_bounds__(kBlockThreads) void shard_pack_kernel(const int32_t* begins, const int32_t* ends,
                                                                          const int32_t* ids, long long rows, long long n_ids,
                                                                          ShardGeom g, uint8_t* wire) {
    int32_t* hdr = reinterpret_cast<int32_t*>(wire);
    int32_t* w_ends = hdr + kShardHeaderBytes / 4;
    uint8_t* wid = wire + kShardHeaderBytes + g.max_rows * 4;
    const long long stride = (long long)gridDim.x * kBlockThreads, t0 = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
    if (t0 == 0) {
        hdr[0] = int32_t(n_ids);
        hdr[1] = int32_t(rows);
        hdr[2] = hdr[3] = 0;
    }
    const int32_t origin = rows > 0 ? begins[0] : 0;  // offsets travel relative to the shard's first id
    for (long long i = t0; i < g.max_rows; i += stride) w_ends[i] = i < rows ? ends[i] - origin : 0;
    const long long n = n_ids < g.pad_ids ? n_ids : g.pad_ids;  // a shard larger than the pad is cut; n_ids in the header says so
    if (g.id_bytes == 2) {
        uint16_t* w = reinterpret_cast<uint16_t*>(wid);
        for (long long i = t0; i < n; i += stride) w[i] = uint16_t(ids[origin + i]);
    } else {
        int32_t* w = reinterpret_cast<int32_t*>(wid);
        for (long long i = t0; i < n; i += stride) w[i] = ids[origin + i];
    }
}

// grid (chunks, world): block (x, r) handles rank r's wire -- rows r's offsets and a share of its ids; block (0, 0)
// also writes the verdict (ovtk_shard_result, include/ovtk_amd.h).
static __global__ __launch_bounds__(kBlockThreads) void shard_unpack_kernel(const uint8_t* wires, ShardGeom g, int32_t* out_begins,
                                                                            int32_t* out_ends, int32_t* out_ids, long long out_cap,
                                                                            ovtk_shard_result* res) {
    const long long r = blockIdx.y;
    long long off = 0, total = 0, biggest = 0, row0 = 0, rows_total = 0;
    bool bad_rows = false;
    for (int q = 0; q < g.world; ++q) {  // the world's headers: a few scalar loads per block
        const int32_t* h = reinterpret_cast<const int32_t*>(wires + q * g.stride);
        const long long c = h[0], nr = h[1];
        if (q < r) { off += c; row0 += nr; }
        total += c;
        rows_total += nr;
        biggest = c > biggest ? c : biggest;
        bad_rows = bad_rows || nr < 0 || nr > g.max_rows;
    }
    bad_rows = bad_rows || rows_total != g.n_rows;  // the shards are not a partition of the batch's rows
    const bool cut = biggest > g.pad_ids, too_big = total > INT32_MAX - 1, no_room = total > out_cap;
    if (blockIdx.x == 0 && r == 0 && threadIdx.x == 0) {
        res->n_ids = total;
        res->max_shard_ids = biggest;
        res->reserved = 0;
        res->status = bad_rows ? OVTK_E_ARG : too_big ? OVTK_E_UNSUPPORTED : cut ? OVTK_E_CAPACITY : no_room ? OVTK_E_RANGE : OVTK_OK;
    }
    if (bad_rows || cut || too_big || no_room) return;
    const uint8_t* wire = wires + r * g.stride;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(wire);
    const int32_t* w_ends = hdr + kShardHeaderBytes / 4;
    const long long cnt = hdr[0], rows = hdr[1];
    const long long stride = (long long)gridDim.x * kBlockThreads, t0 = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
    for (long long i = t0; i < rows; i += stride) {
        const long long e = w_ends[i], b = i ? w_ends[i - 1] : 0;
        out_begins[row0 + i] = int32_t(off + b);
        out_ends[row0 + i] = int32_t(off + e);
    }
    const uint8_t* wid = wire + kShardHeaderBytes + g.max_rows * 4;
    int32_t* dst = out_ids + off;
    if (g.id_bytes == 2) {
        const uint16_t* w = reinterpret_cast<const uint16_t*>(wid);
        for (long long i = t0; i < cnt; i += stride) dst[i] = w[i];
    } else {
        const int32_t* w = reinterpret_cast<const int32_t*>(wid);
        for (long long i = t0; i < cnt; i += stride) dst[i] = w[i];
    }
}

}  // namespace ovtk
