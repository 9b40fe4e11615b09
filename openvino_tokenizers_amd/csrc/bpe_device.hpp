// bpe_device.hpp -- device side of BPETokenizerImpl::tokenize_into (src/bpe_tokenizer.cpp:196-339).
//
// The reference pops merge candidates from a std::priority_queue ordered by (rank, seq), seq being a
// creation counter (initial pair k has seq k; the two pairs created by the j-th merge share seq
// n-1+j).  Because stale entries are skipped, "pop" always yields the LIVE adjacent pair with the
// smallest (rank, seq) -- so as long as that minimum is unique the queue can be replaced by a
// min-scan over the live pairs, which is what the two fast paths do:
//   path F  one lane per piece (<= kFastSyms symbols): symbols + pair keys of the 64 pieces of a
//           batch live in the wave's LDS arrays in [symbol][lane] order (bank = lane: conflict-free
//           when the lanes are at the same symbol index), each lane scans / compacts its own column;
//   path W  one wave per piece (up to kChunk symbols): lanes scan the piece's pairs in parallel,
//           wave-min picks the merge, lanes shift the tail cooperatively.
// A non-unique minimum can only be the two pairs pushed by one merge with equal rank;
// its pop order is decided by libstdc++'s heap layout, so such a piece (and any piece too long
// for LDS) is replayed by path X: bpe_exact_piece(), a step-for-step emulation of
// std::push_heap / std::pop_heap on the same (rank, new_id, a, b, seq) entries, stale ones included.
//
// Pair key (u64): rank:22 | seq:10 | new_id:21, all ones = "not a merge"; path F keeps its upper half (rank | seq,
// exactly 32 bits: the order) and the merged id in two u32 LDS arrays.  One lookup of the merge table yields rank
// and merged id together, so a merge step has ONE dependent global round trip (the two lookups for the new
// neighbour pairs, issued together).  seq < 1024 because a piece handled in LDS has at most 512 symbols and seq < 2n.
#pragma once

#include "device_common.hpp"
#include "tables.hpp"

namespace ovtk {

constexpr uint64_t kNoKey = ~0ull;
constexpr int kSeqBits = 10;
constexpr int kIdBits = kMaxVocabBits;                   // 21
constexpr int kFastSyms = 16;                            // pieces with more symbols than this use path L or W
constexpr int kLongSyms = 32;                            // path L: lane per piece again, 32 pieces of <= 32 symbols at a time
constexpr int kChunkSyms = 512;                          // path W limit (symbols incl. end_suffix)
constexpr uint32_t kIdMask = (1u << kIdBits) - 1;

// One lookup of the merge table = both candidate slots, two independent 16-byte loads.
struct MergeFetch { MergeBucket a, b; };
__device__ __forceinline__ MergeFetch merge_fetch(const BpeDev& T, uint64_t key) {
    return MergeFetch{T.merges[merge_h1(key, T.bucket_shift)], T.merges[merge_h2(key, T.bucket_shift)]};
}
__device__ __forceinline__ uint64_t make_pair_key(const MergeSlot& s, uint32_t seq) {
    return ((s.kr & kNoRank) << (kSeqBits + kIdBits)) | (uint64_t(seq) << kIdBits) | (s.nid & kIdMask);
}
__device__ __forceinline__ uint64_t merge_resolve(const MergeFetch& f, uint64_t key, uint32_t seq) {
    // an empty slot holds all ones: its upper 42 bits never equal a key (ids < 2^21 - 1)
    uint64_t r = kNoKey;
    if ((f.a.s[0].kr >> kMaxRankBits) == key) r = make_pair_key(f.a.s[0], seq);
    if ((f.b.s[0].kr >> kMaxRankBits) == key) r = make_pair_key(f.b.s[0], seq);
    return r;
}
__device__ __forceinline__ uint64_t pair_key(const BpeDev& T, uint32_t l, uint32_t r, uint32_t seq) {
    const uint64_t key = merge_key(l, r);
    return merge_resolve(merge_fetch(T, key), key, seq);
}
// Rank only (path X).
__device__ __forceinline__ uint32_t merge_rank(const BpeDev& T, uint32_t l, uint32_t r) {
    const uint64_t k = pair_key(T, l, r, 0);
    return k == kNoKey ? kNoRank : uint32_t(k >> (kSeqBits + kIdBits));
}

// The edge (node, byte) -> e; false when the node has no such child.
__device__ __forceinline__ bool trie_step(const TrieDev& t, int node, uint32_t byte, TrieEdge& e) {
    const uint32_t key = (uint32_t(node) << 8) | byte;
    uint32_t p = (hash_u32(key) >> t.edge_shift) & t.edge_mask;
    for (;;) {
        e = t.edges[p];
        if (e.key == key) return true;
        if (e.key == kNoEdge) return false;
        p = (p + 1) & t.edge_mask;
    }
}

// Longest-match walk (src/utils.cpp:517-538) from text position idx; `root` may point to an LDS copy.
template <class GetByte>
__device__ __forceinline__ int trie_longest(const TrieDev& t, const I2* root, GetByte&& getb, int n, int& idx) {
    const I2 r = root[getb(idx)];
    if (r.y < 0) return -1;
    int best = r.x, best_end = idx + 1, i = idx + 1;
    int cur = r.y & ~kLeafBit;
    bool leaf = (r.y & kLeafBit) != 0;
    while (!leaf && i < n) {
        TrieEdge e;
        if (!trie_step(t, cur, getb(i), e)) break;
        cur = e.child;
        ++i;
        if (e.value != -1) { best = e.value; best_end = i; }
        leaf = e.has_kids == 0;
    }
    if (best == -1) return -1;
    idx = best_end;
    return best;
}

// Initial symbols of one piece (bpe_tokenizer.cpp:230-257): greedy longest trie match, else the
// byte-fallback token, else unk, else the byte is dropped.  Returns the symbol count.
template <class GetByte, class Put>
__device__ __forceinline__ int bpe_symbolize(const BpeDev& T, const I2* root, GetByte&& getb, int n, Put&& put) {
    int idx = 0, cnt = 0;
    while (idx < n) {
        const int tok = trie_longest(T.trie, root, getb, n, idx);
        if (tok != -1) {
            put(cnt++, tok);
            continue;
        }
        const int fb = T.byte_fallback_id[getb(idx)];
        if (fb != -1) put(cnt++, fb);
        else if (T.unk_id != -1) put(cnt++, T.unk_id);  // fuse_unk never changes this
        ++idx;
    }
    return cnt;
}

// Path F.  id / key / nid: the wave's [kFastSyms][64] u32 LDS arrays; this lane owns column lane_id().  n symbols are
// in place.  Symbols never move: a merge writes the new id over its left operand and clears the right operand's bit
// in the lane's `live` mask; key[k] = rank << 10 | seq and nid[k] = merged id describe the pair (k, next live symbol
// after k), key 0xFFFFFFFF = "not a merge" (dead / absent positions too).  The min-scan reads all kFastSyms-1 keys with
// independent, fully unrolled LDS loads.  Returns the final symbol count (symbols compacted to the front of the
// column), or -1 when the minimum was not unique (replay on path X).
constexpr uint32_t kNoKey32 = 0xFFFFFFFFu;
__device__ __forceinline__ void split_pair_key(uint64_t k64, uint32_t& key, uint32_t& nid) {
    key = k64 == kNoKey ? kNoKey32 : uint32_t(k64 >> kIdBits);
    nid = uint32_t(k64) & kIdMask;
}
// IdT: uint32_t, or uint16_t when every id of the vocabulary is below 65536 (8 instead of 12 bytes of LDS per symbol:
// merge_kernel is occupancy-bound by its LDS).
// NSYM x LANES: the arrays' geometry -- kFastSyms x 64 for the pieces of a batch, 32 x 32 for its long pieces (17..32
// symbols, half a wave at a time: the same LDS bytes).  Only lanes < LANES may call it; no wave collectives inside.
template <typename IdT, int NSYM = kFastSyms, int LANES = kWave>
__device__ __forceinline__ int bpe_merge_lane(const BpeDev& T, IdT* id, uint32_t* key, IdT* nid, int n) {
    static_assert(NSYM <= 32, "the live mask is 32 bits");
    const int l = lane_id();
#define OVTK_AT(k) ((k) * LANES + l)
    // initial pair keys: the lookups of a group of 4 are issued together
#pragma unroll
    for (int k0 = 0; k0 < NSYM; k0 += 4) {
        uint64_t mk[4] = {0, 0, 0, 0};
        MergeFetch f[4] = {};
        if (k0 + 1 < n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j;
                mk[j] = (k + 1 < n) ? merge_key(id[OVTK_AT(k)], id[OVTK_AT(k + 1)]) : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = merge_fetch(T, mk[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            if (k < NSYM - 1) {
                uint32_t kk = kNoKey32, nn = 0;
                if (k + 1 < n) split_pair_key(merge_resolve(f[j], mk[j], uint32_t(k)), kk, nn);
                key[OVTK_AT(k)] = kk;
                nid[OVTK_AT(k)] = IdT(nn);
            }
        }
    }
    uint32_t live = n >= 32 ? ~0u : ((1u << n) - 1u);
    uint32_t seq = n > 0 ? uint32_t(n - 1) : 0;
    bool dup = false;
#ifdef OVTK_PROBE
    if (NSYM == kFastSyms) {
        int mx = n;
        for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
        const int pw_ = (int(blockIdx.y) * int(gridDim.x) + int(blockIdx.x)) * kWavesPerBlock + wave_in_block();
        if (lane_id() == 0 && pw_ < 8192) g_ts[pw_][8] = (unsigned long long)mx;
    }
    int iters_ = 0;
#endif
    while (n >= 2) {
#ifdef OVTK_PROBE
        ++iters_;
#endif
        uint32_t v[NSYM - 1];
#pragma unroll
        for (int k = 0; k < NSYM - 1; ++k) v[k] = key[OVTK_AT(k)];
        uint32_t best = kNoKey32;
        int at = 0;
#pragma unroll
        for (int k = 0; k < NSYM - 1; ++k) {
            if (v[k] < best) { best = v[k]; at = k; dup = false; }
            else if (v[k] == best && best != kNoKey32) dup = true;
        }
        if (best == kNoKey32 || dup) break;
        const uint32_t merged = nid[OVTK_AT(at)];
        const uint32_t above = live & ~((2u << at) - 1u);          // live symbols right of `at`
        const int right = __ffs(above) - 1;                        // the right operand (exists: key[at] was a pair)
        live &= ~(1u << right);
        const uint32_t above2 = above & ~(1u << right);
        const uint32_t below = live & ((1u << at) - 1u);
        const int nxt = above2 ? __ffs(above2) - 1 : -1;           // new right neighbour
        const int prv = below ? 31 - __clz(below) : -1;            // left neighbour
        id[OVTK_AT(at)] = IdT(merged);
        --n;
        ++seq;
        // the two new neighbour pairs: both lookups in flight together
        const uint64_t kl = prv >= 0 ? merge_key(id[OVTK_AT(prv >= 0 ? prv : 0)], merged) : 0;
        const uint64_t kr = nxt >= 0 ? merge_key(merged, id[OVTK_AT(nxt >= 0 ? nxt : 0)]) : 0;
        const MergeFetch fl = merge_fetch(T, kl), fr = merge_fetch(T, kr);
        if (prv >= 0) {
            uint32_t kk, nn;
            split_pair_key(merge_resolve(fl, kl, seq), kk, nn);
            key[OVTK_AT(prv)] = kk;
            nid[OVTK_AT(prv)] = IdT(nn);
        }
        uint32_t kk = kNoKey32, nn = 0;
        if (nxt >= 0) split_pair_key(merge_resolve(fr, kr, seq), kk, nn);
        key[OVTK_AT(at)] = kk;
        nid[OVTK_AT(at)] = IdT(nn);
        if (right < NSYM - 1) key[OVTK_AT(right)] = kNoKey32;
    }
#ifdef OVTK_PROBE
    if (NSYM == kFastSyms) {
        int mx = iters_;
        for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
        const int pw_ = (int(blockIdx.y) * int(gridDim.x) + int(blockIdx.x)) * kWavesPerBlock + wave_in_block();
        if (lane_id() == 0 && pw_ < 8192) g_ts[pw_][9] = (unsigned long long)mx;
    }
#endif
    if (dup) return -1;
    // compact the surviving symbols to the front (ascending positions: reads never trail writes)
    int m = 0;
    for (uint32_t rest = live; rest; rest &= rest - 1) {
        const int k = __ffs(rest) - 1;
        const IdT t = id[OVTK_AT(k)];
        id[OVTK_AT(m++)] = t;
    }
#undef OVTK_AT
    return n;
}

// Path W.  All 64 lanes work on ONE piece of n symbols at id/key (LDS, contiguous).  Wave-uniform; returns the
// final count or -1 for a non-unique minimum.
__device__ __forceinline__ int bpe_merge_wave(const BpeDev& T, uint32_t* id, uint64_t* key, int n) {
    const int l = lane_id();
    for (int k = l; k + 1 < n; k += kWave) key[k] = pair_key(T, id[k], id[k + 1], uint32_t(k));
    wave_sync();
    uint32_t seq = n > 0 ? uint32_t(n - 1) : 0;
    while (n >= 2) {
        // minimum key and its position; a second pair with the same key shows up in `hits`
        uint64_t local_best = kNoKey;
        int local_at = 0, hits = 0;
        for (int k = l; k + 1 < n; k += kWave) {
            const uint64_t v = key[k];
            if (v < local_best) { local_best = v; local_at = k; hits = 1; }
            else if (v == local_best && v != kNoKey) ++hits;
        }
        const uint64_t bkey = wave_min_u64(local_best);
        if (bkey == kNoKey) break;
        const int same = wave_sum(local_best == bkey ? hits : 0);
        if (same > 1) return -1;
        const unsigned long long owner = __ballot(local_best == bkey);
        const int at = __shfl(local_at, __ffsll(owner) - 1);
        const uint32_t nid = uint32_t(bkey) & kIdMask;
        // cooperative shift-left of (at+1, n)
        for (int base = at + 1; base + 1 < n; base += kWave) {
            const int k = base + l;
            uint32_t a = 0;
            uint64_t b = 0;
            if (k + 1 < n) { a = id[k + 1]; b = key[k + 1]; }
            wave_sync();
            if (k + 1 < n) { id[k] = a; key[k] = b; }
            wave_sync();
        }
        --n;
        ++seq;
        if (l == 0) {
            id[at] = nid;
            if (at > 0) key[at - 1] = pair_key(T, id[at - 1], nid, seq);
        } else if (l == 1) {
            if (at + 1 < n) key[at] = pair_key(T, nid, id[at + 1], seq);
        }
        wave_sync();
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// Path X: exact replay with libstdc++'s binary heap (bits/stl_heap.h __push_heap / __adjust_heap,
// as used by std::priority_queue::emplace / pop) and the reference's symbol list.  One lane per
// piece, all state in global scratch.  `less(x, y)` is CompareRank(x, y): x has the LARGER
// (rank, seq) (bpe_tokenizer.cpp:166-172).
// ---------------------------------------------------------------------------------------------
struct HeapEntry { int32_t rank, seq, a, b; };  // new_id is a function of rank (T.new_id)

__device__ __forceinline__ bool heap_less(const HeapEntry& x, const HeapEntry& y) {
    return x.rank != y.rank ? x.rank > y.rank : x.seq > y.seq;
}
__device__ __forceinline__ void heap_push(HeapEntry* h, int& size, HeapEntry v) {
    int hole = size++;
    int parent = (hole - 1) / 2;
    while (hole > 0 && heap_less(h[parent], v)) {
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h[hole] = v;
}
__device__ __forceinline__ HeapEntry heap_pop(HeapEntry* h, int& size) {
    const HeapEntry top = h[0];
    if (size > 1) {
        const HeapEntry v = h[size - 1];  // pop_heap: value = *(last-1); *(last-1) = *first; adjust(0, len-1, value)
        const int len = size - 1;
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (heap_less(h[child], h[child - 1])) --child;
            h[hole] = h[child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            h[hole] = h[child - 1];
            hole = child - 1;
        }
        int parent = (hole - 1) / 2;
        while (hole > 0 && heap_less(h[parent], v)) {
            h[hole] = h[parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        h[hole] = v;
    }
    --size;
    return top;
}

// Bytes of scratch bpe_exact_piece needs for a text of n bytes (suffix included).
__host__ __device__ inline uint32_t bpe_exact_scratch_bytes(uint32_t n) {
    return (3u * 2u * n * 4u) + (3u * n + 1u) * uint32_t(sizeof(HeapEntry)) + 64u;
}

// Tokenizes one piece exactly as the reference does; writes the ids to out[0..) and returns their
// number.  scratch: bpe_exact_scratch_bytes(n) bytes, 16-byte aligned.
template <class GetByte>
__device__ inline int bpe_exact_piece(const BpeDev& T, GetByte&& getb, int n, void* scratch, int32_t* out) {
    int32_t* sid = static_cast<int32_t*>(scratch);  // [2n] id
    int32_t* sprev = sid + 2 * n;                   // [2n]
    int32_t* snext = sprev + 2 * n;                 // [2n]  (-2 = dead)
    HeapEntry* heap = reinterpret_cast<HeapEntry*>(
        (reinterpret_cast<uintptr_t>(snext + 2 * n) + 15) & ~uintptr_t(15));
    int nsym = bpe_symbolize(T, T.trie.root, getb, n, [&](int k, int tok) {
        sid[k] = tok;
        sprev[k] = k - 1;
        snext[k] = -1;
        if (k > 0) snext[k - 1] = k;
    });
    int live = nsym, hsize = 0, seq = 0, head = nsym ? 0 : -1;
    auto try_push = [&](int a, int b) {
        const uint32_t rank = merge_rank(T, uint32_t(sid[a]), uint32_t(sid[b]));
        if (rank != kNoRank) heap_push(heap, hsize, HeapEntry{int32_t(rank), seq, a, b});
    };
    for (int a = head; a != -1 && snext[a] != -1; a = snext[a]) {
        try_push(a, snext[a]);
        ++seq;
    }
    while (hsize > 0 && live >= 2) {
        const HeapEntry e = heap_pop(heap, hsize);
        if (snext[e.a] == -2 || snext[e.b] == -2 || snext[e.a] != e.b) continue;  // stale
        const int pv = sprev[e.a], nx = snext[e.b], m = nsym++;
        sid[m] = T.new_id[e.rank];
        sprev[m] = pv;
        snext[m] = nx;
        snext[e.a] = -2;
        snext[e.b] = -2;
        if (pv != -1) snext[pv] = m; else head = m;
        if (nx != -1) sprev[nx] = m;
        --live;
        ++seq;
        if (pv != -1) try_push(pv, m);
        if (nx != -1) try_push(m, nx);
    }
    int cnt = 0;
    for (int i = head; i != -1; i = snext[i]) out[cnt++] = sid[i];
    return cnt;
}

}  // namespace ovtk
