// tables.hpp -- read-only lookup tables of the tokenizer ops: host-side builders (plain C++) and
// the POD views the kernels receive.  Built once per op handle from the op's constant inputs,
// i.e. what the reference builds lazily on first evaluate():
//   BPE   vocab map / merges map / trie      src/bpe_tokenizer.cpp:50-120, 341-388
//   trie  Trie::add / find_longest           src/utils.cpp:464-538
//   WordPiece root + "##" tries              src/wordpiece_tokenizer.cpp:51-73
//   VocabEncoder string -> value map         src/vocab_encoder.cpp:62-79
//
// Device layouts (all flat arrays, sized for L2 residency, probed with one or two 8/16-byte loads):
//   trie      root[256] {value, child|leaf bit}; edges: open-addressing table of 16-byte {key = node<<8|byte, child,
//             value at the child, child has children} -- one load per byte of the longest-match walk
//   merges    cuckoo table, 2 hash functions x buckets of two 16-byte slots {left:21 | right:21 |
//             rank:22} + {new_id}: a lookup is FOUR independent global_load_dwordx4 (both buckets),
//             never a probe chain -- a wave waits for one round trip, not for its unluckiest lane;
//             new_id[rank] also kept as a dense side array for the exact-heap path
//   pieces    the piece memo: cuckoo table, 2 hash functions x one 32-byte entry {piece bytes
//             (<= 15) + length : 16 B, ids[3], count} holding BPE(piece) for every vocabulary token
//             used as a whole piece, computed once at create time by the device BPE itself (see
//             api_encode.cpp); a lookup is two independent 32-byte loads (a divergent load costs the texture addresser ~64 line lookups whatever its width: fewer, not narrower)
//   strings   (VocabEncoder) open-addressing table of {hash32, key index}; keys stay in the
//             decomposed begins/ends/chars form for the final byte compare
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

namespace ovtk {

struct I2 { int32_t x, y; };

constexpr uint64_t kEmptySlot = ~0ull;
constexpr int kMaxVocabBits = 21;            // ids < 2^21 - 1
constexpr int kMaxRankBits = 22;             // merges < 2^22 - 1
constexpr uint32_t kNoRank = (1u << kMaxRankBits) - 1;
constexpr int kMaxSuffix = 6;                // end_suffix bytes supported on the device
constexpr int32_t kLeafBit = 1 << 30;        // root[b].y / node child field: node has no children

// ---- device views -----------------------------------------------------------------------
// One edge of the flattened trie WITH what the walk wants to know about the node it leads to: a step of the longest-match
// walk is one 16-byte load (edge and node apart were two dependent loads per byte, and the walk is the longest chain
// of dependent loads in merge_kernel).
struct alignas(16) TrieEdge {
    uint32_t key;       // parent node << 8 | byte; kNoEdge = free slot
    int32_t child;      // node index
    int32_t value;      // token id ending at the child, or -1
    int32_t has_kids;   // 0: the child is a leaf
};
constexpr uint32_t kNoEdge = 0xFFFFFFFFu;
struct TrieDev {
    const I2* root;          // [256] x = token id ending at this byte or -1; y = node | kLeafBit, or -1 (no child)
    const TrieEdge* edges;   // [edge_mask+1], open addressing (linear probing, load <= 0.5)
    uint32_t edge_mask;
    uint32_t edge_shift;     // 32 - log2(capacity)
};

// TrieTokenizer's own form of the edges (round 6): BUCKETS of four 8-byte entries, one 32-byte sector per step whether the edge exists or
// not (the open-addressed 16-byte TrieEdge costs 1.5 reads when it does and 2.5 when it does not, and the walk is bound by the rate of
// exactly these reads), half the bytes per edge.  An entry: key = has_kids << 31 | parent node << 8 | byte (all ones: free), value = the
// token that ends here or -1.  The node an entry leads to is the entry's own index, 4 * bucket + slot (the root's edges are entries like the
// others, parent kTrieRoot, and once more a 256-entry table for the kernels' LDS).  A bucket that is full sends its surplus to the next
// one: a lookup goes on only from a bucket without a free entry.
struct alignas(32) TrieBucket {
    uint32_t kv[8];   // key0, value0, key1, value1, ...
};
constexpr uint32_t kTrieFree = 0xFFFFFFFFu;
constexpr uint32_t kTrieKids = 0x80000000u;
// the bucket of the edge (node, byte): two 24-bit multiplies (full rate on gfx950; a 32-bit one is a quarter) -- nodes are entry indices,
// spread by the hash that placed them
__host__ __device__ inline uint32_t trie_bucket_of(uint32_t node, uint32_t byte, uint32_t mask) {
    const uint32_t h = (node & 0xFFFFFFu) * 0x9E3779u + byte * 0x85EBCBu;   // (both products below 2^48: the low 32 bits, as v_mul_u32_u24 gives them)
    return ((h >> 9) ^ h) & mask;
}
constexpr int kTrieRoot = (1 << 23) - 2;    // (a key, node << 8 | byte, stays below 2^31 - 1: never a free entry's all-ones less the kids bit)
constexpr int kTrieMaxNodes = kTrieRoot - 1;
struct TrieBucketsDev {
    const I2* root;   // [256] x = token id ending at this byte or -1; y = node | kLeafBit when nothing goes on from it, or -1 (no token starts with this byte)
    const TrieBucket* buckets;
    uint32_t bucket_mask, bucket_shift;
};

struct alignas(16) MergeSlot {
    uint64_t kr;   // merge_key(left, right) << kMaxRankBits | rank; kEmptySlot = free
    uint64_t nid;  // id of the merged token
};

constexpr int kPieceKeyBytes = 15;  // longest piece the memo table can key
constexpr int kPieceMaxIds = 3;     // longest id sequence it stores
constexpr int kPieceMaxIds6 = 6;    // ... of a table whose ids are u16 pairs (PieceTableDev::packed6: the fused WordPiece path's word memo,
                                    // every id a vocabulary index below 65 536): tok[] holds six u16 ids
struct alignas(32) PieceEntry {
    uint64_t k0, k1;  // piece bytes 0..7 / 8..14 (little endian, zero padded) | length << 56; k1 == 0: free (length >= 1)
    int32_t tok[kPieceMaxIds];
    uint32_t tag;     // piece_tag(piece_mix(k0, k1), id count): the payload half says which key it belongs to
};
// merge_kernel adds entries while lookups of another stream read the table (the dynamic part of the memo, below).  The
// two 16-byte halves of an entry are written by two stores, so a reader may see a key whose payload has not arrived:
// the tag makes that a miss (24 bits of the key's hash + a bit that a zeroed payload lacks) instead of a wrong answer.
// What the protocol relies on, stated once: (i) a slot is written ONCE -- claimed by CAS, payload stored, key stored -- and never
// changes afterwards, so the only incomplete state a reader can meet is "key there, payload still zero", which the tag's valid
// bit rejects; (ii) each half is ONE aligned 16-byte store (global_store_dwordx4) and is read by ONE aligned 16-byte load: both
// lie inside one 32-byte sector of one cache line, and the memory system moves sectors, not dwords -- a reader sees the half
// before or after the store, not a mixture (the HIP memory model does not promise this; gfx9 hardware behaves so, and the
// multi-stream learning soak, tests/test_bpe_parity.py::test_memo_learns_under_concurrent_lookups, is the regression test).
// The piece store below, whose lookups are off the hot path, additionally carries a checksum of its ids in the tag.
constexpr uint32_t kPieceBusy = 0xFF000000u;  // dword 3 of a key (length byte 255) while a writer fills the slot it claimed
struct PieceTableDev {
    const PieceEntry* slots;  // nullptr: no memo (every piece takes the merge path)
    uint32_t shift;           // 32 - log2(slots)
    int32_t* room;            // entries merge_kernel may still add (ovtk_bpe_params::memo_learn at create: cache_capacity, or as many as the
                              // store holds); nullptr: a fixed table.
                              // kRoomShards counters, kRoomStride ints apart (one per 128-byte line), that share the capacity:
                              // every wave-batch with something to file takes its room with a RETURNING add (that is what keeps
                              // `learned <= the room at create` exact), and 3 700 of those per launch on ONE address are most of a
                              // 50-us kernel (a capacity that never fills: 0.125 -> 0.166 ms per step at cache_capacity = 200 000)
    uint32_t room_mask;       // kRoomShards - 1, or 0 for small capacities (one counter holds it all)
    uint32_t packed6;         // 1: tok[] of every entry is six u16 ids (id 2k in the low half of tok[k]), up to kPieceMaxIds6 of them
};
constexpr int kRoomShards = 16, kRoomStride = 32;
// ---- the piece store: the memo's second level, probed by merge_kernel only (never by the lookup kernels).
// The first level above is sized for an XCD's L2 and for ONE round trip in the hot loop: 15-byte keys, 3 ids, the
// vocabulary's own tokens plus the learned pieces (up to six u16 ids each where the ids fit; round 5).  What it does not hold reaches merge_kernel as a deferred
// piece; before merging it, merge_kernel asks the store -- 64-byte entries, pieces up to 31 bytes, up to 15 ids (u16, every
// id < 65536) or 7 ids (i32) -- and files there what it had to merge.  A hit costs the piece one round trip instead of a chain
// of ~12 dependent ones.  Same pure function piece -> ids, same insert-only protocol as the first level (a slot is claimed
// with a CAS on the key's last dword, payload and key are written once and never change; a reader takes an entry whose 32 key
// bytes match and whose payload carries the valid bit and the checksum of its own ids).
constexpr int kStoreKeyBytes = 31;
constexpr int kStoreIds16 = 15, kStoreIds32 = 7;
// The word store's entry for a word without a WordPiece segmentation: one id that is no vocabulary index (a narrow store belongs to a
// vocabulary of at most 65 535 words), read back as the call's unk_token_id (input 8 may differ from call to call).
constexpr int32_t kStoreUnk16 = 0xFFFF, kStoreUnk32 = 0x7FFFFFFF;
struct alignas(64) StoreEntry {
    uint32_t key[8];  // piece bytes 0..30 (little endian, zero padded), byte 31 = length; key[7] == 0: free (length >= 1)
    uint32_t pay[8];  // narrow: u16 ids[15], u16 tag   wide: i32 ids[7], u32 tag
};
struct PieceStoreDev {
    StoreEntry* slots;  // nullptr: no store
    uint32_t shift;     // 32 - log2(capacity)
    int32_t* room;      // entries that may still be added
    int32_t narrow;     // every id < 65536
};
__host__ __device__ inline uint32_t store_mix(const uint32_t (&key)[8]) {
    uint32_t h = 0x811C9DC5u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 8; ++i) {
        h = (h ^ key[i]) * 0x9E3779B1u;
        h ^= h >> 15;
    }
    return h;
}
constexpr int kStoreWays = 2;  // candidate slots of a piece: the two halves of ONE 128-byte line (a lookup fetches both, an insert
                               // takes the first free one).  Measured on the way: two independent lines per piece doubled the
                               // kernel's counted HBM traffic for the same hit rate (merge_kernel 47 -> 81 MB per config-2 batch);
                               // four ways: 1 % instead of 6 % failed inserts, merge_kernel 58 -> 69 us alone, step 0.142 -> 0.146 ms
__host__ __device__ inline uint32_t store_h(uint32_t mix, int which, uint32_t shift) {  // shift = 32 - log2(lines)
    return (((mix * 0x2C1B3C6Du) >> shift) << 1) | uint32_t(which);
}
// A piece whose line is full goes to the first of the next kStoreOverflow lines that has room, and a lookup follows it there --
// but only past lines it has seen FULL (entries never leave: a line with a free slot ends the search).  Nearly every lookup is still
// the one line; what this buys is that no piece is left without a place: at a table a tenth full one insert in a hundred found its
// line taken, and every later call sent exactly those pieces down the slow path -- a dozen dependent probes of one lane, which
// is what merge_kernel and wordpiece_deferred_kernel then took as a whole (round 4).
constexpr int kStoreOverflow = 3;
__host__ __device__ inline uint32_t store_h_next(uint32_t slot, int step, uint32_t shift) {  // slot `which` of the step-th line behind slot's
    const uint32_t lines = 1u << (32u - shift);
    return ((((slot >> 1) + uint32_t(step)) & (lines - 1u)) << 1) | (slot & 1u);
}
// tag of a payload: valid bit | id count | checksum of the ids (a payload that is not completely there does not pass)
__host__ __device__ inline uint32_t store_fold(const uint32_t (&pay)[8], bool narrow) {
    uint32_t x = pay[0] ^ pay[1] * 3u ^ pay[2] * 5u ^ pay[3] * 7u ^ pay[4] * 11u ^ pay[5] * 13u ^ pay[6] * 17u ^
                 (narrow ? (pay[7] & 0xFFFFu) * 19u : 0u);
    x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t store_tag16(const uint32_t (&pay)[8], int cnt) {   // the upper half of pay[7]
    return 0x8000u | (uint32_t(cnt) << 11) | (store_fold(pay, true) & 0x7FFu);
}
__host__ __device__ inline uint32_t store_tag32(const uint32_t (&pay)[8], int cnt) {   // pay[7]
    return 0x80000000u | (uint32_t(cnt) << 24) | (store_fold(pay, false) & 0xFFFFFFu);
}

struct alignas(16) MergeBucket { MergeSlot s[1]; };  // one slot per bucket: a lookup is two 16-byte loads (divergent loads cost per instruction)

struct BpeDev {
    TrieDev trie;
    const MergeBucket* merges;  // [1 << (32 - bucket_shift)]
    uint32_t bucket_shift;      // 32 - log2(buckets)
    const int32_t* new_id;      // [n_merges]
    PieceTableDev pieces;
    PieceStoreDev store;
    const int32_t* byte_fallback_id;  // [256], -1 = none (all -1 when byte_fallback is off)
    int32_t unk_id;
    int32_t suffix_len;
    uint8_t suffix[8];
};

struct StringMapDev {
    const uint64_t* slots;   // {hash32 : 32 | key index : 32}, kEmptySlot = free
    uint32_t mask;
    const int32_t* key_begins;
    const int32_t* key_ends;
    const uint8_t* key_chars;
    const void* values;      // i32 or i64 [n_keys]
    int32_t value_size;
};

__host__ __device__ inline uint32_t hash_u32(uint32_t k) { return k * 0x9E3779B1u; }
__host__ __device__ inline uint64_t hash_u64(uint64_t k) { return k * 0x9E3779B97F4A7C15ull; }
__host__ __device__ inline uint64_t merge_key(uint32_t l, uint32_t r) { return (uint64_t(l) << kMaxVocabBits) | r; }
// Cuckoo hash functions: the top log2(buckets) bits of odd-constant multiplies (shift = 64 - log2(buckets) < 64).
// (32-bit arithmetic: a 64-bit multiply costs ~5 vector instructions at a quarter of the rate; key < 2^42)
__host__ __device__ inline uint32_t merge_mix(uint64_t key) {
    uint32_t h = uint32_t(key) * 0x9E3779B1u + uint32_t(key >> 32) * 0x85EBCA77u;
    return h ^ (h >> 15);
}
__host__ __device__ inline uint32_t merge_h1(uint64_t key, uint32_t shift) { return (merge_mix(key) * 0x2C1B3C6Du) >> shift; }
__host__ __device__ inline uint32_t merge_h2(uint64_t key, uint32_t shift) { return (merge_mix(key) * 0xD6E8FEB9u) >> shift; }
// Memo hashing with 24-bit multiplies only.  A full 32-bit multiply (v_mul_lo_u32) issues at a quarter of the vector rate
// on CDNA; v_mul_u32_u24 / v_mad_u32_u24 (the low 32 bits of a 24 x 24-bit product, what the compiler selects for operands
// it knows to be 24 bits wide) issue at the full rate.  The 16 key bytes are taken as six overlapping 24-bit chunks, each
// times an odd 24-bit constant.
// (round 2: 7 v_mul_lo_u32 became 15 24-bit multiplies / multiply-adds, 100.4 -> 98.8 us; the tables fill as before.)
__host__ __device__ inline uint32_t mul24(uint32_t a, uint32_t k) { return (a & 0xFFFFFFu) * (k & 0xFFFFFFu); }
__host__ __device__ inline uint32_t piece_mix(uint64_t k0, uint64_t k1) {
    const uint32_t d0 = uint32_t(k0), d1 = uint32_t(k0 >> 32), d2 = uint32_t(k1), d3 = uint32_t(k1 >> 32);
    const uint32_t c1 = (d0 >> 24) | (d1 << 8), c2 = (d1 >> 16) | (d2 << 16);   // funnel shifts; bits above 24 are ignored
    uint32_t h = mul24(d0, 0x9E3779u) + mul24(c1, 0x85EBCBu) + mul24(c2, 0xC2B2AFu) + mul24(d2 >> 8, 0x27D4EBu) +
                 mul24(d3, 0x165667u) + mul24(d3 >> 8, 0xD6E8FFu);
    // (two more stirring products were measured and dropped in round 3, the closing h ^ (h >> 15) in round 4: the slot index is the
    // sum's top bits, the tag only has to be a function of the key)
    return h;
}
__host__ __device__ inline uint32_t piece_tag(uint32_t mix, int cnt) { return (mix & 0xFFFFFF00u) | 0x80u | uint32_t(cnt); }
// A piece has ONE candidate entry (round 4): its 32-byte slot, two 16-byte loads at consecutive addresses, one key compare.
// History: two independent candidates (cuckoo, r01: two cache lines per lookup, one of them cold), then the two halves of one
// 64-byte bucket (r03: one line, four loads), now a direct-mapped table eight times sparser.  The lookup kernels are bound by
// instruction issue and by the texture addresser's per-instruction cost of divergent loads: the second candidate was two
// loads and ~12 vector instructions of every probe for the sake of a few hundred vocabulary tokens.  The price is placement
// freedom, paid in table size: at 1/12 full ~3 % of the entries find their slot taken (on the host too; they are the late,
// rare ones -- a vocabulary is placed in ascending id order) and simply stay misses: merge path / piece store, same result.
__host__ __device__ inline uint32_t piece_h(uint32_t mix, uint32_t shift) {  // shift = 32 - log2(slots)
    return mix >> shift;
}
// Hash of a byte string, four bytes per step (little-endian words, the last one zero-padded), the same function on host
// (table build) and device (probe).  The device feeds it words it already holds in registers: hash_words().
__host__ __device__ inline uint32_t hash_word_step(uint32_t h, uint32_t w) { return (h ^ w) * 0x9E3779B1u; }
__host__ __device__ inline uint32_t hash_finish(uint32_t h, int n) {
    h ^= uint32_t(n) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    return h ^ (h >> 13);
}
__host__ __device__ inline uint32_t hash_bytes(const uint8_t* p, int n) {
    uint32_t h = 2166136261u;
    for (int i = 0; i < n; i += 4) {
        uint32_t w = 0;
        for (int j = 0; j < 4 && i + j < n; ++j) w |= uint32_t(p[i + j]) << (8 * j);
        h = hash_word_step(h, w);
    }
    return hash_finish(h, n);
}

uint32_t pow2_at_least(uint64_t n);
int log2u(uint32_t c);

// ---- host builders ----------------------------------------------------------------------
struct TrieHost {
    std::vector<I2> root;
    std::vector<TrieEdge> edges;
    uint32_t edge_mask = 0, edge_shift = 32;

    // Incremental form used while inserting.
    struct Build { std::vector<std::vector<std::pair<uint8_t, int>>> kids; std::vector<int32_t> value; } b;
    TrieHost();
    void add(const uint8_t* s, size_t n, int32_t value);  // later add of the same string overwrites
    void finalize();                                      // flattens b -> root/node/edges
};

// TrieHost's nodes as TrieBucketsDev's tables; false: more nodes than an entry's key can name
struct TrieBucketsHost {
    std::vector<I2> root;
    std::vector<TrieBucket> buckets;
    uint32_t bucket_mask = 0, bucket_shift = 32;
    bool build(const TrieHost& t);
};

struct BpeHost {
    TrieHost trie;
    std::vector<MergeBucket> merges;
    uint32_t bucket_shift = 30;
    std::vector<int32_t> new_id;
    std::vector<int32_t> byte_fallback_id;
    int32_t unk_id = -1;
    std::string suffix;
};

struct StringsView { const int32_t* begins; const int32_t* ends; const uint8_t* chars; int64_t n; };

// Returns 0 or a negative OVTK_E_* code with `err` filled.
int build_bpe(const StringsView& vocab, const StringsView& merges_left, const StringsView* merges_right,
              const StringsView& added, const int32_t* added_ids, const std::string& unk_token,
              const std::string& end_suffix, bool byte_fallback, BpeHost& out, std::string& err);

// The piece memo from (piece string, its ids) pairs: pieces of 1..kPieceKeyBytes bytes with at most
// kPieceMaxIds ids are stored (a repeated string keeps its first entry -- all entries of one string are equal).
struct PieceTableHost {
    std::vector<PieceEntry> slots;   // direct-mapped: piece_h() is the slot
    uint32_t shift = 30;             // 32 - log2(slots)
    size_t stored = 0, refused = 0;  // refused: pieces whose slot was taken (they stay misses)
};
// extra: entries the table is sized for beyond the stored ones (cache_capacity; what is learned beyond that competes for the slots there are).
void build_piece_table(const StringsView& pieces, const int32_t* id_begins, const int32_t* id_ends, const int32_t* ids,
                       PieceTableHost& out, size_t extra = 0, bool packed6 = false);

int build_wordpiece(const StringsView& vocab, const std::string& suffix_indicator, TrieHost& root, TrieHost& sub,
                    std::string& err);

struct StringMapHost {
    std::vector<uint64_t> slots;
    uint32_t mask = 0;
    std::vector<int32_t> key_begins, key_ends;
    std::vector<uint8_t> key_chars;
};
int build_string_map(const StringsView& keys, StringMapHost& out, std::string& err);

}  // namespace ovtk
