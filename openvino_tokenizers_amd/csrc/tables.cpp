// tables.cpp -- host-side construction of the device lookup tables (see tables.hpp).
#include <hip/hip_runtime.h>

#include "tables.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>

#include "../../include/ovtk_amd.h"

namespace ovtk {

namespace {
std::string view_str(const StringsView& v, int64_t i) {
    return std::string(reinterpret_cast<const char*>(v.chars) + v.begins[i],
                       reinterpret_cast<const char*>(v.chars) + v.ends[i]);
}
}  // namespace
uint32_t pow2_at_least(uint64_t n) {
    uint32_t c = 1;
    while (c < n) c <<= 1;
    return c;
}
int log2u(uint32_t c) {
    int b = 0;
    while ((1u << b) < c) ++b;
    return b;
}

// ------------------------------------------------------------------------------- trie
TrieHost::TrieHost() {
    b.kids.emplace_back();
    b.value.push_back(-1);
}

void TrieHost::add(const uint8_t* s, size_t n, int32_t value) {
    int cur = 0;
    for (size_t i = 0; i < n; ++i) {
        auto& ks = b.kids[cur];
        auto it = std::lower_bound(ks.begin(), ks.end(), s[i], [](const auto& k, uint8_t c) { return k.first < c; });
        if (it == ks.end() || it->first != s[i]) {
            const int fresh = int(b.kids.size());
            it = ks.insert(it, {s[i], fresh});
            const int child = it->second;  // read before the vectors below may reallocate b.kids
            b.kids.emplace_back();
            b.value.push_back(-1);
            cur = child;
        } else {
            cur = it->second;
        }
    }
    b.value[cur] = value;  // same string added twice: the later value stays (src/utils.cpp:476-478)
}

void TrieHost::finalize() {
    const size_t n = b.kids.size();
    size_t n_edges = 0;
    for (size_t i = 1; i < n; ++i) n_edges += b.kids[i].size();
    root.assign(256, I2{-1, -1});
    for (const auto& k : b.kids[0]) {
        const int child = k.second;
        root[k.first] = I2{b.value[child], child | (b.kids[child].empty() ? kLeafBit : 0)};
    }
    const uint32_t cap = std::max<uint32_t>(8, pow2_at_least(uint64_t(n_edges) * 2 + 1));
    edge_mask = cap - 1;
    edge_shift = 32 - log2u(cap);
    edges.assign(cap, TrieEdge{kNoEdge, -1, -1, 0});
    for (size_t i = 1; i < n; ++i)
        for (const auto& k : b.kids[i]) {
            const uint32_t key = (uint32_t(i) << 8) | k.first;
            uint32_t idx = (hash_u32(key) >> edge_shift) & edge_mask;
            while (edges[idx].key != kNoEdge) idx = (idx + 1) & edge_mask;
            const int child = k.second;
            edges[idx] = TrieEdge{key, child, b.value[child], b.kids[child].empty() ? 0 : 1};
        }
}

bool TrieBucketsHost::build(const TrieHost& t) {
    const auto& kids = t.b.kids;
    const auto& value = t.b.value;
    size_t n_edges = 0;
    for (size_t i = 0; i < kids.size(); ++i) n_edges += kids[i].size();
    const uint32_t n_buckets = std::max<uint32_t>(2, pow2_at_least(uint64_t(n_edges) / 2 + 1));   // one to two entries of four taken
    if (4ull * n_buckets > uint64_t(kTrieMaxNodes)) return false;
    bucket_mask = n_buckets - 1;
    bucket_shift = 32 - log2u(n_buckets);
    TrieBucket free_bucket;
    for (int k = 0; k < 8; ++k) free_bucket.kv[k] = kTrieFree;
    buckets.assign(n_buckets, free_bucket);
    root.assign(256, I2{-1, -1});
    // breadth first: an edge's key holds its parent's number, which is the place the parent's own edge was given
    std::vector<std::pair<int, int>> queue{{0, kTrieRoot}};   // (host node, its number here)
    for (size_t q = 0; q < queue.size(); ++q) {
        const int host = queue[q].first, node = queue[q].second;
        for (const auto& k : kids[size_t(host)]) {
            const uint32_t key = (uint32_t(node) << 8) | k.first;
            uint32_t bk = trie_bucket_of(uint32_t(node), k.first, bucket_mask);
            int slot = -1;
            for (;; bk = (bk + 1) & bucket_mask) {
                for (int j = 0; j < 4 && slot < 0; ++j)
                    if (buckets[bk].kv[2 * j] == kTrieFree) slot = j;
                if (slot >= 0) break;
            }
            const int child = k.second;
            buckets[bk].kv[2 * slot] = key | (kids[size_t(child)].empty() ? 0u : kTrieKids);
            buckets[bk].kv[2 * slot + 1] = uint32_t(value[size_t(child)]);
            queue.emplace_back(child, int(4u * bk + uint32_t(slot)));
            if (host == 0) root[k.first] = I2{value[size_t(child)], int(4u * bk + uint32_t(slot)) | (kids[size_t(child)].empty() ? kLeafBit : 0)};
        }
    }
    return true;
}

// ------------------------------------------------------------------------------- BPE
namespace {
// Cuckoo insertion by random walk (deterministic xorshift): `choices(item, idx)` fills the candidate slot
// indices of an item, `empty(slot)` tells a free slot.  Returns false when the walk does not terminate.
template <class Slot, int N, class Choices, class Empty>
bool cuckoo_insert(std::vector<Slot>& slots, Slot item, Choices&& choices, Empty&& empty, uint64_t& rng) {
    for (int kick = 0; kick < 2000; ++kick) {
        uint32_t idx[N];
        choices(item, idx);
        for (int c = 0; c < N; ++c)
            if (empty(slots[idx[c]])) {
                slots[idx[c]] = item;
                return true;
            }
        rng ^= rng << 13;
        rng ^= rng >> 7;
        rng ^= rng << 17;
        std::swap(item, slots[idx[rng % N]]);
    }
    return false;
}
}  // namespace

int build_bpe(const StringsView& vocab, const StringsView& ml, const StringsView* mr, const StringsView& added,
              const int32_t* added_ids, const std::string& unk_token, const std::string& end_suffix,
              bool byte_fallback, BpeHost& out, std::string& err) {
    if (vocab.n >= (int64_t(1) << kMaxVocabBits) - 1) {
        err = "BPETokenizer: vocabularies of 2^21 or more tokens are not supported by the device tables";
        return OVTK_E_UNSUPPORTED;
    }
    if (ml.n >= int64_t(kNoRank)) {
        err = "BPETokenizer: 2^22 or more merges are not supported by the device tables";
        return OVTK_E_UNSUPPORTED;
    }
    if (end_suffix.size() > size_t(kMaxSuffix)) {
        err = "BPETokenizer: end_suffix longer than 6 bytes is not supported on the device";
        return OVTK_E_UNSUPPORTED;
    }
    // Added tokens: ordered map, first occurrence of a string kept (bpe_tokenizer.cpp:51-67).
    std::map<std::string, int32_t> added_map;
    for (int64_t i = 0; i < added.n; ++i) {
        if (added_ids[i] < 0 || added_ids[i] >= (1 << kMaxVocabBits) - 1) {
            err = "BPETokenizer: added-token id outside [0, 2^21-1) is not supported by the device tables";
            return OVTK_E_UNSUPPORTED;
        }
        added_map.insert({view_str(added, i), added_ids[i]});
    }
    // token string -> id, the later id wins (bpe_tokenizer.cpp:79-82); added tokens never overwrite (:110-114).
    std::unordered_map<std::string, int32_t> tok;
    tok.reserve(size_t(vocab.n + added.n) * 2);
    for (int64_t id = 0; id < vocab.n; ++id) tok[view_str(vocab, id)] = int32_t(id);
    for (const auto& kv : added_map) tok.insert(kv);

    out.unk_id = -1;
    if (auto it = tok.find(unk_token); it != tok.end()) out.unk_id = it->second;  // :353-355

    // Merges in rank order; a repeated (left,right) pair keeps the LAST rank/new_id (bpe_tokenizer.hpp:58-66).
    struct Pair { uint32_t l, r; };
    std::unordered_map<uint64_t, uint32_t> rank_of;
    rank_of.reserve(size_t(ml.n) * 2);
    out.new_id.assign(size_t(ml.n), -1);
    std::vector<std::string> merged;
    merged.reserve(size_t(ml.n));
    for (int64_t i = 0; i < ml.n; ++i) {
        std::string left, right;
        if (mr) {
            left = view_str(ml, i);
            right = view_str(*mr, i);
        } else {  // "left right" split at the first space (:94-95); no space: substr(npos + 1) == whole line
            std::string line = view_str(ml, i);
            const size_t sp = line.find(' ');
            left = line.substr(0, sp);
            right = line.substr(sp + 1);
        }
        auto l = tok.find(left), r = tok.find(right);
        std::string both = left + right;
        auto z = tok.find(both);
        if (l == tok.end() || r == tok.end() || z == tok.end()) {
            err = "BPETokenizer: merge " + std::to_string(i) + " references a token that is not in the vocabulary";
            return OVTK_E_VOCAB;
        }
        rank_of[merge_key(uint32_t(l->second), uint32_t(r->second))] = uint32_t(i);
        out.new_id[size_t(i)] = z->second;
        merged.push_back(std::move(both));
    }
    // Merge results leave the vocabulary; the trie holds what remains (:375-386).
    for (const auto& m : merged) tok.erase(m);
    for (const auto& kv : tok)
        out.trie.add(reinterpret_cast<const uint8_t*>(kv.first.data()), kv.first.size(), kv.second);
    out.trie.finalize();

    out.byte_fallback_id.assign(256, -1);
    if (byte_fallback) {  // "<0x%02X>" looked up in the post-erasure vocabulary (:242-248)
        char buf[8];
        for (int b = 0; b < 256; ++b) {
            std::snprintf(buf, sizeof buf, "<0x%02X>", b);
            if (auto it = tok.find(buf); it != tok.end()) out.byte_fallback_id[size_t(b)] = it->second;
        }
    }
    out.suffix = end_suffix;

    // Cuckoo table: 2 hash functions x 1 slot (load < 0.5 of the slots; grown until every pair is placed).
    for (uint32_t buckets = std::max<uint32_t>(4, pow2_at_least(uint64_t(rank_of.size()) * 2 + 1));; buckets *= 2) {
        out.bucket_shift = 32 - log2u(buckets);
        std::vector<MergeSlot> flat(size_t(buckets), MergeSlot{kEmptySlot, 0});
        uint64_t rng = 0x2545F4914F6CDD1Dull;
        bool ok = true;
        const uint32_t shift = out.bucket_shift;
        for (const auto& kv : rank_of) {
            const MergeSlot item{(kv.first << kMaxRankBits) | kv.second, uint64_t(uint32_t(out.new_id[kv.second]))};
            ok = cuckoo_insert<MergeSlot, 2>(
                flat, item,
                [shift](const MergeSlot& m, uint32_t* idx) {
                    const uint64_t key = m.kr >> kMaxRankBits;
                    idx[0] = merge_h1(key, shift);
                    idx[1] = merge_h2(key, shift);
                },
                [](const MergeSlot& m) { return m.kr == kEmptySlot; }, rng);
            if (!ok) break;
        }
        if (!ok) continue;
        out.merges.resize(buckets);
        std::memcpy(out.merges.data(), flat.data(), flat.size() * sizeof(MergeSlot));
        break;
    }
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- piece memo
void build_piece_table(const StringsView& pieces, const int32_t* id_begins, const int32_t* id_ends, const int32_t* ids,
                       PieceTableHost& out, size_t extra, bool packed6) {
    const int max_ids = packed6 ? kPieceMaxIds6 : kPieceMaxIds;
    // in the order of the input (for a vocabulary: ascending id, i.e. roughly descending frequency -- the pieces a full
    // bucket refuses are the late, rare ones); a repeated string keeps its first entry
    std::unordered_map<std::string, size_t> seen;
    std::vector<PieceEntry> list;
    for (int64_t i = 0; i < pieces.n; ++i) {
        const int len = pieces.ends[i] - pieces.begins[i], cnt = id_ends[i] - id_begins[i];
        if (!(len >= 1 && len <= kPieceKeyBytes && cnt >= 0 && cnt <= max_ids)) continue;
        uint8_t kb[16] = {0};
        std::memcpy(kb, pieces.chars + pieces.begins[i], size_t(len));
        kb[15] = uint8_t(len);
        PieceEntry e{0, 0, {0, 0, 0}, 0};
        std::memcpy(&e.k0, kb, 8);
        std::memcpy(&e.k1, kb + 8, 8);
        e.tag = piece_tag(piece_mix(e.k0, e.k1), cnt);
        for (int k = 0; k < cnt; ++k) {
            const int32_t id = ids[id_begins[i] + k];
            if (packed6) e.tok[k >> 1] |= int32_t(uint32_t(id & 0xFFFF) << (16 * (k & 1)));   // (the caller vouches: every id < 65 536)
            else e.tok[k] = id;
        }
        if (seen.emplace(std::string(reinterpret_cast<const char*>(kb), 16), list.size()).second) list.push_back(e);
    }
    // direct-mapped, no relocation: at 1/12 full about three entries in a hundred find their slot taken (tables.hpp piece_h)
    const uint32_t slots = std::max<uint32_t>(8, pow2_at_least(uint64_t(list.size() + extra) * 12));
    out.shift = 32 - log2u(slots);
    out.slots.assign(size_t(slots), PieceEntry{0, 0, {0, 0, 0}, 0});
    out.stored = out.refused = 0;
    for (const PieceEntry& e : list) {
        PieceEntry& b = out.slots[piece_h(piece_mix(e.k0, e.k1), out.shift)];
        if (b.k1 != 0) { ++out.refused; continue; }
        b = e;
        ++out.stored;
    }
}

// ------------------------------------------------------------------------------- WordPiece
int build_wordpiece(const StringsView& vocab, const std::string& si, TrieHost& root, TrieHost& sub, std::string& err) {
    (void)err;
    for (int64_t id = 0; id < vocab.n; ++id) {  // wordpiece_tokenizer.cpp:59-71
        const uint8_t* p = vocab.chars + vocab.begins[id];
        const size_t n = size_t(vocab.ends[id] - vocab.begins[id]);
        if (n >= si.size() && std::memcmp(p, si.data(), si.size()) == 0) sub.add(p + si.size(), n - si.size(), int32_t(id));
        else root.add(p, n, int32_t(id));
    }
    root.finalize();
    sub.finalize();
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- VocabEncoder
int build_string_map(const StringsView& keys, StringMapHost& out, std::string& err) {
    (void)err;
    const uint32_t cap = std::max<uint32_t>(8, pow2_at_least(uint64_t(keys.n) * 2 + 1));
    out.mask = cap - 1;
    out.slots.assign(cap, kEmptySlot);
    out.key_begins.assign(keys.begins, keys.begins + keys.n);
    out.key_ends.assign(keys.ends, keys.ends + keys.n);
    int64_t hi = 0;
    for (int64_t i = 0; i < keys.n; ++i) hi = std::max<int64_t>(hi, keys.ends[i]);
    out.key_chars.assign(keys.chars, keys.chars + hi);
    for (int64_t i = 0; i < keys.n; ++i) {  // insert(): the FIRST occurrence of a key wins (vocab_encoder.cpp:76)
        const uint8_t* p = keys.chars + keys.begins[i];
        const int n = keys.ends[i] - keys.begins[i];
        const uint32_t h = hash_bytes(p, n);
        uint32_t idx = h & out.mask;
        bool dup = false;
        while (out.slots[idx] != kEmptySlot) {
            if (uint32_t(out.slots[idx] >> 32) == h) {
                const int64_t j = int64_t(uint32_t(out.slots[idx]));
                const int m = keys.ends[j] - keys.begins[j];
                if (m == n && std::memcmp(keys.chars + keys.begins[j], p, size_t(n)) == 0) { dup = true; break; }
            }
            idx = (idx + 1) & out.mask;
        }
        if (!dup) out.slots[idx] = (uint64_t(h) << 32) | uint32_t(i);
    }
    return OVTK_OK;
}

}  // namespace ovtk
