// oracle.cpp -- CPU restatement of the reference's tokenizer hot path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library; the product never does.
//
// Each function states the reference lines it follows (paths relative to /root/reference).
// The data structures deliberately mirror the reference's (node trie with sorted children,
// open-addressing merges map with the same hash, std::priority_queue with the same
// comparator and tuple type, insert-only string cache) so that, timed, this is an honest
// "port" CPU baseline, and so that the libstdc++ heap order -- which decides BPE ties
//  -- is the real one, not an emulation.
#include "oracle.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <shared_mutex>
#include <string>
#include <string_view>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ---------------------------------------------------------------------------------------
// PCRE2 (8-bit) through dlopen: the image ships libpcre2-8.so.0 but no pcre2.h.  The few
// entry points and option bits used by src/utils.cpp:256-272,396-420 are declared here
// from the public PCRE2 API.
// ---------------------------------------------------------------------------------------
struct Pcre2 {
    using code = void;
    using match_data = void;
    code* (*compile)(const uint8_t*, size_t, uint32_t, int*, size_t*, void*) = nullptr;
    int (*jit_compile)(code*, uint32_t) = nullptr;
    match_data* (*md_create)(const code*, void*) = nullptr;
    int (*match)(const code*, const uint8_t*, size_t, size_t, uint32_t, match_data*, void*) = nullptr;
    int (*jit_match)(const code*, const uint8_t*, size_t, size_t, uint32_t, match_data*, void*) = nullptr;
    size_t* (*ovector)(match_data*) = nullptr;
    uint32_t (*ovector_count)(match_data*) = nullptr;
    void (*md_free)(match_data*) = nullptr;
    void (*code_free)(code*) = nullptr;
    bool ok = false;

    static constexpr uint32_t UTF = 0x00080000u;   // PCRE2_UTF
    static constexpr uint32_t UCP = 0x00020000u;   // PCRE2_UCP
    static constexpr uint32_t JIT_COMPLETE = 1u;   // PCRE2_JIT_COMPLETE

    static const Pcre2& get() {
        static Pcre2 p = [] {
            Pcre2 q;
            void* h = dlopen("libpcre2-8.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) return q;
            auto sym = [&](const char* n) { return dlsym(h, n); };
            q.compile = reinterpret_cast<decltype(q.compile)>(sym("pcre2_compile_8"));
            q.jit_compile = reinterpret_cast<decltype(q.jit_compile)>(sym("pcre2_jit_compile_8"));
            q.md_create = reinterpret_cast<decltype(q.md_create)>(sym("pcre2_match_data_create_from_pattern_8"));
            q.match = reinterpret_cast<decltype(q.match)>(sym("pcre2_match_8"));
            q.jit_match = reinterpret_cast<decltype(q.jit_match)>(sym("pcre2_jit_match_8"));
            q.ovector = reinterpret_cast<decltype(q.ovector)>(sym("pcre2_get_ovector_pointer_8"));
            q.ovector_count = reinterpret_cast<decltype(q.ovector_count)>(sym("pcre2_get_ovector_count_8"));
            q.md_free = reinterpret_cast<decltype(q.md_free)>(sym("pcre2_match_data_free_8"));
            q.code_free = reinterpret_cast<decltype(q.code_free)>(sym("pcre2_code_free_8"));
            q.ok = q.compile && q.jit_compile && q.md_create && q.match && q.jit_match && q.ovector && q.ovector_count &&
                   q.md_free && q.code_free;
            return q;
        }();
        return p;
    }
};

std::string str_of(const uint8_t* chars, int32_t b, int32_t e) {
    return std::string(reinterpret_cast<const char*>(chars) + b, reinterpret_cast<const char*>(chars) + e);
}

// ---------------------------------------------------------------------------------------
// Byte trie: src/utils.hpp:111-123, src/utils.cpp:464-538.  One heap node per trie node,
// children kept sorted by byte, binary search per step, longest match wins.
// ---------------------------------------------------------------------------------------
struct TrieNode {
    std::vector<std::pair<uint8_t, std::unique_ptr<TrieNode>>> kids;
    int value = -1;

    const TrieNode* child(uint8_t c) const {
        auto it = std::lower_bound(kids.begin(), kids.end(), c,
                                   [](const auto& k, uint8_t v) { return k.first < v; });
        return (it != kids.end() && it->first == c) ? it->second.get() : nullptr;
    }
    void add(const uint8_t* s, size_t n, int v) {
        TrieNode* cur = this;
        for (size_t i = 0; i < n; ++i) {
            auto it = std::lower_bound(cur->kids.begin(), cur->kids.end(), s[i],
                                       [](const auto& k, uint8_t c) { return k.first < c; });
            if (it == cur->kids.end() || it->first != s[i])
                it = cur->kids.emplace(it, s[i], std::make_unique<TrieNode>());
            cur = it->second.get();
        }
        cur->value = v;  // a later add of the same string overwrites (utils.cpp:476-478)
    }
    // utils.cpp:517-538: precondition idx < n.  Returns id of the longest token starting at
    // idx and moves idx to its end; -1 and idx untouched when nothing matches.
    int find_longest(const uint8_t* s, int n, int& idx) const {
        int best = -1, best_end = idx, i = idx;
        const TrieNode* cur = this;
        uint8_t c = s[i];
        while (const TrieNode* nx = cur->child(c)) {
            cur = nx;
            ++i;
            if (cur->value != -1) {
                best = cur->value;
                best_end = i;
            }
            if (i == n) break;
            c = s[i];
        }
        idx = best_end;
        return best;
    }
};

}  // namespace

extern "C" const char* orc_last_error(void) { return g_err.c_str(); }

// =======================================================================================
// RegexSplit
// =======================================================================================
struct orc_regex {
    void* code = nullptr;
    bool jit = false;
    int mode = 0;  // 0 removed, 1 isolated, 2 merged_with_previous, 3 merged_with_next
    bool invert = false;
    int max_splits = -1;
};

extern "C" int orc_regex_split_create(const char* pattern, int64_t pattern_len, const char* behaviour,
                                      int invert, int max_splits, orc_regex** out) {
    const Pcre2& P = Pcre2::get();
    if (!P.ok) return fail(ORC_E_PCRE2, "libpcre2-8.so.0 not loadable");
    std::string beh(behaviour), pat(pattern, pattern + pattern_len);
    auto r = std::make_unique<orc_regex>();
    // regex_split.cpp:16-22 behaviour map; :113-117 attribute checks.
    if (beh == "remove") r->mode = 0;
    else if (beh == "isolate" || beh == "contiguous") r->mode = 1;
    else if (beh == "mergedwithprevious") r->mode = 2;
    else if (beh == "mergedwithnext") r->mode = 3;
    else return fail(ORC_E_ARG, "unknown split behaviour: " + beh);
    if (!(max_splits == -1 || max_splits > 0)) return fail(ORC_E_ARG, "max_splits must be -1 or > 0");
    // regex_split.cpp:33-37: contiguous wraps the pattern unless it already ends with '+'.
    if (beh == "contiguous" && (pat.empty() || pat.back() != '+')) pat = "(" + pat + ")+";
    // utils.cpp:256-263: compile UTF|UCP, then JIT if available.  A pattern that does not
    // compile leaves code == nullptr: every match "fails" (utils.cpp:264-271,397-399).
    int ec = 0;
    size_t eo = 0;
    r->code = P.compile(reinterpret_cast<const uint8_t*>(pat.data()), pat.size(), Pcre2::UTF | Pcre2::UCP,
                        &ec, &eo, nullptr);
    if (r->code) r->jit = (P.jit_compile(r->code, Pcre2::JIT_COMPLETE) == 0);
    r->invert = invert != 0;
    r->max_splits = max_splits;
    *out = r.release();
    return ORC_OK;
}

// Did pcre2_compile accept the pattern?  (Tests of the null-pattern behaviour, utils.cpp:264-271, name their patterns by this.)
extern "C" int orc_regex_compiled(const orc_regex* r) { return r && r->code ? 1 : 0; }

extern "C" void orc_regex_split_destroy(orc_regex* r) {
    if (!r) return;
    if (r->code) Pcre2::get().code_free(r->code);
    delete r;
}

namespace {
struct MatchData {
    void* md = nullptr;
    explicit MatchData(const orc_regex* r) {
        if (r->code) md = Pcre2::get().md_create(r->code, nullptr);
    }
    ~MatchData() {
        if (md) Pcre2::get().md_free(md);
    }
};
// utils.cpp:396-420 + the non-empty filter of regex_split.cpp:154-161.
bool next_match(const orc_regex* r, MatchData& m, const uint8_t* s, size_t len, size_t start, size_t& b,
                size_t& e) {
    if (!r->code || !m.md) return false;
    const Pcre2& P = Pcre2::get();
    int rc = (r->jit ? P.jit_match : P.match)(r->code, s, len, start, 0, m.md, nullptr);
    if (rc < 0) return false;
    size_t* ov = P.ovector(m.md);
    b = ov[0];
    e = ov[1];
    return b != e;
}
}  // namespace

extern "C" int orc_regex_match(const orc_regex* r, const uint8_t* s, int64_t len, int64_t start, int64_t* m) {
    MatchData md(r);
    size_t b, e;
    if (!next_match(r, md, s, size_t(len), size_t(start), b, e)) return 0;
    m[0] = int64_t(b);
    m[1] = int64_t(e);
    return 1;
}

extern "C" int orc_regex_split_run(const orc_regex* r, const int32_t* rb, const int32_t* re, int64_t B,
                                   const int32_t* begins, const int32_t* ends, int64_t N,
                                   const uint8_t* chars, int64_t nchars, const uint8_t* skips,
                                   int32_t* out_rb, int32_t* out_re, int64_t* n_rows_out,
                                   int32_t* out_begins, int32_t* out_ends, uint8_t* out_skips,
                                   int64_t cap, int64_t* n_out) {
    (void)N;
    // regex_split.cpp:129-143: an all-empty batch yields ragged shape {1} = [0],[0] and the
    // string tensors pass through (caller keeps its inputs).
    if (nchars == 0) {
        out_rb[0] = 0;
        out_re[0] = 0;
        *n_rows_out = 1;
        *n_out = -1;  // "outputs alias the inputs"
        return ORC_OK;
    }
    MatchData md(r);
    int64_t off = 0;
    auto put = [&](int32_t b, int32_t e, bool skip) -> bool {
        if (off >= cap) return false;
        out_begins[off] = b;
        out_ends[off] = e;
        if (out_skips) out_skips[off] = skip ? 1 : 0;
        ++off;
        return true;
    };
    for (int64_t row = 0; row < B; ++row) {
        out_rb[row] = int32_t(off);
        for (int32_t col = rb[row]; col < re[row]; ++col) {
            const int32_t sb = begins[col];
            const uint8_t* s = chars + sb;
            const size_t len = size_t(ends[col] - sb);
            if (skips && skips[col]) {  // regex_split.cpp:231-234
                if (!put(sb, ends[col], true)) return fail(ORC_E_CAPACITY, "regex_split: output overflow");
                continue;
            }
            // regex_split.cpp:240-284.  `last_begin` is a size_t starting at SIZE_MAX; add_split
            // takes int begin/end, so SIZE_MAX travels as -1 and is clamped to 0 on output.
            size_t start = 0, last_begin = size_t(-1);
            uint32_t num_splits = 0;
            bool overflow = false;
            auto add_split = [&](int b, int e, bool flag) {
                switch (r->mode) {
                    case 0: if (flag) return; break;
                    case 1: break;
                    case 2:
                        if (!flag && size_t(e) != len) { last_begin = size_t(b); return; }
                        else if (flag) b = int(last_begin);
                        break;
                    case 3:
                        if (!flag) { if (last_begin != size_t(-1)) b = int(last_begin); }
                        else { last_begin = size_t(b); return; }
                        break;
                }
                b = std::max(0, b);
                e = std::min(int(len), e);
                if (num_splits == uint32_t(r->max_splits)) e = int(len);  // uint32 vs int compare (:278)
                if (!put(sb + b, sb + e, false)) overflow = true;
                ++num_splits;
            };
            size_t mb, me;
            while (next_match(r, md, s, len, start, mb, me)) {  // :286-301
                if (mb != start) add_split(int(start), int(mb), r->invert);
                add_split(int(mb), int(me), !r->invert);
                start = me;
            }
            if (start < len) add_split(int(start), int(len), r->invert);            // :302-304
            else if (r->mode == 3 && last_begin != len) add_split(int(last_begin), int(len), r->invert);  // :305-309
            if (overflow) return fail(ORC_E_CAPACITY, "regex_split: output overflow");
        }
        out_re[row] = int32_t(off);
    }
    *n_rows_out = B;
    *n_out = off;
    return ORC_OK;
}

// =======================================================================================
// SpecialTokensSplit : src/special_tokens_split.cpp:61-162, PCRE2Wrapper::match_and_find_group src/utils.cpp:423-461
// =======================================================================================
extern "C" int orc_special_tokens_split_run(const orc_regex* r, const int32_t* rb, const int32_t* re, int64_t B,
                                            const int32_t* begins, const int32_t* ends, const uint8_t* chars,
                                            const uint8_t* skips, int32_t* out_rb, int32_t* out_re,
                                            int32_t* out_begins, int32_t* out_ends, uint8_t* out_skips, int64_t cap,
                                            int64_t* n_out) {
    const Pcre2& P = Pcre2::get();
    MatchData md(r);
    int64_t off = 0;
    auto put = [&](int32_t b, int32_t e, bool skip) -> bool {
        if (off >= cap) return false;
        out_begins[off] = b;
        out_ends[off] = e;
        out_skips[off] = skip ? 1 : 0;
        ++off;
        return true;
    };
    for (int64_t row = 0; row < B; ++row) {
        out_rb[row] = int32_t(off);
        for (int32_t col = rb[row]; col < re[row]; ++col) {
            const int32_t sb = begins[col];
            const uint8_t* s = chars + sb;
            const size_t len = size_t(ends[col] - sb);
            bool ok = true;
            if (skips && skips[col]) {  // :110-113
                ok = put(sb, ends[col], true);
            } else {
                size_t start = 0;
                while (r->code && md.md) {  // :127-141
                    const int rc = (r->jit ? P.jit_match : P.match)(r->code, s, len, start, 0, md.md, nullptr);
                    if (rc < 0) break;
                    const size_t* ov = P.ovector(md.md);
                    const size_t mb = ov[0], me = ov[1];
                    if (mb == me) break;  // "match.first != match.second" (:119)
                    // utils.cpp:452-457: the capture group that lies inside the full match (the last such one)
                    size_t gb = size_t(-1), ge = size_t(-1);
                    const uint32_t n = P.ovector_count(md.md);
                    for (uint32_t g = 1; g < n; ++g)
                        if (mb <= ov[2 * g] && ov[2 * g] <= me && ov[2 * g + 1] <= me) { gb = ov[2 * g]; ge = ov[2 * g + 1]; }
                    const bool empty_group = gb == size_t(-1) || gb == ge;
                    const size_t group_start = empty_group ? mb : gb;
                    const size_t group_end = (ge == size_t(-1) || empty_group) ? me : ge;
                    if (start < mb) ok = ok && put(sb + int32_t(start), sb + int32_t(mb), false);
                    ok = ok && put(sb + int32_t(group_start), sb + int32_t(group_end), true);
                    start = me;
                }
                if (start < len) ok = ok && put(sb + int32_t(start), sb + int32_t(len), false);  // :143-147
            }
            if (!ok) return fail(ORC_E_CAPACITY, "special_tokens_split: output overflow");
        }
        out_re[row] = int32_t(off);
    }
    *n_out = off;
    return ORC_OK;
}

// =======================================================================================
// BPETokenizer
// =======================================================================================
namespace {

// src/bpe_tokenizer.hpp:40-115: open addressing, linear probing, capacity = next pow2 >=
// n/0.7+1 (min 8), hash = (key * 0x9E3779B97F4A7C15) >> 32, re-insert overwrites.
struct MergeMap {
    struct Slot { uint64_t key = 0; int32_t rank = 0, new_id = 0; bool used = false; };
    std::vector<Slot> slots;
    size_t mask = 0;
    static uint64_t pack(int32_t l, int32_t r) { return (uint64_t(uint32_t(l)) << 32) | uint32_t(r); }
    static size_t hash(uint64_t k) { return size_t((k * 0x9E3779B97F4A7C15ULL) >> 32); }
    void reserve(size_t n) {
        size_t need = size_t(double(n) / 0.7) + 1, capn = 1;
        while (capn < need) capn <<= 1;
        if (capn < 8) capn = 8;
        slots.assign(capn, Slot{});
        mask = capn - 1;
    }
    void put(int32_t l, int32_t r, int32_t rank, int32_t nid) {
        uint64_t k = pack(l, r);
        size_t i = hash(k) & mask;
        while (slots[i].used && slots[i].key != k) i = (i + 1) & mask;
        slots[i] = Slot{k, rank, nid, true};
    }
    const Slot* find(int32_t l, int32_t r) const {
        if (slots.empty()) return nullptr;
        uint64_t k = pack(l, r);
        size_t i = hash(k) & mask;
        while (slots[i].used) {
            if (slots[i].key == k) return &slots[i];
            i = (i + 1) & mask;
        }
        return nullptr;
    }
};

using QEntry = std::tuple<int32_t, int32_t, int32_t, int32_t, int32_t>;  // rank,new_id,a,b,seq (hpp:131)
struct ByRankThenSeq {  // bpe_tokenizer.cpp:166-172
    bool operator()(const QEntry& x, const QEntry& y) const {
        return std::get<0>(x) != std::get<0>(y) ? std::get<0>(x) > std::get<0>(y)
                                               : std::get<4>(x) > std::get<4>(y);
    }
};
struct Sym { int32_t id, prev, next; bool alive; };
struct ReusableQueue : std::priority_queue<QEntry, std::vector<QEntry>, ByRankThenSeq> {
    using Base = std::priority_queue<QEntry, std::vector<QEntry>, ByRankThenSeq>;
    explicit ReusableQueue(std::vector<QEntry>&& st) : Base(ByRankThenSeq{}, std::move(st)) {}
    std::vector<QEntry>&& release() { return std::move(this->c); }
};

}  // namespace

struct orc_bpe {
    std::unordered_map<std::string, unsigned> vocab;  // after erasing merged strings
    MergeMap merges;
    TrieNode trie;
    std::string end_suffix;
    bool byte_fallback = false, fuse_unk = false;
    int32_t unk_id = -1;
    size_t cache_cap = 0;
    std::shared_mutex mu;
    std::unordered_map<std::string, std::vector<int32_t>> cache;
    int64_t tie_events = 0;

    // bpe_tokenizer.cpp:196-339
    void tokenize(std::string_view piece, std::vector<int32_t>& out, std::vector<Sym>& syms,
                  std::vector<QEntry>& qstore) {
        std::string key(piece);
        {
            std::shared_lock<std::shared_mutex> lk(mu);
            auto it = cache.find(key);
            if (it != cache.end()) {
                out.insert(out.end(), it->second.begin(), it->second.end());
                return;
            }
        }
        std::string with_suffix;
        std::string_view text = piece;
        if (!end_suffix.empty()) {
            with_suffix.reserve(piece.size() + end_suffix.size());
            with_suffix.append(piece).append(end_suffix);
            text = with_suffix;
        }
        const uint8_t* s = reinterpret_cast<const uint8_t*>(text.data());
        const int n = int(text.size());
        syms.clear();
        auto append = [&](int32_t id) {
            int32_t i = int32_t(syms.size());
            if (i > 0) syms[i - 1].next = i;
            syms.push_back(Sym{id, i - 1, -1, true});
        };
        for (int idx = 0; idx < n;) {  // :230-257
            int t = trie.find_longest(s, n, idx);
            if (t != -1) { append(t); continue; }
            int32_t fb = -1;
            if (byte_fallback) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "<0x%02X>", unsigned(s[idx]));
                auto it = vocab.find(buf);
                if (it != vocab.end()) fb = int32_t(it->second);
            }
            if (fb != -1) append(fb);
            else if (unk_id != -1 && (!fuse_unk || syms.empty() || syms.back().id != -1)) append(unk_id);
            ++idx;
        }
        const size_t n_init = syms.size(), out0 = out.size();
        size_t live = syms.size();
        qstore.clear();
        qstore.reserve(syms.size());
        ReusableQueue pq(std::move(qstore));  // bpe_tokenizer.cpp:174-183: storage reused across pieces
        int32_t seq = 0;
        auto try_push = [&](int32_t a, int32_t b) -> int32_t {
            const MergeMap::Slot* m = merges.find(syms[a].id, syms[b].id);
            if (!m) return -1;
            pq.emplace(m->rank, m->new_id, a, b, seq);
            return m->rank;
        };
        int32_t head = syms.empty() ? -1 : 0;
        for (int32_t a = head; a != -1 && syms[a].next != -1; a = syms[a].next) {  // :281-285
            try_push(a, syms[a].next);
            ++seq;
        }
        while (!pq.empty() && live >= 2) {  // :287-323
            auto [rank, nid, a, b, sq] = pq.top();
            (void)rank; (void)sq;
            pq.pop();
            if (!syms[a].alive || !syms[b].alive || syms[a].next != b) continue;
            const int32_t pv = syms[a].prev, nx = syms[b].next, m = int32_t(syms.size());
            syms.push_back(Sym{nid, pv, nx, true});
            syms[a].alive = syms[b].alive = false;
            if (pv != -1) syms[pv].next = m; else head = m;
            if (nx != -1) syms[nx].prev = m;
            --live;
            ++seq;
            int32_t r1 = pv != -1 ? try_push(pv, m) : -1;
            int32_t r2 = nx != -1 ? try_push(m, nx) : -1;
            if (r1 != -1 && r1 == r2) ++tie_events;
        }
        for (int32_t i = head; i != -1; i = syms[i].next) out.push_back(syms[i].id);
        qstore = pq.release();
        {
            std::unique_lock<std::shared_mutex> lk(mu);  // :331-338 insert-only, never evicts
            if (cache.size() < cache_cap && n_init > 0)
                cache.emplace(std::move(key), std::vector<int32_t>(out.begin() + out0, out.end()));
        }
    }
};

extern "C" int orc_bpe_create(const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                              const int32_t* ml_begins, const int32_t* ml_ends, const uint8_t* ml_chars,
                              const int32_t* mr_begins, const int32_t* mr_ends, const uint8_t* mr_chars, int64_t M,
                              const int32_t* a_begins, const int32_t* a_ends, const uint8_t* a_chars,
                              const int32_t* a_ids, int64_t A,
                              const char* unk_token, int64_t unk_len, int fuse_unk,
                              const char* suffix_indicator, int64_t si_len,
                              const char* end_suffix, int64_t es_len,
                              int byte_fallback, int64_t cache_capacity, orc_bpe** out) {
    (void)suffix_indicator; (void)si_len;  // stored but never read by the reference (A.2-T1)
    auto t = std::make_unique<orc_bpe>();
    // bpe_tokenizer.cpp:51-67: added tokens into an ordered map, first occurrence kept.
    std::map<std::string, int32_t> added;
    for (int64_t i = 0; i < A; ++i) added.insert({str_of(a_chars, a_begins[i], a_ends[i]), a_ids[i]});
    // :71-82: later id wins for duplicate token strings.
    std::unordered_map<std::string, unsigned> vocab;
    vocab.reserve(size_t(V + A));
    for (int64_t id = 0; id < V; ++id) vocab.insert_or_assign(str_of(v_chars, v_begins[id], v_ends[id]), unsigned(id));
    // :84-107
    std::vector<std::pair<std::string, std::string>> merges;
    merges.reserve(size_t(M));
    for (int64_t i = 0; i < M; ++i) {
        if (!mr_begins) {
            std::string line = str_of(ml_chars, ml_begins[i], ml_ends[i]);
            size_t sp = line.find(' ');
            merges.emplace_back(line.substr(0, sp), line.substr(sp + 1));  // npos+1 == 0, as in the reference
        } else {
            merges.emplace_back(str_of(ml_chars, ml_begins[i], ml_ends[i]), str_of(mr_chars, mr_begins[i], mr_ends[i]));
        }
    }
    for (const auto& kv : added) vocab.insert(kv);  // :110-114 no overwrite
    // ctor, bpe_tokenizer.cpp:341-388
    std::string unk(unk_token, unk_token + unk_len);
    if (auto it = vocab.find(unk); it != vocab.end()) t->unk_id = int32_t(it->second);
    t->merges.reserve(merges.size());
    std::vector<std::string> merged;
    merged.reserve(merges.size());
    for (size_t i = 0; i < merges.size(); ++i) {
        auto l = vocab.find(merges[i].first), r = vocab.find(merges[i].second);
        std::string both = merges[i].first + merges[i].second;
        auto z = vocab.find(both);
        if (l == vocab.end() || r == vocab.end() || z == vocab.end())
            return fail(ORC_E_VOCAB, "merge " + std::to_string(i) + " references a token missing from the vocab");
        t->merges.put(int32_t(l->second), int32_t(r->second), int32_t(i), int32_t(z->second));
        merged.push_back(std::move(both));
    }
    for (const auto& m : merged) vocab.erase(m);
    for (const auto& kv : vocab)
        t->trie.add(reinterpret_cast<const uint8_t*>(kv.first.data()), kv.first.size(), int(kv.second));
    t->vocab = std::move(vocab);
    t->end_suffix.assign(end_suffix, end_suffix + es_len);
    t->byte_fallback = byte_fallback != 0;
    t->fuse_unk = fuse_unk != 0;
    t->cache_cap = size_t(cache_capacity);
    t->cache.reserve(t->cache_cap);
    *out = t.release();
    return ORC_OK;
}

extern "C" void orc_bpe_destroy(orc_bpe* t) { delete t; }
extern "C" int64_t orc_bpe_tie_events(const orc_bpe* t) { return t->tie_events; }
extern "C" void orc_bpe_clear_cache(orc_bpe* t) {
    std::unique_lock<std::shared_mutex> lk(t->mu);
    t->cache.clear();
    t->tie_events = 0;
}

// bpe_tokenizer.cpp:122-163
extern "C" int orc_bpe_run(orc_bpe* t, const int32_t* rb, const int32_t* re, int64_t B,
                           const int32_t* begins, const int32_t* ends, const uint8_t* chars,
                           int32_t* out_begins, int32_t* out_ends, int32_t* out_ids, int64_t cap, int64_t* n_ids) {
    int64_t off = 0;
    std::vector<int32_t> buf;
    std::vector<Sym> syms;
    std::vector<QEntry> qstore;
    buf.reserve(256);
    for (int64_t row = 0; row < B; ++row) {
        out_begins[row] = int32_t(off);
        for (int32_t col = rb[row]; col < re[row]; ++col) {
            buf.clear();
            t->tokenize(std::string_view(reinterpret_cast<const char*>(chars) + begins[col], size_t(ends[col] - begins[col])),
                        buf, syms, qstore);
            for (int32_t id : buf) {
                if (off >= cap) return fail(ORC_E_CAPACITY, "bpe: ids overflow the output buffer");
                out_ids[off++] = id;
            }
        }
        out_ends[row] = int32_t(off);
    }
    *n_ids = off;
    return ORC_OK;
}

// =======================================================================================
// WordpieceTokenizer
// =======================================================================================
struct orc_wordpiece {
    TrieNode root, sub;
    int max_bytes = 100;
};

extern "C" int orc_wordpiece_create(const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                                    const char* suffix_indicator, int64_t si_len, int max_bytes_per_word,
                                    orc_wordpiece** out) {
    auto w = std::make_unique<orc_wordpiece>();
    std::string si(suffix_indicator, suffix_indicator + si_len);
    for (int64_t id = 0; id < V; ++id) {  // wordpiece_tokenizer.cpp:59-71
        std::string word = str_of(v_chars, v_begins[id], v_ends[id]);
        const uint8_t* p = reinterpret_cast<const uint8_t*>(word.data());
        if (word.substr(0, si.size()) == si) w->sub.add(p + si.size(), word.size() - si.size(), int(id));
        else w->root.add(p, word.size(), int(id));
    }
    w->max_bytes = max_bytes_per_word;
    *out = w.release();
    return ORC_OK;
}
extern "C" void orc_wordpiece_destroy(orc_wordpiece* w) { delete w; }

// wordpiece_tokenizer.cpp:74-131.  An empty word is undefined in the reference (it reads
// str[0], utils.cpp:521); the oracle refuses it instead of guessing.
extern "C" int orc_wordpiece_run(const orc_wordpiece* w, const int32_t* rb, const int32_t* re, int64_t B,
                                 const int32_t* begins, const int32_t* ends, const uint8_t* chars, int32_t unk_id,
                                 int32_t* out_begins, int32_t* out_ends, int32_t* out_ids, int64_t cap, int64_t* n_ids) {
    int64_t off = 0;
    auto put = [&](int32_t v) -> bool {
        if (off >= cap) return false;
        out_ids[off++] = v;
        return true;
    };
    for (int64_t row = 0; row < B; ++row) {
        out_begins[row] = int32_t(off);
        for (int32_t col = rb[row]; col < re[row]; ++col) {
            const int len = ends[col] - begins[col];
            if (len > w->max_bytes) {  // strict >, :100-103
                if (!put(unk_id)) return fail(ORC_E_CAPACITY, "wordpiece: ids overflow");
                continue;
            }
            if (len <= 0) return fail(ORC_E_RANGE, "wordpiece: empty word is undefined in the reference");
            const uint8_t* s = chars + begins[col];
            int idx = 0;
            int tok = w->root.find_longest(s, len, idx);
            const int64_t first = off;
            if (!put(tok == -1 ? unk_id : tok)) return fail(ORC_E_CAPACITY, "wordpiece: ids overflow");
            if (tok == -1) continue;
            while (idx < len) {
                tok = w->sub.find_longest(s, len, idx);
                if (tok == -1) {  // :118-123 the whole word collapses to one unk
                    out_ids[first] = unk_id;
                    off = first + 1;
                    break;
                }
                if (!put(tok)) return fail(ORC_E_CAPACITY, "wordpiece: ids overflow");
            }
        }
        out_ends[row] = int32_t(off);
    }
    *n_ids = off;
    return ORC_OK;
}

// =======================================================================================
// VocabEncoder
// =======================================================================================
struct orc_vocab_encoder {
    std::unordered_map<std::string, int64_t> map;  // "absl::flat_hash_map" is std::unordered_map here (bpe_tokenizer.hpp:33-36)
    int elem_size = 4;
};

extern "C" int orc_vocab_encoder_create(const int32_t* k_begins, const int32_t* k_ends, const uint8_t* k_chars,
                                        const void* values, int64_t V, int elem_size, orc_vocab_encoder** out) {
    if (elem_size != 4 && elem_size != 8) return fail(ORC_E_ARG, "VocabEncoder values must be i32 or i64");
    auto e = std::make_unique<orc_vocab_encoder>();
    e->elem_size = elem_size;
    for (int64_t i = 0; i < V; ++i) {  // vocab_encoder.cpp:74-77: insert => first duplicate wins
        int64_t v = elem_size == 4 ? int64_t(static_cast<const int32_t*>(values)[i]) : static_cast<const int64_t*>(values)[i];
        e->map.insert({str_of(k_chars, k_begins[i], k_ends[i]), v});
    }
    *out = e.release();
    return ORC_OK;
}
extern "C" void orc_vocab_encoder_destroy(orc_vocab_encoder* e) { delete e; }

extern "C" int orc_vocab_encoder_run(const orc_vocab_encoder* e, const int32_t* begins, const int32_t* ends,
                                     const uint8_t* chars, int64_t N, const void* default_value, void* out) {
    for (int64_t i = 0; i < N; ++i) {  // vocab_encoder.cpp:88-91
        auto it = e->map.find(str_of(chars, begins[i], ends[i]));
        if (e->elem_size == 4)
            static_cast<int32_t*>(out)[i] = it == e->map.end() ? *static_cast<const int32_t*>(default_value) : int32_t(it->second);
        else
            static_cast<int64_t*>(out)[i] = it == e->map.end() ? *static_cast<const int64_t*>(default_value) : it->second;
    }
    return ORC_OK;
}

// =======================================================================================
// RaggedToDense
// =======================================================================================
extern "C" int orc_ragged_to_dense(const int32_t* begins, const int32_t* ends, int64_t B,
                                   const void* data, int64_t n_data, int elem_size, int64_t inner,
                                   int32_t target_dim, const void* default_value,
                                   int pad_right, int pad_max_length, void* out_dense, uint8_t* out_mask) {
    const char* src = static_cast<const char*>(data);
    const char* dflt = static_cast<const char*>(default_value);
    char* dst = static_cast<char*>(out_dense);
    uint8_t* msk = out_mask;
    const size_t T = size_t(target_dim), cell = size_t(elem_size) * size_t(inner);
    auto fill = [&](size_t count) {
        for (size_t j = 0; j < count * size_t(inner); ++j) { std::memcpy(dst, dflt, size_t(elem_size)); dst += elem_size; }
    };
    for (int64_t i = 0; i < B; ++i) {
        const size_t len = size_t(ends[i] - begins[i]);
        // ragged_to_dense.cpp:132-133 / 152-153: with pad_max_length the copy is T long whatever the row holds.
        const size_t take = pad_max_length ? T : std::min(len, T);
        if (int64_t(begins[i]) + int64_t(take) > n_data || begins[i] < 0)
            return fail(ORC_E_RANGE, "ragged_to_dense: row reads past the data tensor");
        const size_t pad = T - take;
        if (!pad_right) {  // :149-166
            fill(pad);
            if (msk) { std::memset(msk, 0, pad * size_t(inner)); msk += pad * size_t(inner); }
        }
        std::memcpy(dst, src + cell * size_t(begins[i]), cell * take);
        dst += cell * take;
        if (msk) { std::memset(msk, 1, take * size_t(inner)); msk += take * size_t(inner); }
        if (pad_right) {  // :130-147
            fill(pad);
            if (msk) { std::memset(msk, 0, pad * size_t(inner)); msk += pad * size_t(inner); }
        }
    }
    return ORC_OK;
}

// =======================================================================================
// TrieTokenizer : src/trie_tokenizer.cpp:23-81 (RWKV).  Greedy longest match from a trie of vocab[i] -> indices[i];
// where nothing matches the reference's loop never advances (:72-75) -- reported as ORC_E_VOCAB here.
// =======================================================================================
struct orc_trie_tokenizer { TrieNode trie; };

extern "C" int orc_trie_tokenizer_create(const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                                         const int32_t* indices, orc_trie_tokenizer** out) {
    auto t = std::make_unique<orc_trie_tokenizer>();
    for (int64_t i = 0; i < V; ++i) t->trie.add(v_chars + v_begins[i], size_t(v_ends[i] - v_begins[i]), indices[i]);
    *out = t.release();
    return ORC_OK;
}
extern "C" void orc_trie_tokenizer_destroy(orc_trie_tokenizer* t) { delete t; }

extern "C" int orc_trie_tokenizer_run(const orc_trie_tokenizer* t, const int32_t* rb, const int32_t* re, int64_t B,
                                      const int32_t* begins, const int32_t* ends, const uint8_t* chars, int32_t* out_begins,
                                      int32_t* out_ends, int32_t* out_ids, int64_t cap, int64_t* n_ids) {
    int64_t off = 0;
    for (int64_t row = 0; row < B; ++row) {
        out_begins[row] = int32_t(off);
        for (int32_t col = rb[row]; col < re[row]; ++col) {
            const uint8_t* s = chars + begins[col];
            const int n = ends[col] - begins[col];
            int idx = 0;
            while (idx < n) {
                const int before = idx;
                const int tok = t->trie.find_longest(s, n, idx);
                if (idx == before) return fail(ORC_E_VOCAB, "trie tokenizer: no vocabulary entry matches (the reference loops forever)");
                if (off >= cap) return fail(ORC_E_CAPACITY, "trie tokenizer: ids overflow");
                out_ids[off++] = tok;
            }
        }
        out_ends[row] = int32_t(off);
    }
    *n_ids = off;
    return ORC_OK;
}

// =======================================================================================
// UTF8Validate : src/utf8_validate.cpp:18-143.  A byte-at-a-time automaton: `pending` continuation bytes
// are still owed to a symbol of `width` bytes whose code point is being assembled in `cp`.
// Output offsets start at begins[0] like the reference's `out_idx` (:46).
// =======================================================================================
extern "C" int orc_utf8_validate(const int32_t* begins, const int32_t* ends, const uint8_t* chars, int64_t n,
                                 int replace_mode, int32_t* out_begins, int32_t* out_ends, uint8_t* out_chars,
                                 int64_t cap, int64_t* n_chars_out) {
    static const uint8_t kRepl[3] = {0xEF, 0xBF, 0xBD};
    static const uint32_t kMinCp[4] = {0x0, 0x80, 0x800, 0x10000};
    int64_t o = n ? begins[0] : 0;
    const int64_t o0 = o;
    bool overflow = false;
    auto put = [&](const uint8_t* src, int k) {
        if (o + k > cap) { overflow = true; return; }
        std::memcpy(out_chars + o, src, size_t(k));
        o += k;
    };
    auto bad = [&](int times) { if (replace_mode) for (int t = 0; t < times; ++t) put(kRepl, 3); };
    for (int64_t i = 0; i < n; ++i) {
        uint32_t pending = 0, width = 0, cp = 0;
        out_begins[i] = int32_t(o);
        for (int64_t j = begins[i]; j < ends[i]; ++j) {
            const uint8_t c = chars[j];
            if (pending == 0) {
                if (c < 0x80) put(&c, 1);
                else if ((c >> 5) == 0x6) { width = 2; pending = 1; cp = uint32_t(c & 0x1F) << 6; }
                else if ((c >> 4) == 0xE) { width = 3; pending = 2; cp = uint32_t(c & 0x0F) << 12; }
                else if ((c >> 3) == 0x1E) { width = 4; pending = 3; cp = uint32_t(c & 0x07) << 18; }
                else bad(1);
                continue;
            }
            if ((c >> 6) != 0x2) {  // not a continuation: the symbol is broken, this byte starts over (:93-104)
                pending = 0;
                bad(1);
                --j;
                continue;
            }
            --pending;
            cp |= uint32_t(c & 0x3F) << (6 * pending);
            if (pending) continue;
            if (cp < kMinCp[width - 1]) bad(int(width));  // overlong form: one replacement per byte (:111-121)
            else put(chars + j + 1 - width, int(width));
        }
        if (pending) bad(1);  // unfinished symbol at the end of the string (:134-137)
        out_ends[i] = int32_t(o);
        if (overflow) return fail(ORC_E_CAPACITY, "utf8_validate: output overflow");
    }
    *n_chars_out = o - o0;
    return ORC_OK;
}

// =======================================================================================
// Truncate : src/truncate.cpp:37-150 (begins/ends are modified in place there; here in/out arrays)
// =======================================================================================
extern "C" int orc_truncate(int n_inputs, int32_t* b0, int32_t* e0, int32_t* b1, int32_t* e1, int64_t n,
                            int32_t max_length, const char* side, const char* mode) {
    const std::string trunc_side(side), trunc_mode(mode ? mode : "");
    if (trunc_side != "left" && trunc_side != "right") return fail(ORC_E_ARG, "Unknown truncation side: " + trunc_side);
    if (n_inputs == 1) {
        for (int64_t i = 0; i < n; ++i) {
            const int32_t t = std::min(e0[i] - b0[i], max_length);
            if (trunc_side == "right") e0[i] = b0[i] + t; else b0[i] = e0[i] - t;
        }
        return ORC_OK;
    }
    if (n_inputs != 2) return fail(ORC_E_ARG, "Only single or pair inputs are supported in Truncation op");
    if (trunc_mode != "only_first" && trunc_mode != "only_second" && trunc_mode != "longest_first")
        return fail(ORC_E_ARG, "Unknown truncation mode: " + trunc_mode);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t fl = e0[i] - b0[i], sl = e1[i] - b1[i];
        if (fl + sl <= max_length) continue;
        const int32_t fr = (max_length % 2) * (fl >= sl), sr = (max_length % 2) * (fl < sl);
        const int32_t half = max_length / 2, half_up = max_length / 2 + max_length % 2;
        if (trunc_side == "right") {
            if (trunc_mode == "only_first") { if (fl > max_length) e0[i] = b0[i] + max_length; }
            else if (trunc_mode == "only_second") { if (sl > max_length) e1[i] = b1[i] + max_length; }
            else if (fl >= half_up && sl <= half) e0[i] = b0[i] + (max_length - sl);
            else if (fl < half_up && sl > half) e1[i] = b1[i] + (max_length - fl);
            else { e0[i] = b0[i] + half + fr; e1[i] = b1[i] + half + sr; }
        } else {
            if (trunc_mode == "only_first") { if (fl > max_length) b0[i] = e0[i] - max_length; }
            else if (trunc_mode == "only_second") { if (sl > max_length) b1[i] = e1[i] - max_length; }
            else if (fl >= half_up && sl <= half) b0[i] = e0[i] - (max_length - sl);
            else if (fl < half_up && sl > half) b1[i] = e1[i] - (max_length - fl);
            else { b0[i] = e0[i] - (half + fr); b1[i] = e1[i] - (half + sr); }
        }
    }
    return ORC_OK;
}

// =======================================================================================
// CombineSegments : src/combine_segments.cpp:36-134 (i32 elements and ids)
// =======================================================================================
extern "C" int orc_combine_segments(int n_segs, const int32_t* const* begins, const int32_t* const* ends,
                                    const int32_t* const* data, const int64_t* n_rows, const int32_t* seg_ids,
                                    int64_t max_rows, int32_t* out_begins, int32_t* out_ends, int32_t* out_data,
                                    int32_t* out_ids, int64_t cap, int64_t* n_out) {
    int64_t off = 0;
    for (int64_t i = 0; i < max_rows; ++i) {
        out_begins[i] = int32_t(off);
        for (int j = 0; j < n_segs; ++j) {
            const int64_t r = n_rows[j] == 1 ? 0 : i;  // a single-row segment is broadcast (:110-116)
            const int32_t b = begins[j][r], len = ends[j][r] - b;
            if (off + len > cap) return fail(ORC_E_CAPACITY, "combine_segments: output overflow");
            for (int32_t k = 0; k < len; ++k) {
                out_data[off + k] = data[j][b + k];
                out_ids[off + k] = seg_ids[j];
            }
            off += len > 0 ? len : 0;
        }
        out_ends[i] = int32_t(off);
    }
    *n_out = off;
    return ORC_OK;
}

// =======================================================================================
// VocabDecoder / ByteFallback / FuzeRagged
// =======================================================================================
extern "C" int orc_vocab_decoder(const int32_t* ids, int64_t B, int64_t S,
                                 const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                                 const int32_t* skip, int64_t n_skip,
                                 int32_t* out_rb, int32_t* out_re, int32_t* out_begins, int32_t* out_ends,
                                 uint8_t* out_chars, int64_t chars_cap, int64_t* n_chars) {
    const int64_t Sp = S > 0 ? S : 1;  // vocab_decoder.cpp:45-46,58-59
    int64_t off = 0;
    for (int64_t b = 0; b < B; ++b) {
        out_rb[b] = int32_t(b * Sp);
        out_re[b] = int32_t(b * Sp + Sp);
        if (S == 0) {  // :61-65
            out_begins[b] = int32_t(off);
            out_ends[b] = int32_t(off);
            continue;
        }
        for (int64_t k = b * Sp; k < b * Sp + Sp; ++k) {
            const int32_t id = ids[k];
            out_begins[k] = int32_t(off);
            // :70-73 -- `token_id < vocab_size` compares as size_t, so negative ids are out of range
            if (size_t(int64_t(id)) < size_t(V) && std::find(skip, skip + n_skip, id) == skip + n_skip) {
                const int64_t n = v_ends[id] - v_begins[id];
                if (off + n > chars_cap) return fail(ORC_E_CAPACITY, "vocab_decoder: chars overflow");
                std::memcpy(out_chars + off, v_chars + v_begins[id], size_t(n));
                off += n;
            }
            out_ends[k] = int32_t(off);
        }
    }
    *n_chars = off;
    return ORC_OK;
}

extern "C" int orc_byte_fallback(const int32_t* begins, const int32_t* ends, const uint8_t* chars, int64_t N,
                                 int32_t* out_begins, int32_t* out_ends, uint8_t* out_chars, int64_t* n_chars) {
    // sentence_piece.cpp:27-46: the 256 strings "<0x%02X>" (upper-case hex) map to their byte.
    static const std::unordered_map<std::string, uint8_t> piece_to_byte = [] {
        std::unordered_map<std::string, uint8_t> m;
        char buf[8];
        for (int i = 0; i < 256; ++i) { std::snprintf(buf, sizeof buf, "<0x%02X>", i); m[buf] = uint8_t(i); }
        return m;
    }();
    uint32_t off = 0;
    for (int64_t i = 0; i < N; ++i) {
        out_begins[i] = int32_t(off);
        std::string tok = str_of(chars, begins[i], ends[i]);
        if (tok.size() == 6 && tok.rfind('<') == 0 && tok.rfind('>') == 5) {  // byte_fallback.cpp:37
            auto it = piece_to_byte.find(tok);
            int ch = it == piece_to_byte.end() ? -1 : int(it->second);
            out_chars[off++] = uint8_t(ch);  // -1 lands as 0xFF (:39-40)
        } else {
            std::memcpy(out_chars + off, tok.data(), tok.size());
            off += uint32_t(tok.size());
        }
        out_ends[i] = int32_t(off);
    }
    *n_chars = off;
    return ORC_OK;
}

extern "C" int orc_fuze(const int32_t* rb, const int32_t* re, int64_t B,
                        const int32_t* begins, const int32_t* ends, int64_t N,
                        int32_t* out_begins, int32_t* out_ends) {
    for (int64_t r = 0; r < B; ++r) {  // fuze.cpp:35-38
        const int64_t bi = rb[r], ei = re[r] > rb[r] ? re[r] - 1 : re[r];
        if (bi < 0 || bi >= N || ei < 0 || ei >= N)
            return fail(ORC_E_RANGE, "fuze: row indexes past begins/ends (undefined in the reference)");
        out_begins[r] = begins[bi];
        out_ends[r] = ends[ei];
    }
    return ORC_OK;
}
