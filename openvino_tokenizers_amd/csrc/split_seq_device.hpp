// split_seq_device.hpp -- matchers that walk a string position by position, ONE LANE PER ROW: SpecialTokensSplit.
// (The Llama-3 pattern used to live here in the same form; it is a bit-parallel scanner now, split_device.hpp, with its
// literal matcher llama3_match_end() kept there as the fallback.)
#pragma once

#include "device_common.hpp"
#include "split_device.hpp"

namespace ovtk {

// ---------------------------------------------------------------------------------------------
// SpecialTokensSplit (src/special_tokens_split.cpp:61-162).  Its pattern is generated
// (tokenizer_pipeline.py:138-159): alternatives "(?:\s*)?(tok|tok|...)(?:\s*)?", one per (strip_left, strip_right)
// class, the tokens quoted literally.  create() parses it back into token lists; the matcher below reproduces
// PCRE2's leftmost-first search: start positions left to right, groups in pattern order, inside a group the
// greedy \s* gives characters back from the right, tokens in listed order.
// ---------------------------------------------------------------------------------------------
struct SpecialDev {
    const int32_t* tok_begins;   // tokens in pattern order
    const int32_t* tok_ends;
    const uint8_t* tok_chars;
    const int32_t* group_first;  // [n_groups + 1]: tokens of group g = [group_first[g], group_first[g + 1])
    const uint8_t* group_flags;  // bit 0 strip_left, bit 1 strip_right
    int32_t n_groups;
    int32_t n_tokens, n_tok_chars;   // sizes of the token lists / of tok_chars (special_sparse_kernel keeps small lists in LDS)
    uint32_t first_bytes[8];     // bytes a match can start with (token first bytes; whitespace lead bytes if any strip_left)
    // the same two sets as lists, for the 16-bytes-at-a-time filters below (round 5); n < 0: too many distinct bytes, no filter
    int32_t n_tok_first;         // distinct first bytes of the tokens: a string without any of them holds no match at all
    uint8_t tok_first[8];
    int32_t n_first;             // members of first_bytes: the positions a match can start at
    uint8_t first[16];
    int32_t tok_padded;          // tok_chars holds every token padded with zeros to a multiple of 16 bytes (token_at reads 16 at a time)
    int32_t any_strip_left;      // some group strips white space on its left: a match may begin in front of its token
    SplitDev uc;                 // Unicode tables (whitespace class)
};

// The token lists by pointer: SpecialDev's own (global memory), or a block's copy of them in LDS (special_sparse_kernel).  Apart from
// SpecialDev so that a kernel can swap them without copying the struct (its byte lists are indexed dynamically: a copy lives in scratch).
struct SpecialTabs {
    const int32_t* tok_begins;
    const int32_t* tok_ends;
    const uint8_t* tok_chars;
    const int32_t* group_first;
    const uint8_t* group_flags;
};
__device__ __forceinline__ SpecialTabs special_tabs(const SpecialDev& T) {
    return SpecialTabs{T.tok_begins, T.tok_ends, T.tok_chars, T.group_first, T.group_flags};
}

// ---- 16 bytes at a time (a lane walks ITS string: the loads of the 64 lanes go to 64 places, but each brings 16 bytes, and a string
// without a candidate byte -- nearly every string -- costs 32 loads and some packed compares instead of a loop over its 512 bytes with
// a load and a table test per byte: until round 5 that loop was the whole op, one lane per row)
struct __attribute__((packed, aligned(1))) SeqBytes16 { uint32_t d[4]; };
// bit 7 of every byte of v that equals c (exact for any byte values)
__device__ __forceinline__ uint32_t bytes_eq(uint32_t v, uint32_t c) {
    const uint32_t t = v ^ (c * 0x01010101u);
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}
// bit k: byte k of the 16 at p is one of list[0 .. n)
__device__ __forceinline__ uint32_t bytes16_in_list(const uint8_t* p, const uint8_t* list, int n) {
    const SeqBytes16 v = *reinterpret_cast<const SeqBytes16*>(p);
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t f = 0;
        for (int i = 0; i < n; ++i) f |= bytes_eq(v.d[j], list[i]);
        f >>= 7;
        out |= ((f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu) << (4 * j);
    }
    return out;
}
// does s[0, slen) hold one of the tokens' first bytes?
__device__ __forceinline__ bool string_may_hold_token(const SpecialDev& T, const uint8_t* s, int slen) {
    if (T.n_tok_first < 0) return true;
    int p = 0;
    uint32_t any = 0;
    for (; p + 16 <= slen; p += 16) {
        const SeqBytes16 v = *reinterpret_cast<const SeqBytes16*>(s + p);
        for (int i = 0; i < T.n_tok_first; ++i) {
            const uint32_t c = T.tok_first[i] * 0x01010101u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t t = v.d[j] ^ c;
                any |= (t - 0x01010101u) & ~t;   // (bit 7 of some byte is set iff some byte of t is zero)
            }
        }
    }
    if (any & 0x80808080u) return true;
    for (; p < slen; ++p) {
        const uint32_t b = s[p];
        for (int i = 0; i < T.n_tok_first; ++i)
            if (b == T.tok_first[i]) return true;
    }
    return false;
}

__device__ __forceinline__ int space_run_end(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    while (p < slen) {
        const SeqChar c = seq_char(sp, s, p, slen);
        if (c.cls != kClsS) break;
        p += c.len;
    }
    return p;
}
__device__ __forceinline__ bool token_at(const SpecialDev& T, const SpecialTabs& tb, int tok, const uint8_t* s, int slen, int p) {
    const int b = tb.tok_begins[tok], n = tb.tok_ends[tok] - b;
    if (n <= 0 || p + n > slen) return false;
    if (s[p] != tb.tok_chars[b]) return false;   // (nearly every call ends here)
    int k = 0;
    if (T.tok_padded) {   // 16 bytes at a time: the token's text is padded with zeros to a multiple of 16 (create), the string's may be read
                          // as far as it reaches
        for (; k + 16 <= n && p + k + 16 <= slen; k += 16) {
            const SeqBytes16 a = *reinterpret_cast<const SeqBytes16*>(s + p + k), t = *reinterpret_cast<const SeqBytes16*>(tb.tok_chars + b + k);
            if ((a.d[0] ^ t.d[0]) | (a.d[1] ^ t.d[1]) | (a.d[2] ^ t.d[2]) | (a.d[3] ^ t.d[3])) return false;
        }
        if (k < n && p + k + 16 <= slen) {   // the token's last bytes: compared under a mask
            const SeqBytes16 a = *reinterpret_cast<const SeqBytes16*>(s + p + k), t = *reinterpret_cast<const SeqBytes16*>(tb.tok_chars + b + k);
            const int r = n - k;   // 1..15
            uint32_t diff = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int keep = r - 4 * j;
                const uint32_t m = keep >= 4 ? ~0u : (keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u));
                diff |= (a.d[j] ^ t.d[j]) & m;
            }
            return diff == 0u;
        }
    }
    for (; k < n; ++k)
        if (s[p + k] != tb.tok_chars[b + k]) return false;
    return true;
}
// Match at start position p: returns true and the group / match extents.  no_strip_before: strip_left groups are known
// to fail for start positions below it (their candidate token positions shrink as p moves through a whitespace run).
__device__ __forceinline__ bool special_match_at(const SpecialDev& T, const SpecialTabs& tb, const uint8_t* s, int slen, int p, int& no_strip_before,
                                                 int& gb, int& ge, int& me) {
    int fail_until = no_strip_before;
    for (int g = 0; g < T.n_groups; ++g) {
        const uint32_t fl = tb.group_flags[g];
        int hi = p;  // token positions tried: hi, then back through the whitespace run down to p
        if (fl & 1u) {
            if (p < no_strip_before) continue;
            hi = space_run_end(T.uc, s, slen, p);
        }
        for (int q = hi;;) {
            for (int t = tb.group_first[g]; t < tb.group_first[g + 1]; ++t) {
                if (token_at(T, tb, t, s, slen, q)) {
                    gb = q;
                    ge = q + (tb.tok_ends[t] - tb.tok_begins[t]);
                    me = (fl & 2u) ? space_run_end(T.uc, s, slen, ge) : ge;
                    return true;
                }
            }
            if (q <= p) break;
            --q;  // previous character start inside the whitespace run [p, hi)
            while (q > p && (s[q] & 0xC0u) == 0x80u) --q;
        }
        if ((fl & 1u) && hi > p && hi + 1 > fail_until) fail_until = hi + 1;
    }
    no_strip_before = fail_until;  // every strip_left group failed for all of [p, hi]: also for later starts in that run
    return false;
}
// Next match at or after `start`: leftmost start position first.  Returns false when there is none.
// [lo, hi): string positions outside which no token's first byte stands (the caller's sweep; 0 .. slen when it knows nothing).
// Where no group strips on its left a match begins with its token: nothing outside the window is looked at.
__device__ __forceinline__ bool special_next_match(const SpecialDev& T, const SpecialTabs& tb, const uint8_t* s, int slen, int start, int& mb, int& gb,
                                                   int& ge, int& me, int lo = 0, int hi = 0x7FFFFFFF) {
    int no_strip_before = 0;
    int p = start;
    int end = slen;
    if (!T.any_strip_left) {
        if (p < lo) p = lo;
        if (hi < end) end = hi;
    }
    if (T.n_first >= 0) {   // the positions a match can start at, 16 bytes at a time
        for (; p + 16 <= slen && p < end; p += 16) {
            uint32_t cand = bytes16_in_list(s + p, T.first, T.n_first);
            while (cand) {
                const int k = __ffs(cand) - 1;
                cand &= cand - 1u;
                if ((s[p + k] & 0xC0u) == 0x80u) continue;  // matches start at character boundaries (PCRE2_UTF)
                if (special_match_at(T, tb, s, slen, p + k, no_strip_before, gb, ge, me)) {
                    mb = p + k;
                    return true;
                }
            }
        }
    }
    for (; p < end; ++p) {
        const uint32_t b = s[p];
        if ((b & 0xC0u) == 0x80u) continue;
        if (!((T.first_bytes[b >> 5] >> (b & 31u)) & 1u)) continue;
        if (special_match_at(T, tb, s, slen, p, no_strip_before, gb, ge, me)) {
            mb = p;
            return true;
        }
    }
    return false;
}

}  // namespace ovtk
