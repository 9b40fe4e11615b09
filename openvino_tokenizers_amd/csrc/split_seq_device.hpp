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
    uint32_t first_bytes[8];     // bytes a match can start with (token first bytes; whitespace lead bytes if any strip_left)
    SplitDev uc;                 // Unicode tables (whitespace class)
};

__device__ __forceinline__ int space_run_end(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    while (p < slen) {
        const SeqChar c = seq_char(sp, s, p, slen);
        if (c.cls != kClsS) break;
        p += c.len;
    }
    return p;
}
__device__ __forceinline__ bool token_at(const SpecialDev& T, int tok, const uint8_t* s, int slen, int p) {
    const int b = T.tok_begins[tok], n = T.tok_ends[tok] - b;
    if (n <= 0 || p + n > slen) return false;
    for (int k = 0; k < n; ++k)
        if (s[p + k] != T.tok_chars[b + k]) return false;
    return true;
}
// Match at start position p: returns true and the group / match extents.  no_strip_before: strip_left groups are known
// to fail for start positions below it (their candidate token positions shrink as p moves through a whitespace run).
__device__ __forceinline__ bool special_match_at(const SpecialDev& T, const uint8_t* s, int slen, int p, int& no_strip_before,
                                                 int& gb, int& ge, int& me) {
    int fail_until = no_strip_before;
    for (int g = 0; g < T.n_groups; ++g) {
        const uint32_t fl = T.group_flags[g];
        int hi = p;  // token positions tried: hi, then back through the whitespace run down to p
        if (fl & 1u) {
            if (p < no_strip_before) continue;
            hi = space_run_end(T.uc, s, slen, p);
        }
        for (int q = hi;;) {
            for (int t = T.group_first[g]; t < T.group_first[g + 1]; ++t) {
                if (token_at(T, t, s, slen, q)) {
                    gb = q;
                    ge = q + (T.tok_ends[t] - T.tok_begins[t]);
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
__device__ __forceinline__ bool special_next_match(const SpecialDev& T, const uint8_t* s, int slen, int start, int& mb, int& gb,
                                                   int& ge, int& me) {
    int no_strip_before = 0;
    for (int p = start; p < slen; ++p) {
        const uint32_t b = s[p];
        if ((b & 0xC0u) == 0x80u) continue;  // matches start at character boundaries (PCRE2_UTF)
        if (!((T.first_bytes[b >> 5] >> (b & 31u)) & 1u)) continue;
        if (special_match_at(T, s, slen, p, no_strip_before, gb, ge, me)) {
            mb = p;
            return true;
        }
    }
    return false;
}

}  // namespace ovtk
