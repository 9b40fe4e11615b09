// fam_literal.hpp -- the literal matchers of the pattern families of span_fam.hpp (DeepSeek-V3's main pattern, o200k_base): the
// alternatives in the pattern's order, one position at a time -- what PCRE2 does with them (src/regex_split.cpp:286-301,
// src/utils.cpp:396-420), restated.  Included by split_device.hpp in front of scan_string (which walks rows with them where no mask
// form runs: the rows lookup_span_kernel leaves, calls of a few rows); span_fam.hpp has the patterns and the mask form.
#pragma once

namespace ovtk {

enum SpanFamily : int32_t { kFamNone = 0, kFamDs3 = 1, kFamO200k = 2 };

// SplitDev::uc_cls4: four class bits per code point, two to a byte, all 0x110000 of them (544 KiB; the text touches its first 32 KiB)
constexpr uint32_t kC4Other = 0, kC4Upper = 1 /* Lu Lt */, kC4Lower = 2 /* Ll */, kC4Both = 3 /* Lm Lo */, kC4Mark = 4 /* Mn Mc Me */,
                   kC4Num = 5 /* Nd Nl No */, kC4Punct = 6 /* P* S* */, kC4Space = 7 /* PCRE2's \s under UCP */;
// General_Category (the order of unicode_gc.inc: Cn Lu Ll Lt Lm Lo Mn Mc Me Nd Nl No Pc Pd Ps Pe Pi Pf Po Sm Sc Sk So Zs Zl Zp Cc Cf Cs Co)
// -> class; white space comes from the \s table (it overrides: U+0009..U+000D, U+0085 are Cc)
constexpr uint8_t kGcToC4[30] = {0, 1, 2, 1, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 0, 0, 0, 0};

__host__ __device__ inline uint32_t ascii_c4(uint32_t b) {
    if (b - 'A' < 26u) return kC4Upper;
    if (b - 'a' < 26u) return kC4Lower;
    if (b - '0' < 10u) return kC4Num;
    if (b == 0x20u || (b - 9u) < 5u) return kC4Space;
    if (b > 0x20u && b < 0x7Fu) return kC4Punct;   // every printable ASCII character that is not alphanumeric is \p{P} or \p{S}
    return kC4Other;
}
__device__ __forceinline__ uint32_t uc_c4(const SplitDev& sp, uint32_t cp) {
    if (cp >= 0x110000u) return kC4Other;
    return (uint32_t(sp.uc_cls4[cp >> 1]) >> (4 * (cp & 1u))) & 15u;
}
// seq_char with the eight classes (the same reading of broken UTF-8: a stray continuation byte is a character of class "other", a
// lead byte takes the continuation bytes that are there)
__device__ __forceinline__ SeqChar fam_char(const SplitDev& sp, const uint8_t* s, int pos, int slen) {
    const uint32_t b = s[pos];
    if (b < 0x80u) return SeqChar{b, 1, int(ascii_c4(b))};
    if (b < 0xC0u) return SeqChar{b, 1, int(kC4Other)};
    int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
    if (pos + n > slen) n = slen - pos;
    uint32_t cp = b & (0xFFu >> (n + 1));
    int len = 1;
    for (; len < n && (s[pos + len] & 0xC0u) == 0x80u; ++len) cp = (cp << 6) | (s[pos + len] & 0x3Fu);
    return SeqChar{cp, len, int(uc_c4(sp, cp))};
}
__device__ __forceinline__ bool c4_is_letter(int c) { return c >= int(kC4Upper) && c <= int(kC4Both); }

// The white-space alternatives both families share with Llama-3's pattern: \s*[\r\n]+ | \s+(?!\S) | \s+ at p (a white-space character).
__device__ __forceinline__ int fam_space_end(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    int e = p, after_last_break = -1, last_char = p;
    while (e < slen) {
        const SeqChar c = fam_char(sp, s, e, slen);
        if (c.cls != int(kC4Space)) break;
        last_char = e;
        e += c.len;
        if (is_line_break(c.cp)) after_last_break = e;
    }
    if (after_last_break >= 0) return after_last_break;
    if (e == slen) return e;
    if (last_char > p) return last_char;
    return e;
}

// ---- DeepSeek-V3: the end of the match that starts at p, or p when no alternative matches there
__device__ __forceinline__ int ds3_try(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    const SeqChar c0 = fam_char(sp, s, p, slen);
    auto ascii_letter = [](uint32_t b) { return ((b | 0x20u) - 'a') < 26u; };
    // [!-/:-@\[-`{-~][A-Za-z]+
    if (c0.cp < 0x80u && c0.cls == int(kC4Punct) && p + 1 < slen && ascii_letter(s[p + 1])) {
        int q = p + 2;
        while (q < slen && ascii_letter(s[q])) ++q;
        return q;
    }
    // [^\r\n\p{L}\p{P}\p{S}]?[\p{L}\p{M}]+   (with the optional character first; \p{M} is on both sides: either way the run ends where it ends)
    {
        auto word = [](int c) { return c4_is_letter(c) || c == int(kC4Mark); };
        int q = -1;
        if (word(c0.cls)) q = p;
        else if (!is_line_break(c0.cp) && c0.cls != int(kC4Punct) && p + c0.len < slen && word(fam_char(sp, s, p + c0.len, slen).cls)) q = p + c0.len;
        if (q >= 0) {
            while (q < slen) {
                const SeqChar c = fam_char(sp, s, q, slen);
                if (!word(c.cls)) break;
                q += c.len;
            }
            return q;
        }
    }
    //  ?[\p{P}\p{S}]+[\r\n]*
    {
        int q = -1;
        if (c0.cls == int(kC4Punct)) q = p;
        else if (c0.cp == ' ' && p + 1 < slen && fam_char(sp, s, p + 1, slen).cls == int(kC4Punct)) q = p + 1;
        if (q >= 0) {
            while (q < slen) {
                const SeqChar c = fam_char(sp, s, q, slen);
                if (c.cls != int(kC4Punct)) break;
                q += c.len;
            }
            while (q < slen && is_line_break(s[q])) ++q;
            return q;
        }
    }
    if (c0.cls == int(kC4Space)) return fam_space_end(sp, s, slen, p);
    return p;
}
// the piece that starts at p: a match, or the gap up to the next position where one starts
__device__ __forceinline__ int ds3_match_end(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    const int e = ds3_try(sp, s, slen, p);
    if (e > p) return e;
    int q = p + fam_char(sp, s, p, slen).len;
    while (q < slen && ds3_try(sp, s, slen, q) == q) q += fam_char(sp, s, q, slen).len;
    return q;
}

// ---- o200k_base
// [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+ (first) or ...+...* (second) at q, as a backtracking matcher takes them; -1: no match
__device__ __forceinline__ int o200k_word(const SplitDev& sp, const uint8_t* s, int slen, int q, bool first) {
    auto both = [](int c) { return c == int(kC4Both) || c == int(kC4Mark); };
    int a = q, last_both_end = -1;
    while (a < slen) {   // the greedy upper part
        const SeqChar c = fam_char(sp, s, a, slen);
        if (!(c.cls == int(kC4Upper) || both(c.cls))) break;
        a += c.len;
        if (both(c.cls)) last_both_end = a;
    }
    int e = a;
    while (e < slen) {   // the lower part behind it
        const SeqChar c = fam_char(sp, s, e, slen);
        if (!(c.cls == int(kC4Lower) || both(c.cls))) break;
        e += c.len;
    }
    if (first) {
        if (e > a) return e;
        // nothing lower behind the upper part: it gives characters back until one of them can be the lower part -- the LAST character
        // of both kinds; what follows it is upper case only, so the lower part is that one character
        return last_both_end;
    }
    return a > q ? e : -1;
}
__device__ __forceinline__ int o200k_suffix(const SplitDev& sp, const uint8_t* s, int slen, int e) {
    if (e < 0 || e + 1 >= slen || s[e] != '\'') return e;
    const SeqChar c1 = seq_char(sp, s, e + 1, slen);
    const uint32_t f1 = fold_contraction_letter(c1.cp);
    const int p2 = e + 1 + c1.len;
    if (f1 == 's' || f1 == 't' || f1 == 'm' || f1 == 'd') return p2;
    if ((f1 == 'r' || f1 == 'v' || f1 == 'l') && p2 < slen) {
        const SeqChar c2 = seq_char(sp, s, p2, slen);
        const uint32_t f2 = fold_contraction_letter(c2.cp);
        if ((f1 == 'l' && f2 == 'l') || (f1 != 'l' && f2 == 'e')) return p2 + c2.len;
    }
    return e;
}
__device__ __forceinline__ int o200k_match_end(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    const SeqChar c0 = fam_char(sp, s, p, slen);
    const bool pre_ok = !is_line_break(c0.cp) && !c4_is_letter(c0.cls) && c0.cls != int(kC4Num) && p + c0.len < slen;
    for (int alt = 0; alt < 2; ++alt) {   // each word alternative: with the optional character, then without
        if (pre_ok) {
            const int e = o200k_word(sp, s, slen, p + c0.len, alt == 0);
            if (e >= 0) return o200k_suffix(sp, s, slen, e);
        }
        const int e = o200k_word(sp, s, slen, p, alt == 0);
        if (e >= 0) return o200k_suffix(sp, s, slen, e);
    }
    // \p{N}{1,3}
    if (c0.cls == int(kC4Num)) {
        int q = p + c0.len;
        for (int k = 1; k < 3 && q < slen; ++k) {
            const SeqChar c = fam_char(sp, s, q, slen);
            if (c.cls != int(kC4Num)) break;
            q += c.len;
        }
        return q;
    }
    //  ?[^\s\p{L}\p{N}]+[\r\n/]*
    {
        auto other = [](int c) { return c == int(kC4Other) || c == int(kC4Mark) || c == int(kC4Punct); };
        int q = -1;
        if (other(c0.cls)) q = p;
        else if (c0.cp == ' ' && p + 1 < slen && other(fam_char(sp, s, p + 1, slen).cls)) q = p + 1;
        if (q >= 0) {
            while (q < slen) {
                const SeqChar c = fam_char(sp, s, q, slen);
                if (!other(c.cls)) break;
                q += c.len;
            }
            while (q < slen && (is_line_break(s[q]) || s[q] == '/')) ++q;
            return q;
        }
    }
    return fam_space_end(sp, s, slen, p);
}

template <int FAM>
__device__ __forceinline__ int fam_match_end(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    return FAM == kFamDs3 ? ds3_match_end(sp, s, slen, p) : o200k_match_end(sp, s, slen, p);
}

}  // namespace ovtk
