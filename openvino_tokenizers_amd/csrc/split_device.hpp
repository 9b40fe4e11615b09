// split_device.hpp -- pre-tokenizer scanners (the device side of RegexSplit, src/regex_split.cpp:222-314).
//
// PCRE2 is not run on the GPU.  For the pattern families the reference's converter emits, a match
// that starts at p depends only on the text to the right of p (no look-behind), and every
// position matches something, so the sequence of isolate-mode pieces is determined by a LOCAL
// predicate "a piece starts at byte q" over a few neighbouring code points.
//
// GPT-2 byte-level pattern (tokenizer_pipeline.py:453-457), alternatives tried in order:
//   's|'t|'re|'ve|'m|'ll|'d | ?\p{L}+ | ?\p{N}+ | ?[^\s\p{L}\p{N}]+ | \s+(?!\S) | \s+
// With classes L, N, S(\s), O(other) and runs = maximal same-class stretches, PCRE2's leftmost,
// ordered-alternative, backtracking semantics reduce to (derivation in DESIGN.md, checked
// exhaustively against PCRE2 in tests/test_split_rules.py):
//   * a run boundary starts a piece, except that ONE ASCII space directly before an L/N/O run
//     belongs to that run's piece (" ?X+");
//   * inside an S run only its last char can start a piece, and only when the run is followed by
//     a non-space (the `\s+(?!\S)` back-off), never at end of string;
//   * an apostrophe that itself starts a piece and is followed by s,t,m,d,re,ve,ll forms a
//     contraction piece; the byte after the contraction starts a piece (this can split an L run);
//   * digits variant (:448-452, `\p{N}` single, no optional space): every N char is its own piece
//     and a preceding space does not attach to it.
//
// Evaluation is bit-parallel.  One wave scans one string: the window (<= 768 bytes incl. halos) is
// staged in LDS with coalesced dword loads, then
//   * ASCII windows (the common case) take the packed-byte path: lane l owns 8 or 12 window bytes (windows up to
//     512 / 768 bytes), loads the dwords that hold them and evaluates classes and rules on all of them at once with
//     SWAR arithmetic (one flag per byte in bit 7) -- ~250 vector instructions per 768-byte window;
//   * any other window takes the ballot path: for each 64-byte word every lane classifies ONE byte
//     (non-ASCII code points through the two-level Unicode property table generated from PCRE2
//     itself), the per-byte predicates become 64-bit masks by wave ballot, lane w owns the masks of
//     word w and the rules are ~40 AND/OR/shift operations on them, neighbouring words reached with
//     one DPP lane shift.
// Piece starts are finally turned into positions by popcount ranking.
#pragma once

#include "device_common.hpp"
#include "tables.hpp"

namespace ovtk {

enum SplitKind : int32_t {
    kSplitGpt2 = 0,        // byte_level_splitter()
    kSplitGpt2Digits = 1,  // byte_level_splitter(individual_digits=True)
    // "class" patterns: a match is a run of chars of one class (X+) or a single char of it; the text between
    // matches are the gaps; RegexSplit's behaviour / invert decide which of the two kinds of pieces are kept
    kSplitWhitespace = 2,  // bert_whitespace_splitter(): \s+
    kSplitBertPunct = 3,   // bert_keep_delimeters_splitter(): one char of [!-/] [:-@] [\[-`] [{-~] \p{P} or the CJK blocks
    kSplitBertWords = 4,   // both of the above chained (\s+ removed, then delimiters isolated): the fused WordPiece path
    kSplitLlama3 = 5,      // tiktoken-style pattern of Llama-3 (llama3_start_mask; its kernels are separate instantiations)
    kSplitGeneral = 6,     // any other pattern: the DFA of regex_compile.cpp, one lane per row (regex_device.hpp)
};

struct SplitDev {
    int32_t kind;
    const uint16_t* uc_index;  // [0x110000 >> 7]
    const uint8_t* uc_blocks;  // [n_blocks * 64]
    const uint8_t* uc_flat;    // [0x20000 / 4] the two class bits of the code points of planes 0 and 1, four to a byte: ONE load per
                               // character (span_l3.hpp)
    int32_t drop;              // class patterns: 0 keep every piece, 1 drop the matches, 2 drop the gaps
    // kSplitLlama3: what separates the patterns of its family from Llama-3's own (api_encode.cpp picks them by pattern text)
    int32_t l3_digits1;        // 1: `\p{N}` -- every digit a piece (Qwen2) -- instead of `\p{N}{1,3}`
    int32_t l3_tail_ws;        // 1: `\s++$` in front of the white-space alternatives (tiktoken's cl100k_base): the white-space run
                               //    that ends the string is ONE piece, line breaks inside it or not
    // kSplitGeneral with a pattern lookup_span_kernel has a scan for (span_fam.hpp: DeepSeek-V3's, o200k_base): the fused encode takes
    // that scan, every other caller of the handle the compiled DFA
    int32_t family;            // SpanFamily
    const uint8_t* uc_cls4;    // [0x110000 / 2] four class bits per code point (fam_literal.hpp kC4*)
};
constexpr uint16_t kPieceDropped = 0x8000;  // flag in WaveScratch::pstart: the piece is not emitted
constexpr uint16_t kPiecePosMask = 0x7FFF;

constexpr int kLaneDwords = 3;            // text dwords a lane classifies in the packed-byte (ASCII) scanner
constexpr int kChunk = 64 * 4 * kLaneDwords;  // window bytes (text whose piece starts are decided per pass + halos): 768,
                                          // so that a ~512-byte string with its +-10 % is ONE window (and a "simple" row)
constexpr int kLeftHalo = 8;
constexpr int kRightHalo = 12;
constexpr int kTextPad = 4;               // zero bytes in front of the staged text (the packed-byte path reads 4 bytes back)
constexpr int kWinBytes = kChunk + 32;    // pad + alignment skew + read-ahead of the 16-byte key loads, multiple of 4

constexpr uint8_t kClsO = 0, kClsL = 1, kClsN = 2, kClsS = 3;

// Per-wave LDS working set of the lookup / split kernels.
struct WaveScratch {
    uint32_t text_w[kWinBytes / 4];
    uint16_t pstart[kChunk + 2];
};

__device__ __forceinline__ const uint8_t* text_bytes(const WaveScratch& ws) {
    return reinterpret_cast<const uint8_t*>(ws.text_w) + kTextPad;
}
// The same two arrays by pointer: lookup_rows_kernel keeps TWO text windows per wave (the next row's text lands in one while
// the other is scanned) and one piece list.  The window-level functions it shares with the other kernels are templates over
// the scratch type.
struct WsView {
    uint32_t* text_w;
    uint16_t* pstart;
};

__device__ __forceinline__ uint32_t uc_nibble(const SplitDev& sp, uint32_t cp) {
    if (cp >= 0x110000u) return 0;
    const uint32_t blk = sp.uc_index[cp >> 7];
    const uint32_t b = sp.uc_blocks[blk * 64 + ((cp & 127) >> 1)];
    return (cp & 1) ? (b >> 4) : (b & 15);
}

// Class of an ASCII byte under PCRE2_UCP: \p{L} = [A-Za-z], \p{N} = [0-9], \s = [\t\n\v\f\r ] (checked against the
// generated table in tests/test_unicode_tables.py).
__device__ __forceinline__ uint32_t ascii_class(uint32_t b) {
    if (((b | 0x20u) - 'a') < 26u) return kClsL;
    if ((b - '0') < 10u) return kClsN;
    if (b == 0x20u || (b - 9u) < 5u) return kClsS;
    return kClsO;
}

// ---- the Llama-3 pattern, matched literally (one position at a time) ------------------------------
// (?i:'s|'t|'re|'ve|'m|'ll|'d) | [^\r\n\p{L}\p{N}]?\p{L}+ | \p{N}{1,3} | ?[^\s\p{L}\p{N}]+[\r\n]* | \s*[\r\n]+ | \s+(?!\S) | \s+
// The alternatives are tried in PCRE2's order, each with its greedy / back-off behaviour.  Used by the lane-per-row
// kernels of split_seq_device.hpp and as the fallback of the bit-parallel Llama-3 scanner below.
struct SeqChar {
    uint32_t cp;
    int len;   // bytes
    int cls;   // kClsO / kClsL / kClsN / kClsS (line breaks are kClsS)
};

__device__ __forceinline__ SeqChar seq_char(const SplitDev& sp, const uint8_t* s, int pos, int slen) {
    const uint32_t b = s[pos];
    if (b < 0x80u) return SeqChar{b, 1, int(ascii_class(b))};
    if (b < 0xC0u) return SeqChar{b, 1, kClsO};  // stray continuation byte (invalid UTF-8: parity is undefined)
    int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
    if (pos + n > slen) n = slen - pos;
    uint32_t cp = b & (0xFFu >> (n + 1));
    int len = 1;
    for (; len < n && (s[pos + len] & 0xC0u) == 0x80u; ++len) cp = (cp << 6) | (s[pos + len] & 0x3Fu);
    return SeqChar{cp, len, int(uc_nibble(sp, cp) & 3u)};
}
// Unicode simple case folding restricted to what can equal one of s t r e v m l d (PCRE2_UCP caseless matching):
// ASCII upper case, and U+017F LATIN SMALL LETTER LONG S which folds to 's'.
__device__ __forceinline__ uint32_t fold_contraction_letter(uint32_t cp) {
    if (cp - 'A' < 26u) return cp + 32u;
    if (cp == 0x17Fu) return 's';
    return cp;
}
__device__ __forceinline__ bool is_line_break(uint32_t cp) { return cp == '\r' || cp == '\n'; }

// End of the match that starts at p (p < slen).
__device__ __forceinline__ int llama3_match_end(const SplitDev& sp, const uint8_t* s, int slen, int p) {
    const SeqChar c0 = seq_char(sp, s, p, slen);
    // (?i:'s|'t|'re|'ve|'m|'ll|'d)
    if (c0.cp == '\'' && p + 1 < slen) {
        const SeqChar c1 = seq_char(sp, s, p + 1, slen);
        const uint32_t f1 = fold_contraction_letter(c1.cp);
        const int p2 = p + 1 + c1.len;
        if (f1 == 's' || f1 == 't' || f1 == 'm' || f1 == 'd') return p2;
        if ((f1 == 'r' || f1 == 'v' || f1 == 'l') && p2 < slen) {
            const SeqChar c2 = seq_char(sp, s, p2, slen);
            const uint32_t f2 = fold_contraction_letter(c2.cp);
            if ((f1 == 'l' && f2 == 'l') || (f1 != 'l' && f2 == 'e')) return p2 + c2.len;
        }
    }
    // [^\r\n\p{L}\p{N}]?\p{L}+   (the optional char is tried first; without it \p{L}+ must start at p)
    {
        int q = -1;
        if (c0.cls == kClsL) q = p;
        else if (c0.cls != kClsN && !is_line_break(c0.cp) && p + c0.len < slen && seq_char(sp, s, p + c0.len, slen).cls == kClsL)
            q = p + c0.len;
        if (q >= 0) {
            while (q < slen) {
                const SeqChar c = seq_char(sp, s, q, slen);
                if (c.cls != kClsL) break;
                q += c.len;
            }
            return q;
        }
    }
    // \p{N}{1,3}
    if (c0.cls == kClsN) {
        int q = p + c0.len;
        for (int k = 1; k < (sp.l3_digits1 ? 1 : 3) && q < slen; ++k) {
            const SeqChar c = seq_char(sp, s, q, slen);
            if (c.cls != kClsN) break;
            q += c.len;
        }
        return q;
    }
    //  ?[^\s\p{L}\p{N}]+[\r\n]*
    {
        int q = -1;
        if (c0.cls == kClsO) q = p;
        else if (c0.cp == ' ' && p + 1 < slen && seq_char(sp, s, p + 1, slen).cls == kClsO) q = p + 1;
        if (q >= 0) {
            while (q < slen) {
                const SeqChar c = seq_char(sp, s, q, slen);
                if (c.cls != kClsO) break;
                q += c.len;
            }
            while (q < slen && is_line_break(s[q])) ++q;
            return q;
        }
    }
    // whitespace run [p, e): \s*[\r\n]+ (up to its last line break) | \s+(?!\S) (all but the last char, or all of it
    // at the end of the string) | \s+
    int e = p, after_last_break = -1, last_char = p;
    while (e < slen) {
        const SeqChar c = seq_char(sp, s, e, slen);
        if (c.cls != kClsS) break;
        last_char = e;
        e += c.len;
        if (is_line_break(c.cp)) after_last_break = e;
    }
    if (sp.l3_tail_ws && e == slen) return e;   // \s++$ (tried before \s*[\r\n])
    if (after_last_break >= 0) return after_last_break;
    if (e == slen) return e;
    if (last_char > p) return last_char;
    return e;
}

// ---- the GPT-2 family's pattern matched literally: the piece that starts at byte p of the string s[0, slen) (p a true piece start),
// alternatives in the pattern's order.  For pieces longer than a scan block of lookup_span_kernel (the block scan finds no second start to
// end them with), and what the bit-mask form of the rules (span_l3.hpp) is fuzzed against.
//   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+      (digits: \p{N} for " ?\p{N}+")
__device__ __forceinline__ int gpt2_match_end(const SplitDev& sp, const uint8_t* s, int slen, int p, bool digits) {
    const SeqChar c0 = seq_char(sp, s, p, slen);
    if (c0.cp == '\'' && p + 1 < slen) {
        const uint32_t a = s[p + 1], b = p + 2 < slen ? s[p + 2] : 0u;
        if (a == 's' || a == 't' || a == 'm' || a == 'd') return p + 2;
        if (((a == 'r' || a == 'v') && b == 'e') || (a == 'l' && b == 'l')) return p + 3;
    }
    int q = -1, cls = -1;   // a run of class `cls` from q on, behind an optional U+0020
    if (c0.cls != kClsS) {
        q = p;
        cls = c0.cls;
    } else if (c0.cp == ' ' && p + 1 < slen) {
        const SeqChar c1 = seq_char(sp, s, p + 1, slen);
        if (c1.cls != kClsS && !(digits && c1.cls == kClsN)) {
            q = p + 1;
            cls = c1.cls;
        }
    }
    if (q >= 0) {
        if (digits && cls == kClsN) return q + seq_char(sp, s, q, slen).len;
        while (q < slen) {
            const SeqChar c = seq_char(sp, s, q, slen);
            if (c.cls != cls) break;
            q += c.len;
        }
        return q;
    }
    int e = p, last_char = p;   // white space: all of the run at the string's end, else all but its last character (or that one alone)
    while (e < slen) {
        const SeqChar c = seq_char(sp, s, e, slen);
        if (c.cls != kClsS) break;
        last_char = e;
        e += c.len;
    }
    if (e == slen || last_char == p) return e;
    return last_char;
}
// Stage bytes [w0, w1) of the string at `str` (global) into ws.text_w.  Returns the skew: string
// byte p lives at text_bytes(ws)[p - w0 + skew].  Whole aligned dwords are fetched; a dword that straddles an end of
// the string is masked down to the string's own bytes (the others are staged as zeros) -- it is still one load as long
// as it lies inside the chars tensor [buf, buf_end); only at the tensor's own unaligned ends are bytes loaded singly.
template <class WS>
__device__ __forceinline__ int stage_window(WS& ws, const uint8_t* str, int slen, int w0, int w1, const uint8_t* buf,
                                            const uint8_t* buf_end) {
    const uint8_t* g = str + w0;
    const int skew = int(reinterpret_cast<uintptr_t>(g) & 3);
    const uint8_t* ga = g - skew;
    const int nwords = (skew + (w1 - w0) + 3) >> 2;
    const int lo = skew - w0, hi = slen - w0 + skew;  // the string's own bytes, as offsets from ga
    if (lane_id() == 0) ws.text_w[0] = 0;  // kTextPad
    for (int k = lane_id(); k < nwords; k += kWave) {
        const int b0 = k * 4;
        uint32_t w;
        if (b0 >= lo && b0 + 4 <= hi) {
            w = stream_load(reinterpret_cast<const uint32_t*>(ga + b0));
        } else if (ga + b0 >= buf && ga + b0 + 4 <= buf_end) {
            int cl = lo - b0, ch = hi - b0;  // keep bytes [cl, ch) of the dword
            cl = cl < 0 ? 0 : (cl > 4 ? 4 : cl);
            ch = ch < 0 ? 0 : (ch > 4 ? 4 : ch);
            const uint32_t below_ch = ch >= 4 ? ~0u : ((1u << (8 * ch)) - 1u);
            const uint32_t below_cl = cl >= 4 ? ~0u : ((1u << (8 * cl)) - 1u);
            w = *reinterpret_cast<const uint32_t*>(ga + b0) & below_ch & ~below_cl;
        } else {
            w = 0;
            for (int j = 0; j < 4; ++j)
                if (b0 + j >= lo && b0 + j < hi) w |= uint32_t(ga[b0 + j]) << (8 * j);
        }
        ws.text_w[kTextPad / 4 + k] = w;
    }
    return skew;
}

// Code point of the char whose lead byte b (>= 0xC0) is window byte i; a char cut off by the window's end is decoded
// from the bytes that are there.  The four bytes from i come as two aligned LDS dwords and a funnel shift: no loop.
template <class WS>
__device__ __forceinline__ uint32_t decode_lead(const WS& ws, int skew, int i, uint32_t b, int wlen) {
    const int off = kTextPad + skew + i;
    const int a = off >> 2, sh = (off & 3) * 8;
    const uint32_t x = funnel_shr(ws.text_w[a], ws.text_w[a + 1], sh);
    const int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
    const int have = wlen - i < n ? wlen - i : n;
    uint32_t cp = b & (0xFFu >> (n + 1));
    if (have >= 2) cp = (cp << 6) | ((x >> 8) & 0x3Fu);
    if (have >= 3) cp = (cp << 6) | ((x >> 16) & 0x3Fu);
    if (have >= 4) cp = (cp << 6) | ((x >> 24) & 0x3Fu);
    return cp;
}

// ---- 64-bit masks, lane w = window bytes [64w, 64w + 64) ---------------------------------------
using Mask = unsigned long long;
// Bit i of the result = bit (i - k) of the window-wide mask (k in 1..4): "property of the byte k places before".
template <int K>
__device__ __forceinline__ Mask mask_from_before(Mask v) { return (v << K) | (row_prev(v) >> (64 - K)); }
// Bit i of the result = bit (i + k): "property of the byte k places after".
template <int K>
__device__ __forceinline__ Mask mask_from_after(Mask v) { return (v >> K) | (row_next(v) << (64 - K)); }

// Piece-start mask of the window [w0, w1) staged at `skew`: lane w returns the bits of window bytes
// [64w, 64w+64).  Bit i <=> "a piece starts at string position w0 + i" according to the GPT-2 family rules;
// positions w0 + i == 0 and chunk starts are forced by the caller.  Wave-uniform call.
__device__ __forceinline__ const uint8_t* text_bytes(const WsView& ws) {
    return reinterpret_cast<const uint8_t*>(ws.text_w) + kTextPad;
}
template <class WS>
__device__ __forceinline__ Mask gpt2_start_mask(const WS& ws, const SplitDev& sp, int skew, int wlen, bool digits) {
    const int l = lane_id();
    const uint8_t* t = text_bytes(ws) + skew;
    Mask mL = 0, mN = 0, mS = 0, mSP = 0, mCONT = 0, mAP = 0, mX1 = 0, mX2 = 0, mXE = 0, mXL = 0;
    const int nwords = (wlen + 63) >> 6;
    for (int w = 0; w < nwords; ++w) {
        const int i = w * 64 + l;
        const bool valid = i < wlen;
        const uint32_t b = valid ? t[i] : 0u;
        uint32_t cls = kClsO;
        if (b < 0x80u) {
            cls = valid ? ascii_class(b) : kClsO;
        } else if (b >= 0xC0u) {  // lead byte: decode (truncated at the window edge: the right halo covers real chars)
            const uint32_t cp = decode_lead(ws, skew, i, b, wlen);
            cls = uc_nibble(sp, cp) & 3u;
        }
        const Mask bL = __ballot(cls == kClsL), bN = __ballot(cls == kClsN), bS = __ballot(cls == kClsS);
        const Mask bSP = __ballot(b == 0x20u), bCONT = __ballot((b & 0xC0u) == 0x80u), bAP = __ballot(b == 0x27u);
        const Mask bX1 = __ballot(b == 's' || b == 't' || b == 'm' || b == 'd');
        const Mask bX2 = __ballot(b == 'r' || b == 'v'), bXE = __ballot(b == 'e'), bXL = __ballot(b == 'l');
        if (l == w) {
            mL = bL; mN = bN; mS = bS; mSP = bSP; mCONT = bCONT;
            mAP = bAP; mX1 = bX1; mX2 = bX2; mXE = bXE; mXL = bXL;
        }
    }
    // window-wide validity: byte i exists
    const int rem = wlen - l * 64;
    const Mask mV = rem >= 64 ? ~0ull : (rem > 0 ? ((1ull << rem) - 1ull) : 0ull);
    // continuation bytes take the class of their lead byte (chars are <= 4 bytes)
    if (__ballot(mCONT != 0)) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            mL |= mask_from_before<1>(mL) & mCONT;
            mN |= mask_from_before<1>(mN) & mCONT;
            mS |= mask_from_before<1>(mS) & mCONT;
        }
    }
    const Mask mO = mV & ~(mL | mN | mS);
    const Mask pL = mask_from_before<1>(mL), pN = mask_from_before<1>(mN), pS = mask_from_before<1>(mS);
    const Mask pO = mask_from_before<1>(mO), pSP = mask_from_before<1>(mSP);
    // contractions: an apostrophe followed by s|t|m|d resp. re|ve|ll (all inside the string) ...
    const Mask c1 = mAP & mask_from_after<1>(mX1);
    const Mask c2 = mAP & ((mask_from_after<1>(mX2) & mask_from_after<2>(mXE)) | (mask_from_after<1>(mXL) & mask_from_after<2>(mXL)));
    // ... fires when the apostrophe itself starts a piece: previous char is neither class O nor U+0020
    const Mask f1 = c1 & ~(pO | pSP), f2 = c2 & ~(pO | pSP);
    const Mask same = (mL & pL) | (mN & pN) | (mS & pS) | (mO & pO);
    const Mask attaches = ~mS & (digits ? ~mN : ~0ull);
    Mask start = ~same & ~(pSP & attaches);
    // same class as the previous char: the last char of a whitespace run before a non-space; every digit (digits variant)
    const Mask cs = mV & ~mCONT, nscs = cs & ~mS, nsv = mV & ~mS;
    const Mask a1 = mask_from_after<1>(mCONT), a2 = mask_from_after<2>(mCONT), a3 = mask_from_after<3>(mCONT);
    const Mask next_nonspace = mask_from_after<1>(nscs) | (a1 & mask_from_after<2>(nscs)) | (a1 & a2 & mask_from_after<3>(nscs)) |
                               (a1 & a2 & a3 & mask_from_after<4>(nsv));
    start |= same & ((mS & next_nonspace) | (digits ? mN : 0ull));
    start |= mask_from_before<2>(f1) | mask_from_before<3>(f2);      // the byte after a contraction
    start &= ~mask_from_before<1>(f1 | f2);                          // the contraction's first letter stays with it
    return start & cs;
}

// ---- the Llama-3 pattern, bit-parallel ------------------------------------------------------------
// Derived from the matcher above (no look-behind in the pattern: what matches at a piece start depends only on the
// text from there on).  With byte-run masks L / N / W (white space) / O, NL = line breaks, SP = U+0020:
//   * a run start is a piece start, except: an L run takes ONE char in front of it when that char is white space other
//     than a line break, or an O run of a single char that is neither preceded by U+0020 nor a contraction;
//     an O run hands its start to a U+0020 directly in front of it; a W run that follows an O char hands its leading
//     line breaks to that piece (the next start is after them);
//   * digits: every third position of an N run (ASCII digits; other \p{N} chars make the window fall back);
//   * inside a W run: after its LAST line break, and in front of its last char when at least two chars follow the last
//     line break and the run is followed by something;
//   * an apostrophe that starts a piece and is followed by s|t|m|d or re|ve|ll (any case; U+017F falls back) is a
//     piece of its own with those letters, and the next piece starts right behind them.
// The three non-local parts (digit phase, "a line break follows in this run", "these line breaks follow an O char")
// are carry ripples over the at most nine 64-byte words of the window, done on the scalar unit, forward and backward.
__device__ __forceinline__ Mask readlane_mask(Mask v, int w) {
    const unsigned lo = unsigned(wave_readlane(int(unsigned(v)), w)), hi = unsigned(wave_readlane(int(unsigned(v >> 32)), w));
    return (static_cast<Mask>(hi) << 32) | lo;
}
// Bits of `run` reached from `seed` (a subset of run) going up through contiguous run bits; carry: the ripple entered
// through bit 0 / left through bit 63.
__device__ __forceinline__ Mask ripple_up(Mask run, Mask seed, bool& carry) {
    const Mask x = seed | ((carry && (run & 1ull)) ? 1ull : 0ull);
    const Mask sum = run + x;
    carry = sum < run;
    return (run & ~sum) | x;
}
// lead byte of the char whose LAST byte is flagged in `at_last` (chars are at most 4 bytes)
__device__ __forceinline__ Mask to_lead(Mask at_last, Mask cont) {
    const Mask a1 = mask_from_after<1>(cont), a2 = mask_from_after<2>(cont), a3 = mask_from_after<3>(cont);
    return ~cont & (at_last | (a1 & mask_from_after<1>(at_last)) | (a1 & a2 & mask_from_after<2>(at_last)) |
                    (a1 & a2 & a3 & mask_from_after<3>(at_last)));
}

// Piece starts of the window [w0, w1) (wlen bytes, staged at `skew`); lo = window position of the chunk start (a piece
// start by construction).  at_end: the window reaches the end of the string.  undecided: starts at window positions
// >= undecided may depend on text behind the window.  fallback (wave-uniform): the window holds something only the
// literal matcher handles.
__device__ __forceinline__ Mask llama3_start_mask(const WaveScratch& ws, const SplitDev& sp, int skew, int wlen, int lo, bool at_end,
                                                  int& undecided, bool& fallback) {
    const int l = lane_id();
    const uint8_t* t = text_bytes(ws) + skew;
    Mask mL = 0, mN = 0, mW = 0, mNL = 0, mSP = 0, mCONT = 0, mAP = 0, mX1 = 0, mX2 = 0, mXE = 0, mXL = 0;
    bool odd = false;
    Mask ap_tail = 0;  // wave-uniform: an apostrophe among the last two bytes of the previous word
    const int nwords = (wlen + 63) >> 6;
    for (int w = 0; w < nwords; ++w) {
        const int i = w * 64 + l;
        const bool valid = i < wlen;
        const uint32_t b = valid ? t[i] : 0u;
        uint32_t cls = kClsO;
        if (b < 0x80u) {
            cls = valid ? ascii_class(b) : kClsO;
        } else if (b >= 0xC0u) {
            const uint32_t cp = decode_lead(ws, skew, i, b, wlen);
            cls = uc_nibble(sp, cp) & 3u;
            if (cls == kClsN || cp == 0x17Fu) odd = true;  // a non-ASCII digit; LATIN SMALL LETTER LONG S folds to 's'
        }
        const uint32_t f = b | 0x20u;  // ASCII letters folded to lower case
        const Mask bL = __ballot(cls == kClsL), bN = __ballot(cls == kClsN), bW = __ballot(cls == kClsS);
        const Mask bNL = __ballot(b == '\r' || b == '\n'), bSP = __ballot(b == 0x20u), bCONT = __ballot((b & 0xC0u) == 0x80u);
        const Mask bAP = __ballot(b == 0x27u);
        if (l == w) {
            mL = bL; mN = bN; mW = bW; mNL = bNL; mSP = bSP; mCONT = bCONT;
            mAP = bAP;
        }
        // contraction letters matter in the two bytes behind an apostrophe only: this word's, or the first two bytes of
        // the next word when the previous one ended in an apostrophe (most words of a window have neither)
        if (bAP | ap_tail) {
            const Mask bX1 = __ballot(f == 's' || f == 't' || f == 'm' || f == 'd');
            const Mask bX2 = __ballot(f == 'r' || f == 'v'), bXE = __ballot(f == 'e'), bXL = __ballot(f == 'l');
            if (l == w) { mX1 = bX1; mX2 = bX2; mXE = bXE; mXL = bXL; }
        }
        ap_tail = bAP >> 62;
    }
    fallback = __ballot(odd) != 0;
    const int rem = wlen - l * 64;
    const Mask mV = rem >= 64 ? ~0ull : (rem > 0 ? ((1ull << rem) - 1ull) : 0ull);
    if (__ballot(mCONT != 0)) {  // continuation bytes take the class of their lead byte
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            mL |= mask_from_before<1>(mL) & mCONT;
            mW |= mask_from_before<1>(mW) & mCONT;
        }
    }
    const Mask mO = mV & ~(mL | mN | mW);
    const Mask cs = mV & ~mCONT;
    const Mask pL = mask_from_before<1>(mL), pW = mask_from_before<1>(mW), pO = mask_from_before<1>(mO);
    const Mask pSP = mask_from_before<1>(mSP), pNL = mask_from_before<1>(mNL);
    // the chunk start is a piece start: digit groups count from it
    Mask mNd = mN;
    if (lo > 0 && l == ((lo - 1) >> 6)) mNd &= ~(1ull << ((lo - 1) & 63));
    const Mask seedN = mNd & ~mask_from_before<1>(mNd);            // N run starts
    const Mask seedA = mNL & ~pNL & pO;                             // line breaks right behind an O char
    // ---- forward ripples: digit groups G, absorbed line breaks F (windows without a digit / a line break skip theirs)
    const bool any_n = __ballot(mN != 0) != 0, any_nl = __ballot(mNL != 0) != 0;
    Mask G = 0, F = 0;
    if (any_nl) {
        bool cF = false;
        for (int w = 0; w < nwords; ++w) {
            const Mask f = ripple_up(readlane_mask(mNL, w), readlane_mask(seedA, w), cF);
            if (l == w) F = f;
        }
    }
    if (any_n && sp.l3_digits1) {
        G = mN;   // `\p{N}`: every digit starts a piece
    } else if (any_n) {
        int k_prev = 0;  // digits in the last group of the previous word when its run reaches the word's end, else 0
        for (int w = 0; w < nwords; ++w) {
            const Mask Nw = readlane_mask(mNd, w), Sw = readlane_mask(seedN, w);
            Mask g = Sw;
            if (k_prev && (Nw & 1ull)) {  // the run continues from the previous word: its next group starts 3 - k_prev digits in
                const Mask head = Nw & ~(Nw + 1ull);  // the run of ones that starts at bit 0
                g |= (1ull << (3 - k_prev)) & head;
            }
            const Mask a3 = Nw & (Nw << 1) & (Nw << 2) & (Nw << 3);
            for (;;) {
                const Mask ng = ((g << 3) & a3) & ~g;
                if (!ng) break;
                g |= ng;
            }
            k_prev = 0;
            if (Nw >> 63) {  // digits of the group still open at the end of this word (3: the next digit starts a group)
                const Mask top = ~Nw ? (~0ull << (64 - __clzll(~Nw))) : ~0ull;  // the run of ones that ends at bit 63
                const Mask gt = g & top;                                        // never empty: a run has a start every 3 digits
                k_prev = __clzll(gt) % 3 + 1;
            }
            if (l == w) G = g;
        }
    }
    // ---- backward ripple: D = "a line break follows (or is here) in this white-space run"
    Mask D = 0;
    if (any_nl) {
        bool c = false;
        for (int w = nwords - 1; w >= 0; --w) {
            const Mask Ww = readlane_mask(mW, w), NLw = readlane_mask(mNL, w);
            const Mask r = __brevll(ripple_up(__brevll(Ww), __brevll(NLw), c));
            if (l == w) D = r;
        }
    }
    // ---- local rules
    const Mask sL = mL & ~pL, sO = mO & ~pO, sW = mW & ~pW;
    const Mask endO = mO & ~mask_from_after<1>(mO), endW = mW & ~mask_from_after<1>(mW);
    const Mask single_o = sO & to_lead(endO, mCONT);                       // O runs of one char
    const Mask c1 = mAP & mask_from_after<1>(mX1);
    const Mask c2 = mAP & ((mask_from_after<1>(mX2) & mask_from_after<2>(mXE)) | (mask_from_after<1>(mXL) & mask_from_after<2>(mXL)));
    const Mask fire1 = c1 & sO & ~pSP, fire2 = c2 & sO & ~pSP & ~fire1;
    Mask takes = single_o & ~pSP & ~(fire1 | fire2);                       // this O char goes in front of the letters behind it
    if (__ballot(mCONT != 0)) {
#pragma unroll
        for (int r = 0; r < 3; ++r) takes |= mask_from_before<1>(takes) & mCONT;
    }
    const Mask supL = sL & (mask_from_before<1>(mW & ~mNL) | mask_from_before<1>(takes) | mask_from_before<1>(fire1 | fire2));
    const Mask supO = sO & pSP;
    const Mask supW = sW & pO & mNL;
    const Mask endNL = mNL & ~mask_from_after<1>(mNL);
    const Mask b_abs = mask_from_before<1>(F & endNL) & mW;                // behind the line breaks an O piece took
    Mask b_ln = mask_from_before<1>(mNL & ~mask_from_after<1>(D)) & mW;  // behind the run's last line break
    if (sp.l3_tail_ws && at_end && any_nl) {
        // `\s++$`: the run that ends the string is not cut behind its last line break.  T = the white-space bytes from which the
        // string's end is reached through white space only (a ripple down from the last byte)
        Mask T = 0;
        bool c = false;
        const int v = wlen - 1;
        for (int w = nwords - 1; w >= 0; --w) {
            const Mask Ww = readlane_mask(mW, w);
            const Mask seed = (w == (v >> 6)) ? ((1ull << (v & 63)) & Ww) : 0ull;
            const Mask r = __brevll(ripple_up(__brevll(Ww), __brevll(seed), c));
            if (l == w) T = r;
        }
        b_ln &= ~T;
    }
    const Mask last_w = to_lead(endW & mask_from_after<1>(mV), mCONT);     // last char of a W run that is followed by a char
    const Mask b_last = last_w & mW & ~mNL & pW & ~pNL;
    const Mask b_con = mask_from_before<2>(fire1) | mask_from_before<3>(fire2);
    const Mask start = ((sL & ~supL) | G | (sO & ~supO) | (sW & ~supW) | b_abs | b_ln | b_last | b_con) & cs;
    // ---- how far the window decides
    undecided = 0x7FFFFFFF;
    if (!at_end) {
        undecided = wlen > 8 ? wlen - 8 : 0;
        const int v = wlen - 1;
        if ((readlane_mask(mW, v >> 6) >> (v & 63)) & 1ull) {  // the window ends inside a white-space run: where does it start?
            int a = 0;
            for (int w = v >> 6; w >= 0; --w) {
                Mask nonw = ~readlane_mask(mW, w);
                if (w == (v >> 6) && (v & 63) != 63) nonw &= (1ull << ((v & 63) + 1)) - 1ull;
                if (nonw) {
                    a = w * 64 + (63 - __clzll(nonw)) + 1;
                    break;
                }
            }
            if (a + 1 < undecided) undecided = a + 1;
        }
    }
    return start;
}

// ---- class patterns (ballot path) ----------------------------------------------------------------
// The delimiter class of bert_keep_delimeters_splitter() (tokenizer_pipeline.py:397-426).
__device__ __forceinline__ bool bert_delimiter(uint32_t cp, uint32_t nibble) {
    if (cp < 0x80u) return (cp - 0x21u) < 15u || (cp - 0x3Au) < 7u || (cp - 0x5Bu) < 6u || (cp - 0x7Bu) < 4u;
    if (nibble & 4u) return true;  // \p{P}
    return (cp - 0x4E00u) < 0x5200u || (cp - 0x3400u) < 0x19C0u || (cp - 0x20000u) < 0xA6E0u || (cp - 0x2A700u) < 0x1040u ||
           (cp - 0x2B740u) < 0xE0u || (cp - 0x2B820u) < 0x1690u || (cp - 0xF900u) < 0x200u || (cp - 0x2F800u) < 0x220u;
}
// Piece starts and "dropped piece" flags of the window for the class patterns; lane w = window bytes [64w, 64w+64).
template <class WS>
__device__ __forceinline__ void class_start_mask(const WS& ws, const SplitDev& sp, int skew, int wlen, Mask& start,
                                                 Mask& dropped) {
    const int l = lane_id();
    const uint8_t* t = text_bytes(ws) + skew;
    Mask mS = 0, mP = 0, mCONT = 0;
    const int nwords = (wlen + 63) >> 6;
    for (int w = 0; w < nwords; ++w) {
        const int i = w * 64 + l;
        const bool valid = i < wlen;
        const uint32_t b = valid ? t[i] : 0u;
        bool is_s = false, is_p = false;
        if (b < 0x80u) {
            is_s = valid && ascii_class(b) == kClsS;
            is_p = valid && bert_delimiter(b, 0);
        } else if (b >= 0xC0u) {
            const uint32_t cp = decode_lead(ws, skew, i, b, wlen);
            const uint32_t nib = uc_nibble(sp, cp);
            is_s = (nib & 3u) == kClsS;
            is_p = bert_delimiter(cp, nib);
        }
        const Mask bS = __ballot(is_s), bP = __ballot(is_p), bCONT = __ballot((b & 0xC0u) == 0x80u);
        if (l == w) { mS = bS; mP = bP; mCONT = bCONT; }
    }
    const int rem = wlen - l * 64;
    const Mask mV = rem >= 64 ? ~0ull : (rem > 0 ? ((1ull << rem) - 1ull) : 0ull);
    if (__ballot(mCONT != 0)) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            mS |= mask_from_before<1>(mS) & mCONT;
            mP |= mask_from_before<1>(mP) & mCONT;
        }
    }
    const Mask cs = mV & ~mCONT;
    const Mask pS = mask_from_before<1>(mS), pP = mask_from_before<1>(mP);
    Mask m = 0;
    if (sp.kind == kSplitWhitespace) {
        m = mS;
        start = mS ^ pS;                 // a run of whitespace is one match
    } else if (sp.kind == kSplitBertPunct) {
        m = mP;
        start = mP | pP;                 // every delimiter char is a match of its own
    } else {                             // kSplitBertWords
        m = mS;
        start = (mS ^ pS) | mP | pP;
    }
    start &= cs;
    dropped = (sp.drop == 1 ? m : (sp.drop == 2 ? ~m : 0ull)) & cs;
}

// The LB + 1 aligned dwords that cover a lane's LB text dwords at byte offset `off` of the LDS window -- or (default) the LB
// dwords themselves, read at their byte address: r[j] and r[j + 1] then funnel-shift to themselves (sh = 0 below).
template <int N>
struct __attribute__((packed, aligned(1))) LdsDwords { uint32_t d[N]; };
template <int LB>
__device__ __forceinline__ int lds_lane_dwords(const uint32_t* text_w, int off, uint32_t (&r)[LB + 1]) {
    const LdsDwords<LB> v = *reinterpret_cast<const LdsDwords<LB>*>(reinterpret_cast<const uint8_t*>(text_w) + off);
#pragma unroll
    for (int j = 0; j < LB; ++j) r[j] = v.d[j];
    r[LB] = 0;
    return 0;
}

// ---- packed-byte (SWAR) path: ASCII windows (at most kChunk bytes) -------------------------------
// All values are 4 packed bytes with one flag per byte in bit 7.  Inputs must be < 0x80 per byte (no carries).
constexpr uint32_t kB7 = 0x80808080u;
__device__ __forceinline__ uint32_t swar_eq(uint32_t x, uint32_t c) {  // x == c, per byte
    return ~((x ^ (c * 0x01010101u)) + 0x7F7F7F7Fu) & kB7;
}
__device__ __forceinline__ uint32_t swar_range(uint32_t x, uint32_t lo, uint32_t hi) {  // lo <= x <= hi, per byte
    return (x + (0x80u - lo) * 0x01010101u) & ~(x + (0x7Fu - hi) * 0x01010101u) & kB7;
}
// {hi:lo} >> 8*k, the low dword: bytes shifted towards lower positions ("property of the byte k places after").
template <int K>
__device__ __forceinline__ uint32_t swar_after(uint32_t lo, uint32_t hi) { return (lo >> (8 * K)) | (hi << (32 - 8 * K)); }
// "property of the byte k places before": bytes shifted towards higher positions, `lo` = the dword below.
template <int K>
__device__ __forceinline__ uint32_t swar_before(uint32_t lo, uint32_t hi) { return (hi << (8 * K)) | (lo >> (32 - 8 * K)); }

// Piece-start flags of window bytes [LBy*l, LBy*(l+1)) for lane l, LBy = 4*LB (bit k of `flags` = byte LBy*l + k), same
// rules as gpt2_start_mask.  Returns false (wave-uniform) when the window holds a non-ASCII byte: the caller then takes
// the ballot path.  A lane classifies only its own LB dwords; what the rules need from the bytes before and after them
// (the classes of the previous byte, "next byte is not a space", the contraction letters after an apostrophe,
// contractions that fired up to three bytes back) comes from the neighbouring lanes with one DPP wavefront shift per
// value.  Index convention below: [0] = last dword of lane l-1, [1..LB] = own dwords, [LB+1] = first dword of lane l+1.
template <int LB, class WS>
__device__ __forceinline__ bool gpt2_start_flags_ascii(const WS& ws, int skew, int wlen, bool digits, uint32_t& flags) {
    constexpr int LBy = 4 * LB;
    const int l = lane_id();
    const int off = kTextPad + skew + LBy * l;  // byte offset of the lane's first byte in text_w
    int nv = wlen - LBy * l;  // how many of the lane's bytes exist
    nv = nv < 0 ? 0 : (nv > LBy ? LBy : nv);
    uint32_t x[LB + 1], V[LB + 1], r[LB + 1];
    const int sh = lds_lane_dwords<LB>(&ws.text_w[0], off, r);
    uint32_t any = 0;
#pragma unroll
    for (int j = 1; j <= LB; ++j) {
        const int have = nv - 4 * (j - 1);  // bytes of dword j that exist
        const uint32_t m = have >= 4 ? ~0u : (have <= 0 ? 0u : ((1u << (8 * have)) - 1u));
        x[j] = funnel_shr(r[j - 1], r[j], sh) & m;
        V[j] = m & kB7;
        any |= x[j];
    }
    if (__ballot((any & kB7) != 0)) return false;
    uint32_t L[LB + 1], N[LB + 1], S[LB + 1], SP[LB + 1], AP[LB + 1], O[LB + 1];
#pragma unroll
    for (int j = 1; j <= LB; ++j) {
        L[j] = swar_range(x[j] | 0x20202020u, 'a', 'z');
        N[j] = swar_range(x[j], '0', '9');
        SP[j] = swar_eq(x[j], 0x20) & V[j];  // a non-existing byte is 0x00: never a letter, digit, apostrophe; mask the rest
        S[j] = SP[j] | (swar_range(x[j], 9, 13) & V[j]);
        AP[j] = swar_eq(x[j], 0x27);
        O[j] = V[j] & ~(L[j] | N[j] | S[j]);
    }
    L[0] = lane_prev(L[LB]);
    N[0] = lane_prev(N[LB]);
    S[0] = lane_prev(S[LB]);
    O[0] = lane_prev(O[LB]);
    SP[0] = lane_prev(SP[LB]);
    uint32_t NS[LB + 2];  // "exists and is not white space"
#pragma unroll
    for (int j = 1; j <= LB; ++j) NS[j] = V[j] & ~S[j];
    NS[LB + 1] = lane_next(NS[1]);
    // Contractions ('x / 'xx) firing at an apostrophe.  Apostrophes are few (about one per row of text), so only the
    // dwords that hold one look at the two bytes behind it -- a divergent branch per dword of the lane whose body most
    // waves skip -- instead of classifying every byte of the window as a contraction letter.
    uint32_t f1[LB + 1], f2[LB + 1];
#pragma unroll
    for (int j = 0; j <= LB; ++j) f1[j] = f2[j] = 0;
    uint32_t any_ap = 0;
#pragma unroll
    for (int j = 1; j <= LB; ++j) any_ap |= AP[j];
    bool fired = false;  // wave-uniform: some contraction fired in the window
    if (__ballot(any_ap != 0)) {
        uint32_t xn[LB + 2];
#pragma unroll
        for (int j = 1; j <= LB; ++j) xn[j] = x[j];
        xn[LB + 1] = lane_next(x[1]);
#pragma unroll
        for (int j = 1; j <= LB; ++j) {
            if (AP[j]) {
                const uint32_t n1 = swar_after<1>(xn[j], xn[j + 1]), n2 = swar_after<2>(xn[j], xn[j + 1]);  // the next two bytes
                const uint32_t c1 = AP[j] & (swar_range(n1, 's', 't') | swar_eq(n1, 'm') | swar_eq(n1, 'd'));
                const uint32_t c2 = AP[j] & (((swar_eq(n1, 'r') | swar_eq(n1, 'v')) & swar_eq(n2, 'e')) | (swar_eq(n1, 'l') & swar_eq(n2, 'l')));
                // the apostrophe itself starts a piece unless the previous char is class O or U+0020
                const uint32_t blocked = swar_before<1>(O[j - 1] | SP[j - 1], O[j] | SP[j]);
                f1[j] = c1 & ~blocked;
                f2[j] = c2 & ~blocked;
            }
        }
        uint32_t any_f = 0;
#pragma unroll
        for (int j = 1; j <= LB; ++j) any_f |= f1[j] | f2[j];
        fired = __ballot(any_f != 0) != 0;
        if (fired) {
            f1[0] = lane_prev(f1[LB]);
            f2[0] = lane_prev(f2[LB]);
        }
    }
    uint32_t st[LB + 1];
#pragma unroll
    for (int j = 1; j <= LB; ++j) {
        const uint32_t pL = swar_before<1>(L[j - 1], L[j]), pN = swar_before<1>(N[j - 1], N[j]);
        const uint32_t pS = swar_before<1>(S[j - 1], S[j]), pO = swar_before<1>(O[j - 1], O[j]);
        const uint32_t pSP = swar_before<1>(SP[j - 1], SP[j]);
        const uint32_t same = (L[j] & pL) | (N[j] & pN) | (S[j] & pS) | (O[j] & pO);
        const uint32_t attaches = ~S[j] & (digits ? ~N[j] : ~0u);
        st[j] = ~same & ~(pSP & attaches);
        const uint32_t next_nonspace = swar_after<1>(NS[j], NS[j + 1]);
        st[j] |= same & ((S[j] & next_nonspace) | (digits ? N[j] : 0u));
    }
    if (fired) {
#pragma unroll
        for (int j = 1; j <= LB; ++j) {
            const uint32_t f12_lo = f1[j - 1] | f2[j - 1], f12 = f1[j] | f2[j];
            st[j] |= swar_before<2>(f1[j - 1], f1[j]) | swar_before<3>(f2[j - 1], f2[j]);  // the byte after a contraction
            st[j] &= ~swar_before<1>(f12_lo, f12);                                           // its first letter stays with it
        }
    }
    flags = 0;
#pragma unroll
    for (int j = 1; j <= LB; ++j) {
        // bit 7 of the dword's four bytes -> four adjacent bits
        const uint32_t t = (st[j] & V[j]) >> 7;
        flags |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu) << (4 * (j - 1));
    }
    return true;
}

// Packed-byte scan of an ASCII window, LB dwords per lane: ranks the starts of window bytes [lo, hi) (lo itself forced)
// into ws.pstart, relative to lo.  false (wave-uniform): the window is not ASCII, nothing was written.
template <int LB, class WS>
__device__ __forceinline__ bool gpt2_packed_starts(WS& ws, int skew, int wlen, bool digits, int lo, int hi, int& np) {
    uint32_t fl = 0;
    if (!gpt2_start_flags_ascii<LB>(ws, skew, wlen, digits, fl)) return false;
    // lane l holds the flags of window bytes [LBy*l, LBy*(l+1)), one bit per byte
    constexpr int LBy = 4 * LB;
    const int l = lane_id();
    int k_lo = lo - LBy * l, k_hi = hi - LBy * l;
    k_lo = k_lo < 0 ? 0 : (k_lo > LBy ? LBy : k_lo);
    k_hi = k_hi < 0 ? 0 : (k_hi > LBy ? LBy : k_hi);
    fl &= ((1u << k_hi) - 1u) & ~((1u << k_lo) - 1u);
    if (lo / LBy == l) fl |= 1u << (lo % LBy);
    const int cnt = __popc(fl);
    const int incl = wave_incl_sum(cnt);
    int at = incl - cnt;
    const int first = LBy * l - lo;
    while (fl) {
        ws.pstart[at++] = uint16_t(first + __ffs(fl) - 1);
        fl &= fl - 1;
    }
    np = wave_readlane(incl, kWave - 1);
    return true;
}

// ---- the Llama-3 pattern, packed bytes ------------------------------------------------------------------------------
// llama3_start_mask's rules with every lane deciding its own 12 bytes (three dwords, one flag per byte in bit 7) instead of
// nine lanes holding a 64-byte word each: the rule algebra is the same, but all 64 lanes do useful work and nothing is
// gathered by ballots.  What a rule needs from the bytes around the lane's own comes from the two neighbouring lanes' dwords
// (DPP moves): index 0..1 = the two dwords before, 2..4 = own, 5..6 = the two after.  Non-ASCII characters are classified by
// a per-lane loop over the lane's lead bytes (table look-up per character, as in the ballot form).  The three ripples
// become bounded look-arounds, and the window goes to the ballot form (return false, wave-uniform) when a bound is hit: a
// digit run of nine or more, five line breaks in a row, a line break followed by four or more further white-space bytes;
// and, as there, a non-ASCII digit or U+017F sends it to the literal matcher (fallback).
template <int K, int M>
__device__ __forceinline__ uint32_t l3_bk(const uint32_t (&f)[M], int a) {  // flag of the byte K places before (K in 1..8)
    constexpr int q = K / 4, r = K % 4;
    if constexpr (r == 0) return f[a - q];
    else return (f[a - q] << (8 * r)) | (f[a - q - 1] >> (32 - 8 * r));
}
template <int K, int M>
__device__ __forceinline__ uint32_t l3_ak(const uint32_t (&f)[M], int a) {  // flag of the byte K places after (K in 1..8)
    constexpr int q = K / 4, r = K % 4;
    if constexpr (r == 0) return f[a + q];
    else return (f[a + q] >> (8 * r)) | (f[a + q + 1] << (32 - 8 * r));
}
template <int M>
__device__ __forceinline__ void l3_halo(uint32_t (&f)[M]) {  // own dwords [2 .. M-3] are set: fetch the neighbours'
    f[0] = lane_prev(f[M - 4]);
    f[1] = lane_prev(f[M - 3]);
    f[M - 2] = lane_next(f[2]);
    f[M - 1] = lane_next(f[3]);
}
// Ranks the piece starts of window bytes [lo, hi2) into ws.pstart (relative to lo; lo itself forced), hi2 = min(hi, und).
// false: the window is for llama3_start_mask (nothing written); `fallback`: for the literal matcher.
template <int LB, class WS>
__device__ __forceinline__ bool llama3_packed_starts(WS& ws, const SplitDev& sp, int skew, int wlen, int lo, int hi, bool at_end,
                                                     int& np, int& undecided, bool& fallback) {
    constexpr int LBy = 4 * LB, M = LB + 4, AL = LB + 1;  // own dwords at [2 .. AL]
    const int l = lane_id();
    const int off = kTextPad + skew + LBy * l;
    int nv = wlen - LBy * l;
    nv = nv < 0 ? 0 : (nv > LBy ? LBy : nv);
    uint32_t r[LB + 1], x[M], V[M], L[M], N[M], W[M], NL[M], SP[M], CT[M], AP[M];
    x[0] = x[1] = x[M - 2] = x[M - 1] = 0;
    AP[0] = AP[1] = AP[M - 2] = AP[M - 1] = 0;
    const int sh = lds_lane_dwords<LB>(&ws.text_w[0], off, r);
    uint32_t leads = 0;  // bit k: own byte k is a lead byte (>= 0xC0)
    bool odd = false;
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int have = nv - 4 * j;
        const uint32_t m = have >= 4 ? ~0u : (have <= 0 ? 0u : ((1u << (8 * have)) - 1u));
        const uint32_t xx = funnel_shr(r[j], r[j + 1], sh) & m;
        const int a = j + 2;
        x[a] = xx;
        V[a] = m & kB7;
        const uint32_t hi7 = xx & kB7, lo7 = xx & 0x7F7F7F7Fu;  // the packed compares want bytes below 0x80
        const uint32_t asc = ~hi7 & V[a];
        L[a] = swar_range(lo7 | 0x20202020u, 'a', 'z') & asc;
        N[a] = swar_range(lo7, '0', '9') & asc;
        SP[a] = swar_eq(lo7, 0x20) & asc;
        NL[a] = (swar_eq(lo7, '\n') | swar_eq(lo7, '\r')) & asc;
        W[a] = (SP[a] | swar_range(lo7, 9, 13)) & asc;
        AP[a] = swar_eq(lo7, 0x27) & asc;
        CT[a] = hi7 & ~((xx << 1) & kB7);                     // 10xxxxxx
        const uint32_t ld = hi7 & ((xx << 1) & kB7);           // 11xxxxxx
        const uint32_t t = ld >> 7;
        leads |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu) << (4 * j);
    }
    uint32_t ct_any = 0, nl_any = 0, ap_any = 0;
#pragma unroll
    for (int a = 2; a <= AL; ++a) { ct_any |= CT[a]; nl_any |= NL[a]; ap_any |= AP[a]; }
    // non-ASCII characters: class of each lead byte of the lane.  Up to six at a time (twelve bytes of two-byte characters)
    // with the table loads of all of them in flight together: one lead after the other, every character was two dependent
    // round trips, and a window of Greek or Cyrillic text waited for twelve of them.
    for (uint32_t rest = leads; __ballot(rest != 0);) {
        int kk[6];
        uint32_t cps[6], i1[6], nb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            kk[j] = rest ? __ffs(rest) - 1 : -1;
            rest &= rest - 1;   // (0 stays 0)
            cps[j] = 0x110000u;
            if (kk[j] >= 0) {
                const uint32_t b = (x[2 + (kk[j] >> 2)] >> (8 * (kk[j] & 3))) & 0xFFu;
                cps[j] = decode_lead(ws, skew, LBy * l + kk[j], b, wlen);
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) i1[j] = cps[j] < 0x110000u ? uint32_t(sp.uc_index[cps[j] >> 7]) : 0u;
#pragma unroll
        for (int j = 0; j < 6; ++j) nb[j] = cps[j] < 0x110000u ? uint32_t(sp.uc_blocks[i1[j] * 64 + ((cps[j] & 127) >> 1)]) : 0u;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (kk[j] < 0) continue;
            const uint32_t cls = ((cps[j] & 1) ? (nb[j] >> 4) : (nb[j] & 15u)) & 3u;
            if (cls == kClsN || cps[j] == 0x17Fu) odd = true;
            const uint32_t bit = 0x80u << (8 * (kk[j] & 3));
            if (cls == kClsL) L[2 + (kk[j] >> 2)] |= bit;
            if (cls == kClsS) W[2 + (kk[j] >> 2)] |= bit;
        }
    }
    fallback = __ballot(odd) != 0;
    if (fallback) return true;  // (the caller looks at fallback first)
    const bool any_ct = __ballot(ct_any != 0) != 0;
    // digit groups count from the chunk start: the digit in front of it is not part of the run
    uint32_t Nd[M];
#pragma unroll
    for (int a = 2; a <= AL; ++a) Nd[a] = N[a];
    if (lo > 0 && (lo - 1) / LBy == l) Nd[2 + ((lo - 1) % LBy >> 2)] &= ~(0x80u << (8 * ((lo - 1) & 3)));
    uint32_t nd_any = 0;
#pragma unroll
    for (int a = 2; a <= AL; ++a) nd_any |= Nd[a];
    l3_halo(V); l3_halo(L); l3_halo(N); l3_halo(Nd); l3_halo(W); l3_halo(NL); l3_halo(SP); l3_halo(CT);
    if (any_ct) {  // continuation bytes take the class of their lead byte (up to three of them behind it)
        uint32_t l2[M], w2[M];
#pragma unroll
        for (int a = 0; a < M; ++a) { l2[a] = L[a]; w2[a] = W[a]; }
#pragma unroll
        for (int a = 2; a <= AL; ++a) {
            const uint32_t c1 = CT[a], c2 = c1 & l3_bk<1>(CT, a), c3 = c2 & l3_bk<2>(CT, a);
            l2[a] |= (c1 & l3_bk<1>(L, a)) | (c2 & l3_bk<2>(L, a)) | (c3 & l3_bk<3>(L, a));
            w2[a] |= (c1 & l3_bk<1>(W, a)) | (c2 & l3_bk<2>(W, a)) | (c3 & l3_bk<3>(W, a));
        }
#pragma unroll
        for (int a = 2; a <= AL; ++a) { L[a] = l2[a]; W[a] = w2[a]; }
        l3_halo(L);
        l3_halo(W);
    }
    uint32_t O[M];
#pragma unroll
    for (int a = 0; a < M; ++a) O[a] = V[a] & ~(L[a] | N[a] | W[a]);
    // ---- bounded forms of the ripples
    uint32_t G[M], FE[M], LN[M];  // digit group starts; last byte of a line-break run that began right behind an O char;
                                  // the last line break of its white-space run
#pragma unroll
    for (int a = 0; a < M; ++a) G[a] = FE[a] = LN[a] = 0;
    bool far = false;
    const bool any_n = __ballot(nd_any != 0) != 0, any_nl = __ballot(nl_any != 0) != 0;
    if (any_n && sp.l3_digits1) {
#pragma unroll
        for (int a = 2; a <= AL; ++a) G[a] = N[a];   // `\p{N}`: every digit starts a piece
    } else if (any_n) {
#pragma unroll
        for (int a = 2; a <= AL; ++a) {
            const uint32_t n1 = l3_bk<1>(Nd, a), n2 = l3_bk<2>(Nd, a), n3 = l3_bk<3>(Nd, a), n4 = l3_bk<4>(Nd, a), n5 = l3_bk<5>(Nd, a),
                           n6 = l3_bk<6>(Nd, a), n7 = l3_bk<7>(Nd, a), n8 = l3_bk<8>(Nd, a);
            const uint32_t r2 = Nd[a] & n1 & n2, r5 = r2 & n3 & n4 & n5;  // a digit here and at the 2 / 5 bytes before
            const uint32_t s0 = Nd[a] & ~n1;                              // run start here
            const uint32_t s3 = r2 & n3 & ~n4, s6 = r5 & n6 & ~n7;        // run start 3 / 6 bytes back
            G[a] = s0 | s3 | s6;
            if (r5 & n6 & n7 & n8) far = true;                            // nine digits in a row
        }
    }
    if (any_nl) {
#pragma unroll
        for (int a = 2; a <= AL; ++a) {
            const uint32_t e = NL[a] & ~l3_ak<1>(NL, a);                  // last byte of a line-break run
            const uint32_t b1 = l3_bk<1>(NL, a), b2 = l3_bk<2>(NL, a), b3 = l3_bk<3>(NL, a), b4 = l3_bk<4>(NL, a);
            const uint32_t o1 = l3_bk<1>(O, a), o2 = l3_bk<2>(O, a), o3 = l3_bk<3>(O, a), o4 = l3_bk<4>(O, a);
            FE[a] = e & ((~b1 & o1) | (b1 & ~b2 & o2) | (b1 & b2 & ~b3 & o3) | (b1 & b2 & b3 & ~b4 & o4));
            if (e & b1 & b2 & b3 & b4) far = true;                        // five line breaks in a row
            // a later line break in the same white-space run, within four bytes
            const uint32_t w1 = l3_ak<1>(W, a), w2 = w1 & l3_ak<2>(W, a), w3 = w2 & l3_ak<3>(W, a), w4 = w3 & l3_ak<4>(W, a);
            const uint32_t later = (w1 & l3_ak<1>(NL, a)) | (w2 & l3_ak<2>(NL, a)) | (w3 & l3_ak<3>(NL, a)) | (w4 & l3_ak<4>(NL, a));
            LN[a] = NL[a] & ~later;
            if (NL[a] & w4 & ~later) far = true;                          // the run goes on: cannot tell from here
        }
    }
    if (__ballot(far)) return false;
    l3_halo(FE);
    l3_halo(LN);
    // ---- local rules (llama3_start_mask's)
    uint32_t sO[M], f1[M], f2[M], TK[M];
#pragma unroll
    for (int a = 0; a < M; ++a) sO[a] = f1[a] = f2[a] = TK[a] = 0;
    const bool any_ap = __ballot(ap_any != 0) != 0;
    x[M - 2] = any_ap ? lane_next(x[2]) : 0u;
#pragma unroll
    for (int a = 2; a <= AL; ++a) {
        const uint32_t pO = l3_bk<1>(O, a), pSP = l3_bk<1>(SP, a);
        sO[a] = O[a] & ~pO;
        if (any_ap && AP[a]) {
            // the two bytes behind the apostrophe (raw bytes of this dword and the next one; a non-ASCII byte matches no letter)
            const uint32_t nxt = x[a + 1];
            const uint32_t n1 = ((x[a] >> 8) | (nxt << 24)), n2 = ((x[a] >> 16) | (nxt << 16));
            const uint32_t m1 = ~n1 & kB7, m2 = ~n2 & kB7;  // ASCII bytes only
            const uint32_t l1 = (n1 & 0x7F7F7F7Fu) | 0x20202020u, l2 = (n2 & 0x7F7F7F7Fu) | 0x20202020u;
            const uint32_t c1 = AP[a] & m1 & (swar_range(l1, 's', 't') | swar_eq(l1, 'm') | swar_eq(l1, 'd'));
            const uint32_t c2 = AP[a] & m1 & m2 & (((swar_eq(l1, 'r') | swar_eq(l1, 'v')) & swar_eq(l2, 'e')) | (swar_eq(l1, 'l') & swar_eq(l2, 'l')));
            f1[a] = c1 & sO[a] & ~pSP;
            f2[a] = c2 & sO[a] & ~pSP & ~f1[a];
        }
    }
    if (any_ap) {
        l3_halo(f1);
        l3_halo(f2);
    }
    // O runs of one char that go in front of the letters behind them
#pragma unroll
    for (int a = 2; a <= AL; ++a) {
        const uint32_t endO = O[a] & ~l3_ak<1>(O, a);
        const uint32_t eo1 = l3_ak<1>(O, a) & ~l3_ak<2>(O, a), eo2 = l3_ak<2>(O, a) & ~l3_ak<3>(O, a), eo3 = l3_ak<3>(O, a) & ~l3_ak<4>(O, a);
        const uint32_t c1 = l3_ak<1>(CT, a), c2 = c1 & l3_ak<2>(CT, a), c3 = c2 & l3_ak<3>(CT, a);
        const uint32_t lead_of_last = ~CT[a] & (endO | (c1 & eo1) | (c2 & eo2) | (c3 & eo3));   // to_lead(endO)
        const uint32_t single_o = sO[a] & lead_of_last;
        TK[a] = single_o & ~l3_bk<1>(SP, a) & ~(f1[a] | f2[a]);
    }
    l3_halo(TK);
    if (any_ct) {  // spread over the char's continuation bytes
        uint32_t t2[M];
#pragma unroll
        for (int a = 2; a <= AL; ++a) {
            const uint32_t c1 = CT[a], c2 = c1 & l3_bk<1>(CT, a), c3 = c2 & l3_bk<2>(CT, a);
            t2[a] = TK[a] | (c1 & l3_bk<1>(TK, a)) | (c2 & l3_bk<2>(TK, a)) | (c3 & l3_bk<3>(TK, a));
        }
#pragma unroll
        for (int a = 2; a <= AL; ++a) TK[a] = t2[a];
        TK[1] = lane_prev(TK[AL]);
    }
    uint32_t flags = 0;
    uint32_t nonw = 0;
    uint32_t ln_only = 0;   // l3_tail_ws: starts that exist for no other reason than "behind the run's last line break"
    const bool tail_ws = sp.l3_tail_ws && at_end && any_nl;
#pragma unroll
    for (int a = 2; a <= AL; ++a) {
        const uint32_t pL = l3_bk<1>(L, a), pW = l3_bk<1>(W, a), pO = l3_bk<1>(O, a), pSP = l3_bk<1>(SP, a), pNL = l3_bk<1>(NL, a);
        const uint32_t sL = L[a] & ~pL, sW = W[a] & ~pW;
        const uint32_t f12[2] = {f1[a - 1] | f2[a - 1], f1[a] | f2[a]};
        const uint32_t supL = sL & ((pW & ~pNL) | ((TK[a] << 8) | (TK[a - 1] >> 24)) | ((f12[1] << 8) | (f12[0] >> 24)));
        const uint32_t supO = sO[a] & pSP;
        const uint32_t supW = sW & pO & NL[a];
        const uint32_t b_abs = l3_bk<1>(FE, a) & W[a];
        const uint32_t b_ln = l3_bk<1>(LN, a) & W[a];
        // last char of a white-space run that is followed by a char
        const uint32_t aV1 = l3_ak<1>(V, a), aV2 = l3_ak<2>(V, a), aV3 = l3_ak<3>(V, a), aV4 = l3_ak<4>(V, a);
        const uint32_t ew0 = W[a] & ~l3_ak<1>(W, a) & aV1, ew1 = l3_ak<1>(W, a) & ~l3_ak<2>(W, a) & aV2,
                       ew2 = l3_ak<2>(W, a) & ~l3_ak<3>(W, a) & aV3, ew3 = l3_ak<3>(W, a) & ~l3_ak<4>(W, a) & aV4;
        const uint32_t c1 = l3_ak<1>(CT, a), c2 = c1 & l3_ak<2>(CT, a), c3 = c2 & l3_ak<3>(CT, a);
        const uint32_t last_w = ~CT[a] & (ew0 | (c1 & ew1) | (c2 & ew2) | (c3 & ew3));
        const uint32_t b_last = last_w & W[a] & ~NL[a] & pW & ~pNL;
        const uint32_t b_con = ((f1[a] << 16) | (f1[a - 1] >> 16)) | ((f2[a] << 24) | (f2[a - 1] >> 8));
        const uint32_t st_else = ((sL & ~supL) | G[a] | (sO[a] & ~supO) | (sW & ~supW) | b_abs | b_last | b_con) & V[a] & ~CT[a];
        const uint32_t st = st_else | (b_ln & V[a] & ~CT[a]);
        const uint32_t t = st >> 7;
        flags |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu) << (4 * (a - 2));
        if (tail_ws) {
            const uint32_t q = (st & ~st_else) >> 7;
            ln_only |= ((q | (q >> 7) | (q >> 14) | (q >> 21)) & 0xFu) << (4 * (a - 2));
        }
        const uint32_t u = (V[a] & ~W[a]) >> 7;
        nonw |= ((u | (u >> 7) | (u >> 14) | (u >> 21)) & 0xFu) << (4 * (a - 2));
    }
    if (tail_ws) {
        // `\s++$`: the white-space run that ends the string is one piece -- no start behind its last line break
        const unsigned long long nwl = __ballot(nonw != 0);
        int nonw_end = 0;  // window position behind the last byte that is not white space
        if (nwl) {
            const int hl = 63 - __clzll(nwl);
            nonw_end = hl * LBy + (32 - __clz(wave_readlane(int(nonw), hl)));
        }
        int k0 = nonw_end - LBy * l;   // the lane's bytes from k0 on belong to that run
        k0 = k0 < 0 ? 0 : (k0 > LBy ? LBy : k0);
        const uint32_t run = k0 >= 32 ? 0u : ~((1u << k0) - 1u);
        flags &= ~(ln_only & run);
    }
    // ---- how far the window decides (llama3_start_mask's rule)
    undecided = 0x7FFFFFFF;
    if (!at_end) {
        undecided = wlen > 8 ? wlen - 8 : 0;
        const unsigned long long nwl = __ballot(nonw != 0);
        int nonw_end = 0;  // window position behind the last byte that is not white space
        if (nwl) {
            const int hl = 63 - __clzll(nwl);
            nonw_end = hl * LBy + (32 - __clz(wave_readlane(int(nonw), hl)));
        }
        if (nonw_end < wlen && nonw_end + 1 < undecided) undecided = nonw_end + 1;
    }
    const int hi2 = hi < undecided ? hi : undecided;
    np = 0;
    if (hi2 <= lo) return true;
    int k_lo = lo - LBy * l, k_hi = hi2 - LBy * l;
    k_lo = k_lo < 0 ? 0 : (k_lo > LBy ? LBy : k_lo);
    k_hi = k_hi < 0 ? 0 : (k_hi > LBy ? LBy : k_hi);
    uint32_t fl = flags & ((1u << k_hi) - 1u) & ~((1u << k_lo) - 1u);
    if (lo / LBy == l) fl |= 1u << (lo % LBy);
    const int cnt = __popc(fl);
    const int incl = wave_incl_sum(cnt);
    int at = incl - cnt;
    const int first = LBy * l - lo;
    while (fl) {
        ws.pstart[at++] = uint16_t(first + __ffs(fl) - 1);
        fl &= fl - 1;
    }
    np = wave_readlane(incl, kWave - 1);
    return true;
}

// The class patterns (kSplitWhitespace / kSplitBertPunct / kSplitBertWords) on an ASCII window, packed bytes, LB dwords
// per lane: same result as class_start_mask + the ranking loop of scan_string -- pstart entries of window bytes
// [lo, hi) relative to lo, kPieceDropped on the pieces that are not emitted.  false (wave-uniform): not ASCII.
template <int LB, class WS>
__device__ __forceinline__ bool class_packed_starts(WS& ws, const SplitDev& sp, int skew, int wlen, int lo, int hi, int& np) {
    constexpr int LBy = 4 * LB;
    const int l = lane_id();
    const int off = kTextPad + skew + LBy * l;
    int nv = wlen - LBy * l;
    nv = nv < 0 ? 0 : (nv > LBy ? LBy : nv);
    uint32_t x[LB + 1], V[LB + 1], r[LB + 1];
    const int sh = lds_lane_dwords<LB>(&ws.text_w[0], off, r);
    uint32_t any = 0;
#pragma unroll
    for (int j = 1; j <= LB; ++j) {
        const int have = nv - 4 * (j - 1);
        const uint32_t m = have >= 4 ? ~0u : (have <= 0 ? 0u : ((1u << (8 * have)) - 1u));
        x[j] = funnel_shr(r[j - 1], r[j], sh) & m;
        V[j] = m & kB7;
        any |= x[j];
    }
    if (__ballot((any & kB7) != 0)) return false;
    uint32_t S[LB + 1], P[LB + 1];
#pragma unroll
    for (int j = 1; j <= LB; ++j) {
        S[j] = (swar_eq(x[j], 0x20) | swar_range(x[j], 9, 13)) & V[j];
        P[j] = (swar_range(x[j], 0x21, 0x2F) | swar_range(x[j], 0x3A, 0x40) | swar_range(x[j], 0x5B, 0x60) |
                swar_range(x[j], 0x7B, 0x7E)) & V[j];   // bert_delimiter() below 0x80
    }
    S[0] = lane_prev(S[LB]);
    P[0] = lane_prev(P[LB]);
    uint32_t fl = 0, dr = 0;
#pragma unroll
    for (int j = 1; j <= LB; ++j) {
        const uint32_t pS = swar_before<1>(S[j - 1], S[j]), pP = swar_before<1>(P[j - 1], P[j]);
        uint32_t st, m;
        if (sp.kind == kSplitWhitespace) {
            m = S[j];
            st = S[j] ^ pS;
        } else if (sp.kind == kSplitBertPunct) {
            m = P[j];
            st = P[j] | pP;
        } else {
            m = S[j];
            st = (S[j] ^ pS) | P[j] | pP;
        }
        const uint32_t d = sp.drop == 1 ? m : (sp.drop == 2 ? ~m : 0u);
        const uint32_t t = (st & V[j]) >> 7, u = (d & V[j]) >> 7;
        fl |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu) << (4 * (j - 1));
        dr |= ((u | (u >> 7) | (u >> 14) | (u >> 21)) & 0xFu) << (4 * (j - 1));
    }
    int k_lo = lo - LBy * l, k_hi = hi - LBy * l;
    k_lo = k_lo < 0 ? 0 : (k_lo > LBy ? LBy : k_lo);
    k_hi = k_hi < 0 ? 0 : (k_hi > LBy ? LBy : k_hi);
    fl &= ((1u << k_hi) - 1u) & ~((1u << k_lo) - 1u);
    if (lo / LBy == l) fl |= 1u << (lo % LBy);
    const int cnt = __popc(fl);
    const int incl = wave_incl_sum(cnt);
    int at = incl - cnt;
    const int first = LBy * l - lo;
    while (fl) {
        const int bit = __ffs(fl) - 1;
        ws.pstart[at++] = uint16_t((first + bit) | (((dr >> bit) & 1u) ? kPieceDropped : 0));
        fl &= fl - 1;
    }
    np = wave_readlane(incl, kWave - 1);
    return true;
}

}  // namespace ovtk
#include "fam_literal.hpp"
namespace ovtk {

// Scans string `str` (slen bytes) and hands complete pieces to the caller chunk by chunk.
//   on_chunk(np, c0, w0, skew): pstart[0..np] (positions relative to c0, pstart[np] = end of the last piece)
//                               describe np complete pieces; the LDS text covers them (string byte p at
//                               text_bytes(ws)[p - w0 + skew]).
//                               class patterns: entries carry kPieceDropped when the piece is not to be emitted
//                               (read positions through kPiecePosMask).
//   on_long(b, e, dropped):     a piece of more than kChunk bytes, not staged in LDS.
// [buf, buf_end): the chars tensor the string lies in (what may be read).
// Wave-uniform; every lane must call it with the same arguments.
// SCANNER: the kernel is compiled for the Llama-3 pattern only (1) / for the GPT-2 family and the class patterns (0) -- the two share no
// scanner code, and either one alone fits the register budget of the lookup kernel -- / for the families of span_fam.hpp, matched
// literally by lane 0 (2: the rows lookup_span_kernel leaves to the generic kernel, calls of a few rows).
constexpr int kScanGeneric = 0, kScanLlama3 = 1, kScanFamLiteral = 2;
template <int SCANNER, class OnChunk, class OnLong>
__device__ __forceinline__ void scan_string(WaveScratch& ws, const SplitDev& sp, const uint8_t* str, int slen, const uint8_t* buf,
                                            const uint8_t* buf_end, OnChunk&& on_chunk, OnLong&& on_long) {
    const bool digits = sp.kind == kSplitGpt2Digits;
    const int l = lane_id();
    int c0 = 0;
    while (c0 < slen) {
        const int w0 = c0 > kLeftHalo ? c0 - kLeftHalo : 0;
        // The window [w0, w1) never exceeds kChunk bytes.  Balanced chunks: a 600-byte string is scanned as
        // 2 x 300, not 492 + 108 (fuller lane batches).
        const int rest = slen - c0, room = kChunk - (c0 - w0);
        int qlim = slen;
        if (rest > room) {
            const int budget = room - kRightHalo;
            const int nchunks = (rest + budget - 1) / budget;
            qlim = c0 + (rest + nchunks - 1) / nchunks;
        }
        const int w1 = (qlim + kRightHalo < slen) ? qlim + kRightHalo : slen;
        wave_sync();  // previous consumers of the LDS window are done
        const int skew = stage_window(ws, str, slen, w0, w1, buf, buf_end);
        wave_sync();
        // rank the starts of [c0, qlim) (window bytes [lo, hi)); c0 itself is a start by construction
        const int lo = c0 - w0, hi = qlim - w0;
        int np = 0;
        if constexpr (SCANNER == kScanFamLiteral) {
            // lane 0 matches from the chunk start, as far as the pieces stay inside the staged text
            int n_seq = 0, p = c0;
            if (l == 0) {
                while (p < slen && n_seq < kChunk) {
                    const int e = sp.family == kFamDs3 ? ds3_match_end(sp, str, slen, p) : o200k_match_end(sp, str, slen, p);
                    if (e > w1) break;
                    ws.pstart[n_seq++] = uint16_t(p - c0);
                    p = e;
                }
                ws.pstart[n_seq] = uint16_t(p - c0);
                if (n_seq == 0) p = sp.family == kFamDs3 ? ds3_match_end(sp, str, slen, c0) : o200k_match_end(sp, str, slen, c0);  // a piece longer than the window
            }
            n_seq = wave_readlane(n_seq, 0);
            p = wave_readlane(p, 0);
            wave_sync();
            if (n_seq) on_chunk(n_seq, c0, w0, skew);
            else on_long(c0, p, false);
            c0 = p;
            (void)lo; (void)hi; (void)np; (void)digits;
            continue;
        }
        if constexpr (SCANNER == kScanLlama3) {
            int und = 0;
            bool seq = false;
            // every lane its own 12 bytes (packed bytes); the windows that form does not cover: lane w = 64-byte word w (ballots)
            const bool packed = w1 - w0 <= 512 ? llama3_packed_starts<2>(ws, sp, skew, w1 - w0, lo, hi, w1 == slen, np, und, seq)
                                               : llama3_packed_starts<3>(ws, sp, skew, w1 - w0, lo, hi, w1 == slen, np, und, seq);
#ifdef OVTK_SIMT_EMULATOR
            if (packed && !seq) {  // the emulator build checks the packed form against the ballot form on every window
                int und0 = 0;
                bool seq0 = false;
                const Mask start0 = llama3_start_mask(ws, sp, skew, w1 - w0, lo, w1 == slen, und0, seq0);
                const int h0 = hi < und0 ? hi : und0;
                int np0 = 0;
                bool same = !seq0 && und0 == und;
                if (same && h0 > lo) {
                    for (int w = lo >> 6; w * 64 < h0; ++w) {
                        Mask m = wave_readlane(start0, w);
                        if (w == (lo >> 6)) m = (m & ~((1ull << (lo & 63)) - 1ull)) | (1ull << (lo & 63));
                        if (h0 - w * 64 < 64) m &= (1ull << (h0 - w * 64)) - 1ull;
                        if (((m >> l) & 1ull) && ws.pstart[np0 + rank_below(m)] != uint16_t(w * 64 + l - lo)) same = false;
                        np0 += __popcll(m);
                    }
                    if (np0 != np) same = false;
                }
                if (__ballot(!same)) {
                    if (l == 0) printf("llama3 packed starts differ: wlen %d lo %d hi %d und %d/%d np %d/%d seq0 %d\n", w1 - w0, lo, hi, und, und0, np, np0, int(seq0));
                    __builtin_trap();
                }
            }
#endif
            if (!packed) {
                np = 0;
                const Mask start = llama3_start_mask(ws, sp, skew, w1 - w0, lo, w1 == slen, und, seq);
                const int h2 = hi < und ? hi : und;
                if (!seq && h2 > lo) {
                    for (int w = lo >> 6; w * 64 < h2; ++w) {
                        Mask m = wave_readlane(start, w);
                        if (w == (lo >> 6)) m = (m & ~((1ull << (lo & 63)) - 1ull)) | (1ull << (lo & 63));
                        if (h2 - w * 64 < 64) m &= (1ull << (h2 - w * 64)) - 1ull;
                        if ((m >> l) & 1ull) ws.pstart[np + rank_below(m)] = uint16_t(w * 64 + l - lo);
                        np += __popcll(m);
                    }
                }
            }
            const int hi2 = hi < und ? hi : und;
            const bool whole = qlim == slen && hi2 == hi;  // every piece that starts in the window also ends in it
            if (!seq && hi2 > lo) {
                wave_sync();
                if (whole) {
                    if (l == 0) ws.pstart[np] = uint16_t(slen - c0);
                    wave_sync();
                    on_chunk(np, c0, w0, skew);
                    c0 = slen;
                    continue;
                }
                if (np >= 2) {  // the last piece may continue behind the window: the next chunk starts with it
                    on_chunk(np - 1, c0, w0, skew);
                    c0 += int(ws.pstart[np - 1] & kPiecePosMask);
                    continue;
                }
            }
            // Nothing the masks could decide (a window the bit-parallel rules do not cover, or one piece that fills
            // it): lane 0 matches literally from the chunk start, as far as the pieces stay inside the staged text.
            int n_seq = 0, p = c0;
            if (l == 0) {
                while (p < slen && n_seq < kChunk) {
                    const int e = llama3_match_end(sp, str, slen, p);
                    if (e > w1) break;
                    ws.pstart[n_seq++] = uint16_t(p - c0);
                    p = e;
                }
                ws.pstart[n_seq] = uint16_t(p - c0);
                if (n_seq == 0) p = llama3_match_end(sp, str, slen, c0);  // a piece longer than the window
            }
            n_seq = wave_readlane(n_seq, 0);
            p = wave_readlane(p, 0);
            wave_sync();
            if (n_seq) on_chunk(n_seq, c0, w0, skew);
            else on_long(c0, p, false);
            c0 = p;
            continue;
        }
        if (sp.kind >= kSplitWhitespace) {
            const int wl = w1 - w0;
            if (wl <= 256 ? class_packed_starts<1>(ws, sp, skew, wl, lo, hi, np)
                          : (wl <= 512 ? class_packed_starts<2>(ws, sp, skew, wl, lo, hi, np)
                                       : class_packed_starts<kLaneDwords>(ws, sp, skew, wl, lo, hi, np))) {
                // ASCII window: the packed-byte scanner filled pstart
            } else {
            Mask start, dropped;
            class_start_mask(ws, sp, skew, w1 - w0, start, dropped);
            for (int w = lo >> 6; w * 64 < hi; ++w) {
                Mask m = wave_readlane(start, w);
                const Mask d = wave_readlane(dropped, w);
                if (w == (lo >> 6)) m = (m & ~((1ull << (lo & 63)) - 1ull)) | (1ull << (lo & 63));
                if (hi - w * 64 < 64) m &= (1ull << (hi - w * 64)) - 1ull;
                if ((m >> l) & 1ull)
                    ws.pstart[np + rank_below(m)] = uint16_t((w * 64 + l - lo) | (((d >> l) & 1ull) ? kPieceDropped : 0));
                np += __popcll(m);
            }
            }
        } else if (w1 - w0 <= 64 * 4 * (kLaneDwords - 1) ? gpt2_packed_starts<kLaneDwords - 1>(ws, skew, w1 - w0, digits, lo, hi, np)
                                                          : gpt2_packed_starts<kLaneDwords>(ws, skew, w1 - w0, digits, lo, hi, np)) {
            // ASCII window: the packed-byte scanner filled pstart (with as few dwords per lane as cover the window)
        } else {
            const Mask start = gpt2_start_mask(ws, sp, skew, w1 - w0, digits);
            for (int w = lo >> 6; w * 64 < hi; ++w) {
                Mask m = wave_readlane(start, w);
                if (w == (lo >> 6)) m = (m & ~((1ull << (lo & 63)) - 1ull)) | (1ull << (lo & 63));
                if (hi - w * 64 < 64) m &= (1ull << (hi - w * 64)) - 1ull;
                if ((m >> l) & 1ull) ws.pstart[np + rank_below(m)] = uint16_t(w * 64 + l - lo);
                np += __popcll(m);
            }
        }
        wave_sync();
        if (qlim == slen) {  // the string ends in this window: every piece is complete
            if (l == 0) ws.pstart[np] = uint16_t(slen - c0);  // < 0x8000
            wave_sync();
            on_chunk(np, c0, w0, skew);
            c0 = slen;
        } else if (np >= 2) {  // the last piece may continue: restart the next chunk at its start
            on_chunk(np - 1, c0, w0, skew);
            c0 += int(ws.pstart[np - 1] & kPiecePosMask);
        } else {
            // One piece of >= kChunk bytes: look for its end window by window.
            const bool long_dropped = (ws.pstart[0] & kPieceDropped) != 0;
            int e = qlim;
            bool found = false;
            while (!found && e < slen) {
                const int lw0 = e - kLeftHalo;
                const int lspan = kChunk - kLeftHalo - kRightHalo;
                const int lq = (e + lspan < slen) ? e + lspan : slen;
                const int lw1 = (lq + kRightHalo < slen) ? lq + kRightHalo : slen;
                wave_sync();
                const int lskew = stage_window(ws, str, slen, lw0, lw1, buf, buf_end);
                wave_sync();
                Mask ls, ldrop;
                if (sp.kind >= kSplitWhitespace) class_start_mask(ws, sp, lskew, lw1 - lw0, ls, ldrop);
                else ls = gpt2_start_mask(ws, sp, lskew, lw1 - lw0, digits);
                const int llo = e - lw0, lhi = lq - lw0;
                for (int w = llo >> 6; w * 64 < lhi && !found; ++w) {
                    Mask m = wave_readlane(ls, w);
                    if (w == (llo >> 6)) m &= ~((1ull << (llo & 63)) - 1ull);
                    if (lhi - w * 64 < 64) m &= (1ull << (lhi - w * 64)) - 1ull;
                    if (m) {
                        e = lw0 + w * 64 + __ffsll(m) - 1;
                        found = true;
                    }
                }
                if (!found) e = lq;
            }
            on_long(c0, e, long_dropped);
            c0 = e;
        }
    }
}

}  // namespace ovtk
