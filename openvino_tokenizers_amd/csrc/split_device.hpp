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
// Evaluation is bit-parallel.  One wave scans one string: the window (<= 512 bytes + halos) is
// staged in LDS with coalesced dword loads; for each 64-byte word of the window every lane
// classifies ONE byte (ASCII arithmetically, other code points through the two-level Unicode
// property table generated from PCRE2 itself) and the per-byte predicates become 64-bit masks by
// wave ballot.  Lane w then owns the masks of word w, and the rules above are ~40 AND/OR/shift
// operations on those masks -- all words of the window at once, neighbouring words reached with
// one DPP lane shift.  Piece starts are finally turned into positions by popcount ranking.
#pragma once

#include "device_common.hpp"
#include "tables.hpp"

namespace ovtk {

enum SplitKind : int32_t {
    kSplitGpt2 = 0,        // byte_level_splitter()
    kSplitGpt2Digits = 1,  // byte_level_splitter(individual_digits=True)
};

struct SplitDev {
    int32_t kind;
    const uint16_t* uc_index;  // [0x110000 >> 7]
    const uint8_t* uc_blocks;  // [n_blocks * 64]
};

constexpr int kChunk = 512;               // text bytes whose piece starts are decided per pass
constexpr int kLeftHalo = 8;
constexpr int kRightHalo = 12;
constexpr int kWinBytes = kChunk + 32;    // halo + alignment skew, multiple of 4
constexpr int kWinWords = (kChunk + kLeftHalo + kRightHalo + 63) / 64;  // 64-byte mask words per window (9)

constexpr uint8_t kClsO = 0, kClsL = 1, kClsN = 2, kClsS = 3;

// Per-wave LDS working set of the lookup / split kernels.
struct WaveScratch {
    uint32_t text_w[kWinBytes / 4];
    uint16_t pstart[kChunk + 2];
};

__device__ __forceinline__ const uint8_t* text_bytes(const WaveScratch& ws) {
    return reinterpret_cast<const uint8_t*>(ws.text_w);
}

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

// Stage bytes [w0, w1) of the string at `str` (global) into ws.text_w.  Returns the skew: string
// byte p lives at text_bytes(ws)[p - w0 + skew].  Whole dwords are fetched where they lie inside
// the string's own buffer range [0, slen); edge dwords are assembled from byte loads.
__device__ __forceinline__ int stage_window(WaveScratch& ws, const uint8_t* str, int slen, int w0, int w1) {
    const uint8_t* g = str + w0;
    const int skew = int(reinterpret_cast<uintptr_t>(g) & 3);
    const uint8_t* ga = g - skew;
    const int nwords = (skew + (w1 - w0) + 3) >> 2;
    const int lo = skew - w0, hi = slen - w0 + skew;  // the string's own bytes, as offsets from ga
    for (int k = lane_id(); k < nwords; k += kWave) {
        const int b0 = k * 4;
        uint32_t w;
        if (b0 >= lo && b0 + 4 <= hi) {
            w = *reinterpret_cast<const uint32_t*>(ga + b0);
        } else {
            w = 0;
            for (int j = 0; j < 4; ++j)
                if (b0 + j >= lo && b0 + j < hi) w |= uint32_t(ga[b0 + j]) << (8 * j);
        }
        ws.text_w[k] = w;
    }
    return skew;
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
__device__ __forceinline__ Mask gpt2_start_mask(const WaveScratch& ws, const SplitDev& sp, int skew, int wlen, bool digits) {
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
            int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
            uint32_t cp = b & (0xFFu >> (n + 1));
            if (i + n > wlen) n = wlen - i;
            for (int j = 1; j < n; ++j) cp = (cp << 6) | (t[i + j] & 0x3Fu);
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

// Scans string `str` (slen bytes) and hands complete pieces to the caller chunk by chunk.
//   on_chunk(np, c0, w0, skew): pstart[0..np] (positions relative to c0, pstart[np] = end of the last piece)
//                               describe np complete pieces; the LDS text covers them (string byte p at
//                               text_bytes(ws)[p - w0 + skew]).
//   on_long(b, e):              a piece of more than kChunk bytes, not staged in LDS.
// Wave-uniform; every lane must call it with the same arguments.
template <class OnChunk, class OnLong>
__device__ __forceinline__ void scan_string(WaveScratch& ws, const SplitDev& sp, const uint8_t* str, int slen,
                                            OnChunk&& on_chunk, OnLong&& on_long) {
    const bool digits = sp.kind == kSplitGpt2Digits;
    const int l = lane_id();
    int c0 = 0;
    while (c0 < slen) {
        const int w0 = c0 > kLeftHalo ? c0 - kLeftHalo : 0;
        // Balanced chunks: a 600-byte string is scanned as 2 x 300, not 512 + 88 (fuller lane batches).
        const int rest = slen - c0;
        const int nchunks = (rest + kChunk - 1) / kChunk;
        const int qlim = nchunks <= 1 ? slen : c0 + (rest + nchunks - 1) / nchunks;
        const int w1 = (qlim + kRightHalo < slen) ? qlim + kRightHalo : slen;
        wave_sync();  // previous consumers of the LDS window are done
        const int skew = stage_window(ws, str, slen, w0, w1);
        wave_sync();
        const Mask start = gpt2_start_mask(ws, sp, skew, w1 - w0, digits);
        // rank the starts of [c0, qlim) (window bits [c0 - w0, qlim - w0)); c0 itself is a start by construction
        const int lo = c0 - w0, hi = qlim - w0;
        int np = 0;
        for (int w = lo >> 6; w * 64 < hi; ++w) {
            Mask m = wave_readlane(start, w);
            if (w == (lo >> 6)) m = (m & ~((1ull << (lo & 63)) - 1ull)) | (1ull << (lo & 63));
            if (hi - w * 64 < 64) m &= (1ull << (hi - w * 64)) - 1ull;
            if ((m >> l) & 1ull) ws.pstart[np + __popcll(m & lanemask_lt())] = uint16_t(w * 64 + l - lo);
            np += __popcll(m);
        }
        wave_sync();
        if (qlim == slen) {  // the string ends in this window: every piece is complete
            if (l == 0) ws.pstart[np] = uint16_t(slen - c0);
            wave_sync();
            on_chunk(np, c0, w0, skew);
            c0 = slen;
        } else if (np >= 2) {  // the last piece may continue: restart the next chunk at its start
            on_chunk(np - 1, c0, w0, skew);
            c0 += int(ws.pstart[np - 1]);
        } else {
            // One piece of >= kChunk bytes: look for its end window by window.
            int e = qlim;
            bool found = false;
            while (!found && e < slen) {
                const int lw0 = e - kLeftHalo;
                const int lq = (e + kChunk < slen) ? e + kChunk : slen;
                const int lw1 = (lq + kRightHalo < slen) ? lq + kRightHalo : slen;
                wave_sync();
                const int lskew = stage_window(ws, str, slen, lw0, lw1);
                wave_sync();
                const Mask ls = gpt2_start_mask(ws, sp, lskew, lw1 - lw0, digits);
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
            on_long(c0, e);
            c0 = e;
        }
    }
}

}  // namespace ovtk
