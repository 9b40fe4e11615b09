// split_device.hpp -- pre-tokenizer scanners (the device side of RegexSplit, src/regex_split.cpp:222-314).
//
// PCRE2 is not run on the GPU.  For the pattern families the reference's converter emits, a match
// that starts at p depends only on the text to the right of p (no look-behind), and every
// position matches something, so the sequence of isolate-mode pieces is determined by a LOCAL
// predicate "a piece starts at byte q" over a few neighbouring code points.  One wave scans one
// string: the window is staged in LDS with coalesced dword loads, every lane classifies one byte
// per step (two-level Unicode property table generated from PCRE2 itself), and __ballot +
// popcount compacts the piece starts.
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

constexpr uint8_t kClsO = 0, kClsL = 1, kClsN = 2, kClsS = 3;
constexpr uint8_t kClsMask = 3, kClsPunct = 4, kCharStart = 8;

// Per-wave LDS working set of the encode / split kernels.
struct WaveScratch {
    uint32_t text_w[kWinBytes / 4];
    uint8_t cls[kWinBytes];
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

// Classify bytes [w0, w1): cls[i] (i relative to w0, no skew) = class bits of the char the byte
// belongs to, | kCharStart on its first byte.  Continuation bytes whose lead byte lies before the
// window keep class O (never consulted: the left halo is wide enough).
__device__ __forceinline__ void classify_window(WaveScratch& ws, const SplitDev& sp, const uint8_t* ascii_cls,
                                                int skew, int wlen) {
    const uint8_t* t = text_bytes(ws) + skew;
    for (int i = lane_id(); i < wlen; i += kWave)
        if ((t[i] & 0xC0) == 0x80) ws.cls[i] = kClsO;
    wave_sync();
    for (int i = lane_id(); i < wlen; i += kWave) {
        const uint32_t b = t[i];
        if (b < 0x80) {
            ws.cls[i] = ascii_cls[b] | kCharStart;
        } else if ((b & 0xC0) != 0x80) {
            int n = b >= 0xF0 ? 4 : (b >= 0xE0 ? 3 : 2);
            uint32_t cp = b & (0xFFu >> (n + 1));
            if (i + n > wlen) n = wlen - i;  // truncated at the window edge (right halo covers real chars)
            for (int j = 1; j < n; ++j) cp = (cp << 6) | (t[i + j] & 0x3Fu);
            const uint8_t c = uint8_t(uc_nibble(sp, cp) & 7);
            ws.cls[i] = c | kCharStart;
            for (int j = 1; j < n; ++j)
                if ((t[i + j] & 0xC0) == 0x80) ws.cls[i + j] = c; else break;
        }
    }
    wave_sync();
}

// GPT-2 family piece-start predicate for string position q (a char start), see file header.
// t / cls are indexed by string position (the caller passes pointers already shifted by w0/skew).
__device__ __forceinline__ bool gpt2_fires(const uint8_t* t, const uint8_t* cls, int a, int nletters, int slen) {
    if (t[a] != 0x27) return false;
    if (a > 0 && ((cls[a - 1] & kClsMask) == kClsO || t[a - 1] == 0x20)) return false;
    if (a + nletters > slen - 1) return false;
    const uint8_t c1 = t[a + 1];
    if (nletters == 1) return c1 == 's' || c1 == 't' || c1 == 'm' || c1 == 'd';
    const uint8_t c2 = t[a + 2];
    return (c1 == 'r' && c2 == 'e') || (c1 == 'v' && c2 == 'e') || (c1 == 'l' && c2 == 'l');
}

__device__ __forceinline__ bool gpt2_piece_start(const uint8_t* t, const uint8_t* cls, int q, int slen, bool digits) {
    if (q == 0) return true;
    // first letter of a contraction: belongs to the apostrophe's piece
    if (t[q - 1] == 0x27 && (gpt2_fires(t, cls, q - 1, 1, slen) || gpt2_fires(t, cls, q - 1, 2, slen))) return false;
    const uint8_t c = cls[q] & kClsMask, pc = cls[q - 1] & kClsMask;
    bool start;
    if (c != pc) {
        const bool attaches = (c != kClsS) && !(digits && c == kClsN);
        start = !(t[q - 1] == 0x20 && attaches);
    } else {
        start = false;
        if (c == kClsS) {
            int nq = q + 1;
            while (nq < slen && nq < q + 4 && !(cls[nq] & kCharStart)) ++nq;  // \s chars are <= 3 bytes
            if (nq < slen && (cls[nq] & kClsMask) != kClsS) start = true;
        } else if (digits && c == kClsN) {
            start = true;
        }
    }
    if (!start && q >= 2 && gpt2_fires(t, cls, q - 2, 1, slen)) start = true;
    if (!start && q >= 3 && gpt2_fires(t, cls, q - 3, 2, slen)) start = true;
    return start;
}

// Scans string `str` (slen bytes) and hands complete pieces to the caller chunk by chunk.
//   on_chunk(np, w0, skew): pstart[0..np] (string positions, pstart[np] = end of the last piece)
//                           describe np complete pieces; text/cls cover them (text at skew).
//   on_long(b, e):          a piece of more than kChunk bytes, not staged in LDS.
// Wave-uniform; every lane must call it with the same arguments.
template <class OnChunk, class OnLong>
__device__ __forceinline__ void scan_string(WaveScratch& ws, const SplitDev& sp, const uint8_t* ascii_cls,
                                            const uint8_t* str, int slen, OnChunk&& on_chunk, OnLong&& on_long) {
    const bool digits = sp.kind == kSplitGpt2Digits;
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
        classify_window(ws, sp, ascii_cls, skew, w1 - w0);
        const uint8_t* t = text_bytes(ws) + skew - w0;  // index by string position
        const uint8_t* cls = ws.cls - w0;
        int np = 0;
        for (int base = c0; base < qlim; base += kWave) {
            const int q = base + lane_id();
            bool st = false;
            if (q < qlim) st = (q == c0) || ((cls[q] & kCharStart) && gpt2_piece_start(t, cls, q, slen, digits));
            const unsigned long long m = __ballot(st);
            if (st) ws.pstart[np + __popcll(m & lanemask_lt())] = uint16_t(q - c0);
            np += __popcll(m);
        }
        wave_sync();
        if (qlim == slen) {  // the string ends in this window: every piece is complete
            if (lane_id() == 0) ws.pstart[np] = uint16_t(slen - c0);
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
                classify_window(ws, sp, ascii_cls, lskew, lw1 - lw0);
                const uint8_t* lt = text_bytes(ws) + lskew - lw0;
                const uint8_t* lcls = ws.cls - lw0;
                for (int base = e; base < lq && !found; base += kWave) {
                    const int q = base + lane_id();
                    const bool st = q < lq && (lcls[q] & kCharStart) && gpt2_piece_start(lt, lcls, q, slen, digits);
                    const unsigned long long m = __ballot(st);
                    if (m) {
                        e = base + __ffsll(m) - 1;
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
