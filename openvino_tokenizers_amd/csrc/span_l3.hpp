// span_l3.hpp -- the Llama-3 family's rule algebra in streamed form (round 5): the piece starts of a 2 048-byte block of
// lookup_span_kernel, 32 bytes per lane, whatever the rows are and whatever the script is.
//
// The pattern (src/regex_split.cpp:286-301 hands it to PCRE2, src/utils.cpp:256-272; Llama-3's, with Qwen2's and tiktoken-cl100k's as
// the two parameters of SplitDev):
//   (?i:'s|'t|'re|'ve|'m|'ll|'d) | [^\r\n\p{L}\p{N}]?\p{L}+ | \p{N}{1,3} | ?[^\s\p{L}\p{N}]+[\r\n]* | \s*[\r\n]+ | \s+(?!\S) | \s+
// The rules are llama3_start_mask's (split_device.hpp; derived from the literal matcher llama3_match_end, which is what they are
// tested against), but neither in its form (a 64-byte word per lane, gathered by ballots: nine busy lanes) nor in
// llama3_packed_starts' (one flag per byte in bit 7 of a dword: every rule costs an instruction per DWORD, every look-around two,
// and what is not local is a bounded look-around with a fallback).  Here:
//   * the lane's 32 bytes are TRANSPOSED into eight bit planes (plane k, bit i = bit k of byte i: sixteen v_dot4_u32_u8 and a few
//     shifts per plane), the ASCII classes are boolean functions of the planes -- a handful of instructions per class and LANE, not per
//     dword --, and every rule is an instruction on 32-bit masks; "the byte k places before / behind" is one v_alignbit with the
//     neighbouring lane's copy of the mask (one DPP move);
//   * non-ASCII characters are classified per lane: a lane walks its own lead bytes (the code point from the block's LDS text, its
//     class from a flat two-bit table of the BMP -- one load per character, those of a batch in flight together) and writes the class
//     over all bytes of the character, so that runs are runs of bytes;
//   * what is not local -- digit groups of three counted from their run's start, "a line break follows in this white-space run", "these
//     line breaks follow an O character", "this white-space run reaches the string's end" -- is EXACT: a flood through a run is one
//     add (the carry ripples through the run's ones), and the carry from lane to lane is the same add once more on the 64-bit masks
//     "my flood leaves through bit 31" / "I am all run" of the wave, on the scalar unit.  No bound, no second form: five line breaks, a
//     line break with an indented line behind it, forty digits are text like any other;
//   * rows: a row's first byte (`rs`) starts a piece, and nothing looks across it -- "the byte before" is none at a row start, "the
//     byte behind" none at a row's last byte.
// What stays outside: a non-ASCII \p{N} (digit groups count characters, the floods bytes) and U+017F (folds to `s` under (?i)) send
// the BLOCK to the literal matcher on lane 0 (`odd`), as they send a window there in the other forms.
#pragma once

#include "split_device.hpp"

namespace ovtk {

// sum of the four bytes of `a` times the four bytes of `b`, plus c -- v_dot4_u32_u8
__device__ __forceinline__ uint32_t dot4_u8(uint32_t a, uint32_t b, uint32_t c) {
#ifdef OVTK_SIMT_EMULATOR
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
    return c;
#else
    return __builtin_amdgcn_udot4(a, b, c, false);
#endif
}
__device__ __forceinline__ uint32_t bit_reverse(uint32_t v) {
#ifdef OVTK_SIMT_EMULATOR
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
#else
    return __builtin_bitreverse32(v);
#endif
}

// ---- masks over the wave's 2 048 bytes: lane l holds bits [32 l, 32 l + 32)
// bit i = bit (i - K) of the wave-wide mask ("the byte K places before"); vp = the previous lane's word
template <int K>
__device__ __forceinline__ uint32_t bm_before(uint32_t v, uint32_t vp) { return funnel_shr(vp, v, 32 - K); }
// bit i = bit (i + K) ("the byte K places behind"); vn = the next lane's word
template <int K>
__device__ __forceinline__ uint32_t bm_after(uint32_t v, uint32_t vn) { return funnel_shr(v, vn, K); }

// The eight bit planes of the lane's 32 bytes: pl[k] bit i = bit k of byte i.
__device__ __forceinline__ void span_bit_planes(const uint32_t (&x)[8], uint32_t (&pl)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t M = 0x01010101u << k;
        uint32_t a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // eight bytes -> eight bits at [k, k + 8)
            a[j] = dot4_u8(x[2 * j] & M, 0x08040201u, 0u);
            a[j] = dot4_u8(x[2 * j + 1] & M, 0x80402010u, a[j]);
        }
        const uint32_t lo = (a[0] | (a[1] << 8)) >> k, hi = (a[2] | (a[3] << 8)) >> k;
        pl[k] = lo | (hi << 16);
    }
}

// F(p) = seed(p) | (link(p) & F(p - 1)) over the wave's bytes in ascending order: what `seed` reaches going up through positions that
// are linked to the one below them.  One add per lane -- generate = seed, propagate = link: the carry INTO a position is F of the
// position below --, and the carry into a lane from the lanes below by the same add on the wave's masks (a lane generates when its
// own add carries out, propagates when all of it propagates).  `into`: F(p - 1), for callers that want the flood one position on.
__device__ __forceinline__ uint32_t flood_up(uint32_t link, uint32_t seed, uint32_t* into = nullptr) {
    const uint32_t a = link | seed;
    const uint32_t sum = a + seed;
    const bool cout = sum < a;                      // (seed <= a: the add wrapped iff it carried out)
    const uint32_t prop = link & ~seed;
    const uint32_t head = prop ^ (prop + 1u);       // where a carry that enters at bit 0 arrives: the low run of `prop`, and one more
    const unsigned long long G = __ballot(cout), P = __ballot(prop == ~0u);
    const unsigned long long A = P | G;
    const unsigned long long cin = (A + G) ^ A ^ G;   // carry into lane l
    uint32_t E = sum ^ a ^ seed;                    // carry into every bit
    if ((cin >> lane_id()) & 1ull) E |= head;
    if (into) *into = E;
    return seed | (link & E);
}
// F(p) = seed(p) | (link(p) & F(p + 1)): the same going down (link: "linked to the position above"); `into`: F(p + 1).
__device__ __forceinline__ uint32_t flood_down(uint32_t link, uint32_t seed, uint32_t* into = nullptr) {
    const uint32_t rl = bit_reverse(link), rsd = bit_reverse(seed);
    const uint32_t a = rl | rsd;
    const uint32_t sum = a + rsd;
    const bool cout = sum < a;
    const uint32_t prop = rl & ~rsd;
    const uint32_t head = prop ^ (prop + 1u);
    const unsigned long long G = __brevll(__ballot(cout)), P = __brevll(__ballot(prop == ~0u));
    const unsigned long long A = P | G;
    const unsigned long long cin = (A + G) ^ A ^ G;   // bit 63 - l: carry into lane l from the lanes above
    uint32_t E = sum ^ a ^ rsd;
    if ((cin >> (kWave - 1 - lane_id())) & 1ull) E |= head;
    E = bit_reverse(E);
    if (into) *into = E;
    return seed | (link & E);
}

// SplitDev::uc_flat holds the two class bits (kClsO / kClsL / kClsN / kClsS) of the code points below kUcFlatLimit, four to a byte: planes
// 0 and 1 (every script in use, the emoji, the mathematical alphabets); the rest goes through the two-level table.
constexpr uint32_t kUcFlatLimit = 0x20000u;
constexpr int kSpanClassScratch = 4 * kWave * 4 + 1024 * 2;   // four 64-word bit masks and up to 1 024 lead bytes (a block of two-byte characters)
struct __attribute__((packed, aligned(1))) L3Bytes4 { uint32_t v; };

// The class masks of the block's bytes: bit i of a lane's word = byte 32 l + i is of the class; a character's class stands on ALL its
// bytes, so that runs are runs of bytes.
//   x      the lane's 32 bytes (bytes behind the block: anything)          vm    bit i: byte 32 l + i belongs to the block
//   rs     row starts: bit i = byte 32 l + i is the first byte of a row    text  the block's bytes in LDS (the same bytes)
//   at_end the block ends where its text ends (else: it was cut at 2 048 bytes and more follows)
//   scratch kSpanClassScratch bytes of LDS (the characters' list and class masks of a block with non-ASCII text)
struct SpanClassMasks {
    uint32_t L, N, W, SP, NL, AP;   // \p{L}, \p{N}, \s, U+0020, \r \n, the apostrophe
    uint32_t INS;                   // bytes of a character behind its first
    uint32_t V;                     // the block's bytes
    uint32_t p[7];                  // bit planes 0..6 of the bytes (plane k, bit i = bit k of byte i)
    uint32_t asc;                   // V and below 0x80
    bool any_hi;                    // (wave-uniform) the block holds a byte >= 0x80
};
// STRICT_N (the Llama-3 family): false (wave-uniform) when the block holds a non-ASCII \p{N} or U+017F -- what that family's algebra
// does not cover; else a non-ASCII \p{N} is of class N like any digit (the GPT-2 family: ` ?\p{N}+`, or every N character a piece).
template <bool STRICT_N>
__device__ __forceinline__ bool span_class_masks(const uint32_t (&x)[8], uint32_t rs, uint32_t vm, const uint8_t* text, uint32_t* scratch,
                                                 const SplitDev& sp, bool at_end, int b_len, SpanClassMasks& cm) {
    const int l = lane_id();
    uint32_t pl[8];
    span_bit_planes(x, pl);
    const uint32_t p0 = pl[0], p1 = pl[1], p2 = pl[2], p3 = pl[3], p4 = pl[4], p5 = pl[5], p6 = pl[6], p7 = pl[7];
    const uint32_t V = vm;
    // ---- ASCII classes (PCRE2_UCP below 0x80: \p{L} = [A-Za-z], \p{N} = [0-9], \s = [\t-\r ])
    const uint32_t asc = V & ~p7;
    const uint32_t hn0 = asc & ~(p6 | p5 | p4);                                   // 0x00 .. 0x0F
    const uint32_t low5_gt26 = p4 & p3 & (p2 | (p1 & p0));
    uint32_t L = asc & p6 & (p0 | p1 | p2 | p3 | p4) & ~low5_gt26;                // 0x41..0x5A, 0x61..0x7A
    uint32_t N = asc & ~p6 & p5 & p4 & ~(p3 & (p2 | p1));                          // 0x30..0x39
    const uint32_t SP = asc & ~p6 & p5 & ~(p4 | p3 | p2 | p1 | p0);               // 0x20
    const uint32_t NL = hn0 & p3 & (p1 ^ p0) & (p2 ^ p1);                          // 0x0A, 0x0D
    uint32_t W = SP | (hn0 & p3 & (p0 | p1 | p2) & ~(p2 & p1));                    // 0x09..0x0D, 0x20
    const uint32_t AP = asc & ~p6 & p5 & ~(p4 | p3) & p2 & p1 & p0;               // 0x27
    // ---- non-ASCII characters
    const uint32_t HI = V & p7;
    uint32_t INS = 0;
    const bool any_hi = __ballot(HI != 0) != 0;
    if (any_hi) {
        // A lane's own loop over its lead bytes -- code point, table, class, one after the other -- runs as long as the lane with the
        // most characters (sixteen in 32 bytes of Cyrillic) and waits for memory every time round: 1.84 ms for config 4's batch where
        // ASCII text took 0.16 (profiles/r05).  So the WAVE classifies the block's characters, one per lane and round: the lanes list
        // their lead bytes in LDS (a position and the bytes left in the row: a few instructions per character, no memory), then
        // character j is lane j's -- four text bytes from LDS, the code point, ONE table load, and the class OR-ed over the
        // character's bytes in three LDS bit masks (letter, white space, "not a first byte") that the lanes read back as their words.
        const uint32_t lead = HI & p6;
        const uint32_t brk = rs | ~V;   // a row begins here, or the block is over
        const unsigned long long brk64 = (unsigned long long)brk | ((unsigned long long)(lane_next(rs) | ~lane_next(V)) << 32);
        uint32_t* cw = scratch;                                           // [4][64]: L, W, INS, N
        uint16_t* list = reinterpret_cast<uint16_t*>(scratch + 4 * kWave);   // [<= 1024]
        cw[l] = 0;
        cw[kWave + l] = 0;
        cw[2 * kWave + l] = 0;
        if (!STRICT_N) cw[3 * kWave + l] = 0;
        const int cnt = __popc(lead);
        const int incl = wave_incl_sum(cnt);
        const int n_lead = wave_readlane(incl, kWave - 1);
        {
            uint16_t* at = list + (incl - cnt);
            for (uint32_t f = lead; f; f &= f - 1u) {
                const int k = __ffs(f) - 1;
                // a character ends with its row (seq_char clips a lead byte's length to the string's end -- and masks the lead byte by
                // the clipped length: broken UTF-8 has no defined parity, but every form of the rules reads it the same way)
                const int room = __ffs(uint32_t(brk64 >> (k + 1)) | 8u);   // bytes up to the row's end, as far as four
                *at++ = uint16_t(uint32_t(32 * l + k) | (uint32_t(room - 1) << 11));
            }
        }
        wave_sync();
        bool odd = false;
        for (int jb = 0; jb < n_lead; jb += kWave) {
            const int j = jb + l;
            if (j < n_lead) {
                const uint32_t e = list[j];
                const int p = int(e & 0x7FFu), room = int(e >> 11) + 1;
                const uint32_t w4 = reinterpret_cast<const L3Bytes4*>(text + p)->v;
                const uint32_t b = w4 & 0xFFu;
                int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
                // (a character that the end of a CUT block cuts lies in the bytes the block does not decide; what its first bytes
                // happen to spell -- 0xE5 0xBF: U+017F -- must not send the block to the literal matcher: one block in sixteen of
                // config 4's text did, 1.8 ms per batch)
                const bool cut = !at_end && p + n > b_len;
                n = n < room ? n : room;
                const uint32_t notc = (w4 ^ 0x80808000u) & 0xC0C0C000u;   // bytes 1..3: zero where a continuation byte stands
                const int c = notc ? ((__ffs(notc) - 1) >> 3) - 1 : 3;
                const int have = 1 + (c < n - 1 ? c : n - 1);
                uint32_t cp = b & (0xFFu >> (n + 1));
                if (have >= 2) cp = (cp << 6) | ((w4 >> 8) & 0x3Fu);
                if (have >= 3) cp = (cp << 6) | ((w4 >> 16) & 0x3Fu);
                if (have >= 4) cp = (cp << 6) | ((w4 >> 24) & 0x3Fu);
                uint32_t cls;
                if (cp < kUcFlatLimit) cls = (uint32_t(sp.uc_flat[cp >> 2]) >> (2 * (cp & 3u))) & 3u;
                else cls = uc_nibble(sp, cp) & 3u;
                if (STRICT_N && (cls == kClsN || cp == 0x17Fu) && !cut) odd = true;
                const unsigned long long m = ((1ull << have) - 1ull) << (p & 31);
                const uint32_t m_lo = uint32_t(m), m_hi = uint32_t(m >> 32);
                const int word = p >> 5;
                if (cls == kClsL || cls == kClsS || (!STRICT_N && cls == kClsN)) {
                    uint32_t* dst = cw + (cls == kClsS ? kWave : (cls == kClsN ? 3 * kWave : 0)) + word;
                    atomicOr(dst, m_lo);
                    if (m_hi) atomicOr(dst + 1, m_hi);
                }
                const uint32_t i_lo = m_lo & ~(1u << (p & 31));
                if (i_lo) atomicOr(cw + 2 * kWave + word, i_lo);
                if (m_hi) atomicOr(cw + 2 * kWave + word + 1, m_hi);
            }
        }
        if (STRICT_N && __ballot(odd)) return false;
        wave_sync();
        L = (L | cw[l]) & V;
        W = (W | cw[kWave + l]) & V;
        INS = cw[2 * kWave + l] & V;
        if (!STRICT_N) N = (N | cw[3 * kWave + l]) & V;
    }
    cm.L = L; cm.N = N; cm.W = W; cm.SP = SP; cm.NL = NL; cm.AP = AP; cm.INS = INS; cm.V = V; cm.asc = asc; cm.any_hi = any_hi;
    cm.p[0] = p0; cm.p[1] = p1; cm.p[2] = p2; cm.p[3] = p3; cm.p[4] = p4; cm.p[5] = p5; cm.p[6] = p6;
    return true;
}

// Piece starts of the block's bytes (flags bit i = byte 32 l + i starts a piece) under the Llama-3 family's rules; the arguments are
// span_class_masks'.
//   und    (out) starts at positions >= und may depend on what follows the block: they are not to be used
// false (wave-uniform): the block holds a character the algebra does not cover (`odd` above); flags / und are not set.
__device__ __forceinline__ bool span_flags_l3(const uint32_t (&x)[8], uint32_t rs, uint32_t vm, const uint8_t* text, uint32_t* scratch,
                                              const SplitDev& sp, bool at_end, int b_len, uint32_t& flags, int& und) {
    const int l = lane_id();
    SpanClassMasks cm;
    if (!span_class_masks<true>(x, rs, vm, text, scratch, sp, at_end, b_len, cm)) return false;
    const uint32_t L = cm.L, N = cm.N, W = cm.W, SP = cm.SP, NL = cm.NL, AP = cm.AP, INS = cm.INS, V = cm.V, asc = cm.asc;
    const uint32_t p0 = cm.p[0], p1 = cm.p[1], p2 = cm.p[2], p3 = cm.p[3], p4 = cm.p[4], p6 = cm.p[6];
    const bool any_hi = cm.any_hi;
    const uint32_t O = V & ~(L | N | W);
    // ---- the neighbours' words; "the byte before / behind, in my row"
    const uint32_t rs_n = lane_next(rs), V_n = lane_next(V);
    const uint32_t re = bm_after<1>(rs, rs_n) | ~bm_after<1>(V, V_n);   // the last byte of its row (or of the block)
    const uint32_t L_p = lane_prev(L), N_p = lane_prev(N), W_p = lane_prev(W), O_p = lane_prev(O), SP_p = lane_prev(SP), NL_p = lane_prev(NL);
    const uint32_t O_n = lane_next(O), W_n = lane_next(W);
    const uint32_t pL = bm_before<1>(L, L_p) & ~rs, pN = bm_before<1>(N, N_p) & ~rs, pW = bm_before<1>(W, W_p) & ~rs;
    const uint32_t pO = bm_before<1>(O, O_p) & ~rs, pSP = bm_before<1>(SP, SP_p) & ~rs, pNL = bm_before<1>(NL, NL_p) & ~rs;
    const uint32_t aO = bm_after<1>(O, O_n) & ~re, aW = bm_after<1>(W, W_n) & ~re;
    const uint32_t sL = L & ~pL, sO = O & ~pO, sW = W & ~pW;
    // ---- contractions: an apostrophe that starts a piece, and s|t|m|d or re|ve|ll (any case) behind it in its row
    uint32_t fire1 = 0, fire2 = 0;
    if (__ballot(AP != 0)) {
        const uint32_t let = asc & p6;   // 0x40..0x7F: the case bit p5 is not looked at
        const uint32_t X1 = let & ((p4 & ~p3 & ~p2 & p1 & p0) | (p4 & ~p3 & p2 & ~p1 & ~p0) | (~p4 & p3 & p2 & ~p1 & p0) | (~p4 & ~p3 & p2 & ~p1 & ~p0));  // s t m d
        const uint32_t X2 = let & p4 & ~p3 & p1 & ~p0;                  // r (10010) v (10110)
        const uint32_t XE = let & ~p4 & ~p3 & p2 & ~p1 & p0;            // e (00101)
        const uint32_t XL = let & ~p4 & p3 & p2 & ~p1 & ~p0;            // l (01100)
        const uint32_t XE_n = lane_next(XE), XL_n = lane_next(XL);
        const uint32_t Y = (X2 & bm_after<1>(XE, XE_n) & ~re) | (XL & bm_after<1>(XL, XL_n) & ~re);   // re|ve|ll begins here
        const uint32_t X1_n = lane_next(X1), Y_n = lane_next(Y);
        const uint32_t c1 = AP & bm_after<1>(X1, X1_n) & ~re, c2 = AP & bm_after<1>(Y, Y_n) & ~re;
        fire1 = c1 & sO & ~pSP;
        fire2 = c2 & sO & ~pSP & ~fire1;
    }
    const uint32_t fire = fire1 | fire2;
    const uint32_t f1_p = lane_prev(fire1), f2_p = lane_prev(fire2);
    const uint32_t b_con = bm_before<2>(fire1, f1_p) | bm_before<3>(fire2, f2_p);       // the byte behind a contraction
    const uint32_t after_fire = bm_before<1>(fire, f1_p | f2_p);                          // its first letter stays with it
    // ---- an O run of ONE character (not behind U+0020, not a contraction) goes in front of the letters behind it; the last
    // character of a white-space run that something follows starts a piece (\s+(?!\S) backing off)
    uint32_t single_o, last_w, after_takes;
    const uint32_t nonw_follows = bm_after<1>(V & ~W, V_n & ~W_n) & ~re;   // the byte behind is of my row and not white space
    if (any_hi) {
        const uint32_t INS_n = lane_next(INS), INS_p = lane_prev(INS);
        const uint32_t i1 = bm_after<1>(INS, INS_n), i2 = bm_after<2>(INS, INS_n), i3 = bm_after<3>(INS, INS_n);
        // a predicate of a character's LAST byte, at its first byte (characters are at most four bytes)
        auto at_lead = [&](uint32_t q) -> uint32_t {
            const uint32_t q_n = lane_next(q);
            return (~i1 & q) | (i1 & ~i2 & bm_after<1>(q, q_n)) | (i1 & i2 & ~i3 & bm_after<2>(q, q_n)) | (i1 & i2 & i3 & bm_after<3>(q, q_n));
        };
        single_o = sO & ~at_lead(aO);
        last_w = W & ~INS & at_lead(nonw_follows);
        const uint32_t takes = single_o & ~pSP & ~fire;
        // "the character in front of me is `takes`": its first byte is one to four bytes back
        const uint32_t t_p = lane_prev(takes);
        const uint32_t j1 = bm_before<1>(INS, INS_p), j2 = bm_before<2>(INS, INS_p), j3 = bm_before<3>(INS, INS_p);
        after_takes = (~j1 & bm_before<1>(takes, t_p)) | (j1 & ~j2 & bm_before<2>(takes, t_p)) | (j1 & j2 & ~j3 & bm_before<3>(takes, t_p)) |
                      (j1 & j2 & j3 & bm_before<4>(takes, t_p));
    } else {
        single_o = sO & ~aO;
        last_w = W & nonw_follows;
        const uint32_t takes = single_o & ~pSP & ~fire;
        after_takes = bm_before<1>(takes, lane_prev(takes));
    }
    const uint32_t supL = sL & ((pW & ~pNL) | (after_takes & ~rs) | after_fire);
    const uint32_t supO = sO & pSP;
    const uint32_t supW = sW & NL & pO;   // a white-space run that begins with line breaks right behind an O character: they are that piece's
    // ---- line breaks
    uint32_t b_abs = 0, b_ln = 0;
    const bool any_nl = __ballot(NL != 0) != 0;
    if (any_nl) {
        // F: the line breaks an O piece takes ([\r\n]* behind the O run) -- up from its first through consecutive line breaks
        const uint32_t F = flood_up(NL & pNL, supW);
        b_abs = W & ~NL & bm_before<1>(F, lane_prev(F)) & ~rs;
        // the LAST line break of a white-space run: no line break is reached going up through the run
        const uint32_t up = W & aW;   // linked to the byte above: both white space, one row
        uint32_t above = 0;
        flood_down(up, NL, &above);
        const uint32_t last_nl = NL & ~(up & above);
        b_ln = W & bm_before<1>(last_nl, lane_prev(last_nl)) & ~rs;
        if (sp.l3_tail_ws) {   // `\s++$`: the run that ends the string is ONE piece -- no start behind its last line break
            // (a block that was cut ends in a byte that is no row's last; the rows that end inside it end where the next begins)
            const uint32_t row_end = bm_after<1>(rs, rs_n) | (at_end ? ~bm_after<1>(V, V_n) : 0u);
            const uint32_t T = flood_down(up, W & row_end);
            b_ln &= ~T;
        }
    }
    const uint32_t b_last = last_w & ~NL & pW & ~pNL;
    // ---- digits: every third one from the start of its run (`\p{N}{1,3}`), or every one (`\p{N}`)
    uint32_t G = N;
    if (!sp.l3_digits1 && __ballot(N != 0)) {
        // a run's group starts are the digits at its start's position mod 3: flood the runs from the starts of residue 0 and of
        // residue 1 (position 32 l + i: residue (2 l + i) mod 3)
        const int lm = l % 3;
        const uint32_t B0 = 0x49249249u, B1 = 0x92492492u, B2 = 0x24924924u;   // i mod 3 == 0 / 1 / 2
        // (2 l + i) mod 3 == r  <=>  i mod 3 == (r + l) mod 3
        const uint32_t M0 = lm == 0 ? B0 : (lm == 1 ? B1 : B2), M1 = lm == 0 ? B1 : (lm == 1 ? B2 : B0), M2 = lm == 0 ? B2 : (lm == 1 ? B0 : B1);
        const uint32_t sN = N & ~pN, linkN = N & pN;
        const uint32_t F0 = flood_up(linkN, sN & M0), F1 = flood_up(linkN, sN & M1);
        G = (F0 & M0) | (F1 & M1) | (N & ~(F0 | F1) & M2);
    }
    flags = ((sL & ~supL) | G | (sO & ~supO) | (sW & ~supW) | b_abs | b_ln | b_last | b_con) & V & ~INS;
    flags |= rs & V;
    if (l == 0) flags |= 1u;   // the block's first byte: a row's first, or where the block before stopped
    // ---- how far the block decides
    und = b_len;
    if (!at_end) {
        und = b_len > 8 ? b_len - 8 : 0;
        // a white-space run that touches the block's end: undecided from its second byte on.  The block's last three bytes are not
        // asked: a character the block's end cuts is not what its first bytes say (U+2003 cut behind 0xE2 reads as U+0002)
        const int k3 = b_len - 3 - 32 * l;
        // (a run ends with its row: the last byte of a row counts as one that is not white space)
        const uint32_t nonw = ((V & ~W) | bm_after<1>(rs, rs_n)) & (k3 >= 32 ? ~0u : (k3 <= 0 ? 0u : ((1u << k3) - 1u)));
        const unsigned long long nwl = __ballot(nonw != 0);
        int nonw_end = 0;
        if (nwl) {
            const int hl = 63 - __clzll(nwl);
            nonw_end = hl * 32 + (32 - __clz(uint32_t(wave_readlane(int(nonw), hl))));
        }
        if (nonw_end < b_len - 3 && nonw_end + 1 < und) und = nonw_end + 1;
    }
    return true;
}

// ---- the GPT-2 family on the same masks (a block with non-ASCII text: span_kernel.hpp's span_flags is the ASCII form)
//   's|'t|'re|'ve|'m|'ll|'d | ?\p{L}+ | ?\p{N}+ | ?[^\s\p{L}\p{N}]+ | \s+(?!\S) | \s+        (DIGITS: `\p{N}` for ` ?\p{N}+`)
// gpt2_start_mask's rules (split_device.hpp): every look is at most one character ahead and two letters, so a cut block decides all
// but its last kSpanHalo bytes; a non-ASCII \p{N} is a digit like any other.
template <bool DIGITS>
__device__ __forceinline__ void span_flags_gpt2m(const uint32_t (&x)[8], uint32_t rs, uint32_t vm, const uint8_t* text, uint32_t* scratch,
                                                 const SplitDev& sp, bool at_end, int b_len, uint32_t& flags) {
    SpanClassMasks cm;
    span_class_masks<false>(x, rs, vm, text, scratch, sp, at_end, b_len, cm);
    const uint32_t L = cm.L, N = cm.N, S = cm.W, SP = cm.SP, AP = cm.AP, INS = cm.INS, V = cm.V;
    const uint32_t O = V & ~(L | N | S);
    const uint32_t rs_n = lane_next(rs), V_n = lane_next(V);
    const uint32_t re = bm_after<1>(rs, rs_n) | ~bm_after<1>(V, V_n);   // the last byte of its row (or of the block)
    const uint32_t pL = bm_before<1>(L, lane_prev(L)) & ~rs, pN = bm_before<1>(N, lane_prev(N)) & ~rs, pS = bm_before<1>(S, lane_prev(S)) & ~rs;
    const uint32_t pO = bm_before<1>(O, lane_prev(O)) & ~rs, pSP = bm_before<1>(SP, lane_prev(SP)) & ~rs;
    const uint32_t same = (L & pL) | (N & pN) | (S & pS) | (O & pO);
    const uint32_t attaches = ~S & (DIGITS ? ~N : ~0u);
    uint32_t st = ~same & ~(pSP & attaches);
    // the last character of a white-space run that something follows (\s+(?!\S) backing off)
    uint32_t next_nonspace = bm_after<1>(V & ~S, V_n & ~lane_next(S)) & ~re;   // the byte behind is of my row and not white space ...
    if (cm.any_hi) {   // ... asked at the character's last byte
        const uint32_t INS_n = lane_next(INS), q_n = lane_next(next_nonspace);
        const uint32_t i1 = bm_after<1>(INS, INS_n), i2 = bm_after<2>(INS, INS_n), i3 = bm_after<3>(INS, INS_n);
        next_nonspace = (~i1 & next_nonspace) | (i1 & ~i2 & bm_after<1>(next_nonspace, q_n)) | (i1 & i2 & ~i3 & bm_after<2>(next_nonspace, q_n)) |
                        (i1 & i2 & i3 & bm_after<3>(next_nonspace, q_n));
    }
    st |= same & ((S & next_nonspace) | (DIGITS ? N : 0u));
    // contractions: an apostrophe that starts a piece (what stands in front of it is neither of class O nor U+0020) and s|t|m|d or
    // re|ve|ll behind it in its row; the letters stay with it, the byte behind them starts a piece
    if (__ballot(AP != 0)) {
        const uint32_t p0 = cm.p[0], p1 = cm.p[1], p2 = cm.p[2], p3 = cm.p[3], p4 = cm.p[4];
        const uint32_t low = cm.asc & cm.p[6] & cm.p[5];   // 0x60..0x7F
        const uint32_t X1 = low & ((p4 & ~p3 & ~p2 & p1 & p0) | (p4 & ~p3 & p2 & ~p1 & ~p0) | (~p4 & p3 & p2 & ~p1 & p0) | (~p4 & ~p3 & p2 & ~p1 & ~p0));  // s t m d
        const uint32_t X2 = low & p4 & ~p3 & p1 & ~p0;         // r (10010) v (10110)
        const uint32_t XE = low & ~p4 & ~p3 & p2 & ~p1 & p0;   // e (00101)
        const uint32_t XL = low & ~p4 & p3 & p2 & ~p1 & ~p0;   // l (01100)
        const uint32_t Y = (X2 & bm_after<1>(XE, lane_next(XE)) & ~re) | (XL & bm_after<1>(XL, lane_next(XL)) & ~re);   // re|ve|ll begins here
        const uint32_t blocked = pO | pSP;
        const uint32_t f1 = AP & bm_after<1>(X1, lane_next(X1)) & ~re & ~blocked, f2 = AP & bm_after<1>(Y, lane_next(Y)) & ~re & ~blocked;
        const uint32_t f1_p = lane_prev(f1), f2_p = lane_prev(f2);
        st |= bm_before<2>(f1, f1_p) | bm_before<3>(f2, f2_p);
        st &= ~bm_before<1>(f1 | f2, f1_p | f2_p);
    }
    flags = (st & V & ~INS) | (rs & V);
    if (lane_id() == 0) flags |= 1u;
}

// The same by the literal matcher, on lane 0 (wave-uniform call): the blocks span_flags_l3 does not cover -- and, in the emulator
// build, every other block as well, as the check of the algebra.  rs_words[i]: lane i's `rs`; fl_words[64]: LDS scratch.
__device__ __forceinline__ void span_flags_l3_literal(const uint32_t* rs_words, uint32_t* fl_words, const uint8_t* text, const SplitDev& sp,
                                                      bool at_end, int b_len, uint32_t& flags, int& und) {
    const int l = lane_id();
    fl_words[l] = 0;
    wave_sync();
    int und0 = b_len;
    if (l == 0) {
        for (int p = 0; p < b_len;) {
            int e = p + 1;   // the end of the row that holds p: the next row start, or the block's end
            while (e < b_len && !((rs_words[e >> 5] >> (e & 31)) & 1u)) ++e;
            const uint8_t* s = text + p;
            const int slen = e - p;   // (a row the block cuts ends here for the matcher: what that changes lies behind `und`)
            for (int q = 0; q < slen;) {
                fl_words[(p + q) >> 5] |= 1u << ((p + q) & 31);
                q = llama3_match_end(sp, s, slen, q);
            }
            if (e == b_len && !at_end) {
                int nonw_end = 0;   // behind the row's last character that is not white space (the last three bytes are not asked:
                                    // a character the block's end cuts is not what its first bytes say)
                for (int q = 0; q < slen - 3;) {
                    const SeqChar c = seq_char(sp, s, q, slen);
                    q += c.len;
                    if (c.cls != kClsS) nonw_end = q;
                }
                und0 = b_len > 8 ? b_len - 8 : 0;
                if (nonw_end < slen - 3 && p + nonw_end + 1 < und0) und0 = p + nonw_end + 1;
            }
            p = e;
        }
    }
    wave_sync();
    flags = fl_words[l];
    und = wave_readlane(und0, 0);
}

}  // namespace ovtk
