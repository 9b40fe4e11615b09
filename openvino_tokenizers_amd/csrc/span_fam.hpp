// span_fam.hpp -- two more pattern families on the streamed scan of lookup_span_kernel (round 5): DeepSeek-V3's main
// split pattern and tiktoken's o200k_base (GPT-4o), as rule algebra on the bit masks of span_l3.hpp.  Until now both ran the compiled
// DFA, one lane per row (regex_sparse_kernel, 51 / 54 GB/s where the Llama-3 family's scan does 315).
//
// src/regex_split.cpp:286-301 hands the pattern to PCRE2 (PCRE2_UTF | PCRE2_UCP, src/utils.cpp:256-272): leftmost-first alternation with
// backtracking.  Each family below is (1) a literal matcher -- the alternatives in order, one position at a time: what PCRE2 does,
// restated; pinned against the oracle's PCRE2 by tests/test_span_fam.py -- and (2) the same rules on 32-bit masks, a lane's 32 bytes
// at a time, pinned against (1) by tests/emu/l3_flags_fuzz.cpp and, in the emulator build, on every block the kernel scans.
//
// kFamDs3 (DeepSeek-V3, the third Split of its tokenizer.json):
//     [!"#$%&'()*+,\-./:;<=>?@\[\\\]^_`{|}~][A-Za-z]+ | [^\r\n\p{L}\p{P}\p{S}]?[\p{L}\p{M}]+ | ?[\p{P}\p{S}]+[\r\n]* | \s*[\r\n]+ | \s+(?!\S) | \s+
//   What no alternative takes (a digit, a control character with no letter behind it) is a GAP: RegexSplit hands the text between two
//   matches on as a piece of its own (regex_split.cpp:262-284), so a run of such characters is one piece.
// kFamO200k (o200k_base):
//     [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?
//   | [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?
//   | \p{N}{1,3} | ?[^\s\p{L}\p{N}]+[\r\n/]* | \s*[\r\n]+ | \s+(?!\S) | \s+
#pragma once

#include "span_l3.hpp"

namespace ovtk {

// ===================================================================================================================== masks
constexpr int kFamClassWords = 6;                                       // INS and up to five classes, 64 words each
constexpr int kSpanFamScratch = kFamClassWords * kWave * 4 + 1024 * 2;   // ... and the lead-byte list (span_class_masks' layout)

struct FamMasks {
    uint32_t U, Lw, B, M, N, W, Q;   // upper, lower, both (Lm Lo), marks, numbers, white space, P|S  (ds3: U = every letter, Lw = B = N = 0;
                                     // o200k: Q = 0 -- what it does not tell apart stays in "other")
    uint32_t SP, NL, AP, SL;         // U+0020, \r \n, the apostrophe, the slash
    uint32_t AL;                     // ASCII letters
    uint32_t INS, V, asc;
    uint32_t p[7];
    bool any_hi;
};

// span_class_masks for the eight classes.  o200k: false (wave-uniform) for a non-ASCII \p{N} or U+017F, as in the Llama-3 family.
template <int FAM>
__device__ __forceinline__ bool fam_class_masks(const uint32_t (&x)[8], uint32_t rs, uint32_t vm, const uint8_t* text, uint32_t* scratch,
                                                const SplitDev& sp, bool at_end, int b_len, FamMasks& cm) {
    constexpr bool DS3 = FAM == kFamDs3;
    const int l = lane_id();
    uint32_t pl[8];
    span_bit_planes(x, pl);
    const uint32_t p0 = pl[0], p1 = pl[1], p2 = pl[2], p3 = pl[3], p4 = pl[4], p5 = pl[5], p6 = pl[6], p7 = pl[7];
    const uint32_t V = vm;
    const uint32_t asc = V & ~p7;
    const uint32_t hn0 = asc & ~(p6 | p5 | p4);                                   // 0x00 .. 0x0F
    const uint32_t low5_gt26 = p4 & p3 & (p2 | (p1 & p0));
    const uint32_t AL = asc & p6 & (p0 | p1 | p2 | p3 | p4) & ~low5_gt26;         // 0x41..0x5A, 0x61..0x7A
    const uint32_t ND = asc & ~p6 & p5 & p4 & ~(p3 & (p2 | p1));                   // 0x30..0x39
    const uint32_t SP = asc & ~p6 & p5 & ~(p4 | p3 | p2 | p1 | p0);               // 0x20
    const uint32_t NL = hn0 & p3 & (p1 ^ p0) & (p2 ^ p1);                          // 0x0A, 0x0D
    uint32_t W = SP | (hn0 & p3 & (p0 | p1 | p2) & ~(p2 & p1));                    // 0x09..0x0D, 0x20
    const uint32_t AP = asc & ~p6 & p5 & ~(p4 | p3) & p2 & p1 & p0;               // 0x27
    const uint32_t SL = asc & ~p6 & p5 & ~p4 & p3 & p2 & p1 & p0;                 // 0x2F
    const uint32_t DEL = asc & p6 & p5 & p4 & p3 & p2 & p1 & p0;                  // 0x7F
    uint32_t Q = asc & (p6 | p5) & ~(SP | AL | ND | DEL);                          // printable, not alphanumeric: \p{P} | \p{S}
    uint32_t U = DS3 ? AL : (AL & ~p5), Lw = DS3 ? 0u : (AL & p5), B = 0, M = 0;
    // ---- non-ASCII characters: the wave classifies them, a character per lane and turn (span_class_masks has the account)
    const uint32_t HI = V & p7;
    uint32_t INS = 0;
    const bool any_hi = __ballot(HI != 0) != 0;
    if (any_hi) {
        const uint32_t lead = HI & p6;
        const uint32_t brk = rs | ~V;
        const unsigned long long brk64 = (unsigned long long)brk | ((unsigned long long)(lane_next(rs) | ~lane_next(V)) << 32);
        uint32_t* cw = scratch;                                                            // [kFamClassWords][64]: INS, then the classes
        uint16_t* list = reinterpret_cast<uint16_t*>(scratch + kFamClassWords * kWave);   // [<= 1024]
#pragma unroll
        for (int k = 0; k < kFamClassWords; ++k) cw[k * kWave + l] = 0;
        const int cnt = __popc(lead);
        const int incl = wave_incl_sum(cnt);
        const int n_lead = wave_readlane(incl, kWave - 1);
        {
            uint16_t* at = list + (incl - cnt);
            for (uint32_t f = lead; f; f &= f - 1u) {
                const int k = __ffs(f) - 1;
                const int room = __ffs(uint32_t(brk64 >> (k + 1)) | 8u);
                *at++ = uint16_t(uint32_t(32 * l + k) | (uint32_t(room - 1) << 11));
            }
        }
        wave_sync();
        bool odd = false;
        // class -> its mask's index (0: none).  ds3: letters 1, marks 2, P|S 3, white space 4.  o200k: upper 1, lower 2, both 3, marks 4, white space 5.
        constexpr uint32_t kSlot = DS3 ? 0x43021110u : 0x50043210u;   // (nibble k: class k's)
        for (int jb = 0; jb < n_lead; jb += kWave) {
            const int j = jb + l;
            if (j < n_lead) {
                const uint32_t e = list[j];
                const int p = int(e & 0x7FFu), room = int(e >> 11) + 1;
                const uint32_t w4 = reinterpret_cast<const L3Bytes4*>(text + p)->v;
                const uint32_t b = w4 & 0xFFu;
                int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
                const bool cut = !at_end && p + n > b_len;
                n = n < room ? n : room;
                const uint32_t notc = (w4 ^ 0x80808000u) & 0xC0C0C000u;
                const int c = notc ? ((__ffs(notc) - 1) >> 3) - 1 : 3;
                const int have = 1 + (c < n - 1 ? c : n - 1);
                uint32_t cp = b & (0xFFu >> (n + 1));
                if (have >= 2) cp = (cp << 6) | ((w4 >> 8) & 0x3Fu);
                if (have >= 3) cp = (cp << 6) | ((w4 >> 16) & 0x3Fu);
                if (have >= 4) cp = (cp << 6) | ((w4 >> 24) & 0x3Fu);
                const uint32_t cls = uc_c4(sp, cp);
                if (!DS3 && (cls == kC4Num || cp == 0x17Fu) && !cut) odd = true;
                const unsigned long long m = ((1ull << have) - 1ull) << (p & 31);
                const uint32_t m_lo = uint32_t(m), m_hi = uint32_t(m >> 32);
                const int word = p >> 5;
                const uint32_t slot = (kSlot >> (4 * cls)) & 15u;
                if (slot) {
                    uint32_t* dst = cw + slot * kWave + word;
                    atomicOr(dst, m_lo);
                    if (m_hi) atomicOr(dst + 1, m_hi);
                }
                const uint32_t i_lo = m_lo & ~(1u << (p & 31));
                if (i_lo) atomicOr(cw + word, i_lo);
                if (m_hi) atomicOr(cw + word + 1, m_hi);
            }
        }
        if (!DS3 && __ballot(odd)) return false;
        wave_sync();
        INS = cw[l] & V;
        if (DS3) {
            U = (U | cw[kWave + l]) & V;
            M = cw[2 * kWave + l] & V;
            Q = (Q | cw[3 * kWave + l]) & V;
            W = (W | cw[4 * kWave + l]) & V;
        } else {
            U = (U | cw[kWave + l]) & V;
            Lw = (Lw | cw[2 * kWave + l]) & V;
            B = cw[3 * kWave + l] & V;
            M = cw[4 * kWave + l] & V;
            W = (W | cw[5 * kWave + l]) & V;
        }
    }
    cm.U = U; cm.Lw = Lw; cm.B = B; cm.M = M; cm.N = ND; cm.W = W; cm.Q = DS3 ? Q : 0u;
    cm.SP = SP; cm.NL = NL; cm.AP = AP; cm.SL = SL; cm.AL = AL; cm.INS = INS; cm.V = V; cm.asc = asc; cm.any_hi = any_hi;
    cm.p[0] = p0; cm.p[1] = p1; cm.p[2] = p2; cm.p[3] = p3; cm.p[4] = p4; cm.p[5] = p5; cm.p[6] = p6;
    return true;
}

// What the two families' rules share: neighbours inside a row, a predicate of a character's last byte seen at its first, the white
// space alternatives, `und`.
struct FamCtx {
    uint32_t rs, rs_n, V, V_n, re, INS;
    bool any_hi;
    uint32_t i1, i2, i3, j1, j2, j3;
    __device__ __forceinline__ void init(uint32_t rs_, const FamMasks& cm) {
        rs = rs_;
        V = cm.V;
        INS = cm.INS;
        any_hi = cm.any_hi;
        rs_n = lane_next(rs);
        V_n = lane_next(V);
        re = bm_after<1>(rs, rs_n) | ~bm_after<1>(V, V_n);   // the last byte of its row (or of the block)
        i1 = i2 = i3 = j1 = j2 = j3 = 0;
        if (any_hi) {
            const uint32_t INS_n = lane_next(INS), INS_p = lane_prev(INS);
            i1 = bm_after<1>(INS, INS_n); i2 = bm_after<2>(INS, INS_n); i3 = bm_after<3>(INS, INS_n);
            j1 = bm_before<1>(INS, INS_p); j2 = bm_before<2>(INS, INS_p); j3 = bm_before<3>(INS, INS_p);
        }
    }
    // the byte before / behind, in my row
    __device__ __forceinline__ uint32_t before(uint32_t q) const { return bm_before<1>(q, lane_prev(q)) & ~rs; }
    __device__ __forceinline__ uint32_t after(uint32_t q) const { return bm_after<1>(q, lane_next(q)) & ~re; }
    // a predicate of a character's LAST byte, at its first byte
    __device__ __forceinline__ uint32_t at_lead(uint32_t q) const {
        if (!any_hi) return q;
        const uint32_t q_n = lane_next(q);
        return (~i1 & q) | (i1 & ~i2 & bm_after<1>(q, q_n)) | (i1 & i2 & ~i3 & bm_after<2>(q, q_n)) | (i1 & i2 & i3 & bm_after<3>(q, q_n));
    }
    // "the character in front of me is t" (t: first bytes), at a first byte
    __device__ __forceinline__ uint32_t after_char(uint32_t t) const {
        const uint32_t t_p = lane_prev(t);
        if (!any_hi) return bm_before<1>(t, t_p) & ~rs;
        return ((~j1 & bm_before<1>(t, t_p)) | (j1 & ~j2 & bm_before<2>(t, t_p)) | (j1 & j2 & ~j3 & bm_before<3>(t, t_p)) |
                (j1 & j2 & j3 & bm_before<4>(t, t_p))) & ~rs;
    }
};

// The white-space alternatives \s*[\r\n]+ | \s+(?!\S) | \s+ (span_flags_l3's): starts inside white-space runs.
//   tail     the line breaks (and, o200k, slashes) that the piece in front took: no start in them, a start behind them
//   E_tail   `tail` one position on
__device__ __forceinline__ uint32_t fam_space_starts(const FamCtx& cx, uint32_t W, uint32_t NL, uint32_t tail, uint32_t E_tail) {
    const uint32_t pW = cx.before(W), pNL = cx.before(NL);
    const uint32_t sW = W & ~pW;
    uint32_t st = sW | (E_tail & ~tail & ~cx.rs);
    if (__ballot(NL != 0)) {
        const uint32_t up = W & cx.after(W);   // linked to the byte above: both white space, one row
        uint32_t above = 0;
        flood_down(up, NL, &above);
        const uint32_t last_nl = NL & ~(up & above);
        st |= W & cx.before(last_nl);
    }
    const uint32_t nonw_follows = bm_after<1>(cx.V & ~W, cx.V_n & ~lane_next(W)) & ~cx.re;
    const uint32_t last_w = W & ~cx.INS & cx.at_lead(nonw_follows);
    st |= last_w & ~NL & pW & ~pNL;
    return st & ~tail;
}

// How far a cut block decides (span_flags_l3's rule; `hold`: a class whose run, when it touches the block's end, is undecided from its
// FIRST byte on -- o200k's upper-case run: whether a piece starts there depends on what follows the run).
__device__ __forceinline__ int fam_und(const FamCtx& cx, uint32_t W, uint32_t hold, bool at_end, int b_len) {
    if (at_end) return b_len;
    const int l = lane_id();
    int und = b_len > 8 ? b_len - 8 : 0;
    const int k3 = b_len - 3 - 32 * l;
    const uint32_t below = k3 >= 32 ? ~0u : (k3 <= 0 ? 0u : ((1u << k3) - 1u));
    const uint32_t row_last = bm_after<1>(cx.rs, cx.rs_n);
    auto end_of_last = [&](uint32_t m) -> int {   // behind the last set bit of the wave-wide mask (0: none)
        const unsigned long long lanes = __ballot(m != 0);
        if (!lanes) return 0;
        const int hl = 63 - __clzll(lanes);
        return hl * 32 + (32 - __clz(uint32_t(wave_readlane(int(m), hl))));
    };
    const int nonw_end = end_of_last(((cx.V & ~W) | row_last) & below);
    if (nonw_end < b_len - 3 && nonw_end + 1 < und) und = nonw_end + 1;
    if (__ballot(hold != 0)) {
        const int nonh_end = end_of_last(((cx.V & ~hold) | row_last) & below);
        if (nonh_end < b_len - 3 && nonh_end < und) und = nonh_end > 1 ? nonh_end : 1;   // (the block's first byte starts a piece whatever follows)
    }
    return und;
}

// ---- DeepSeek-V3's rules
__device__ __forceinline__ bool span_flags_ds3(const uint32_t (&x)[8], uint32_t rs, uint32_t vm, const uint8_t* text, uint32_t* scratch,
                                               const SplitDev& sp, bool at_end, int b_len, uint32_t& flags, int& und) {
    FamMasks cm;
    fam_class_masks<kFamDs3>(x, rs, vm, text, scratch, sp, at_end, b_len, cm);
    FamCtx cx;
    cx.init(rs, cm);
    const uint32_t V = cm.V, INS = cm.INS, W = cm.W, NL = cm.NL, Q = cm.Q, AL = cm.AL;
    const uint32_t WD = cm.U | cm.M;                 // [\p{L}\p{M}]
    const uint32_t X = V & ~(WD | Q | W);            // what only the optional character of the word alternative takes: \p{N}, \p{C}
    const uint32_t pWD = cx.before(WD), pQ = cx.before(Q), pX = cx.before(X), pSP = cx.before(cm.SP), pW = cx.before(W), pNL = cx.before(NL);
    const uint32_t sWD = WD & ~pWD, sQ = Q & ~pQ;
    // ---- an ASCII punctuation character that starts a piece, ASCII letters behind it: it takes them -- and only them
    uint32_t a1 = sQ & cm.asc & cx.after(AL) & ~pSP, b_a1 = 0, after_a1 = 0;
    if (__ballot(a1 != 0)) {
        after_a1 = cx.before(a1);
        uint32_t E = 0;
        const uint32_t run = flood_up(AL & cx.before(AL), AL & after_a1, &E);
        b_a1 = E & ~cx.rs & ~run & WD;   // a letter that is not ASCII, or a mark, right behind them: the next piece starts inside the run
    }
    // ---- words: a run of letters and marks, behind one optional character that is no line break, letter, punctuation or symbol
    const uint32_t supWD = sWD & ((pW & ~pNL) | pX | after_a1);
    // ---- X characters: in front of a word they start its piece; else they are gap -- a gap's first character starts a piece
    const uint32_t sX = X & ~INS & (cx.at_lead(cx.after(WD)) | ~pX);
    // ---- punctuation runs: behind U+0020 they are that blank's piece; the line breaks behind them are theirs
    const uint32_t supQ = sQ & pSP;
    uint32_t tail = 0, E_tail = 0;
    const uint32_t seed = NL & pQ;
    if (__ballot(seed != 0)) tail = flood_up(NL & pNL, seed, &E_tail);
    const uint32_t ws = fam_space_starts(cx, W, NL, tail, E_tail);
    flags = ((sWD & ~supWD) | b_a1 | sX | (sQ & ~supQ) | ws) & V & ~INS;
    flags |= rs & V;
    if (lane_id() == 0) flags |= 1u;
    und = fam_und(cx, W, 0u, at_end, b_len);
    return true;
}

// ---- o200k_base's rules
// Classes: U (Lu Lt), l (Ll), b (Lm Lo and the marks that belong to a word), N, white space, X (everything else but marks).  Marks are on
// both sides: \p{M} is a word character AND one of [^\s\p{L}\p{N}].  Which it is depends on what the match that reaches it started as:
//   * a STRETCH is a maximal run of X and mark characters.  Inside one, pieces start left to right: an X with a word character behind it
//     is that word's optional character (the marks behind it are word characters, to the end of the run of letters and marks); an X run of
//     two or more, one with no word character behind it, or one behind U+0020 ("killer") starts the ` ?[^\s\p{L}\p{N}]+` piece, and that
//     takes the REST of the stretch, marks included.  So: flood from the killers up through the stretch (`Pz`); marks outside it are word
//     characters.
//   * `[\r\n/]*` behind such a piece takes line breaks and -- behind a line break -- slashes (`tail`); a slash it takes is not an X of
//     the stretch that would begin there.  That is the one place where the rules feed back (tail <- Pz <- X <- tail): iterated, left to
//     right it settles in as many rounds as such pieces follow each other directly; text without "punctuation, line break, slash" takes
//     one.
//   * inside a run of word characters, `[Ub]*[lb]+` takes upper-case characters up to the first lower-case one, then everything up to the
//     next upper-case one: a piece starts at every U whose last non-b character before it is an l (`F1`: flood from the l through b).
//     When no l follows, the upper part gives back to its last b: the upper-case run that ENDS the word starts a piece when something of
//     the word stands in front of it (`Tr`) -- the second alternative takes that run.  A contraction behind a word is the word's
//     (`fire`), the character behind it starts anew.
__device__ __forceinline__ bool span_flags_o200k(const uint32_t (&x)[8], uint32_t rs, uint32_t vm, const uint8_t* text, uint32_t* scratch,
                                                 const SplitDev& sp, bool at_end, int b_len, uint32_t& flags, int& und) {
    const int l = lane_id();
    FamMasks cm;
    if (!fam_class_masks<kFamO200k>(x, rs, vm, text, scratch, sp, at_end, b_len, cm)) return false;
    FamCtx cx;
    cx.init(rs, cm);
    const uint32_t V = cm.V, INS = cm.INS, W = cm.W, NL = cm.NL, N = cm.N, U = cm.U, Lw = cm.Lw, M = cm.M, SP = cm.SP, AP = cm.AP, SL = cm.SL;
    const uint32_t p0 = cm.p[0], p1 = cm.p[1], p2 = cm.p[2], p3 = cm.p[3], p4 = cm.p[4], p6 = cm.p[6];
    const uint32_t L = U | Lw | cm.B;
    const uint32_t Xo = V & ~(L | M | N | W);
    const uint32_t pSP = cx.before(SP), pW = cx.before(W), pNL = cx.before(NL), pN = cx.before(N);
    const uint32_t LM = L | M, pLM = cx.before(LM);
    // ---- contractions behind a word: an apostrophe with a word character in front and s|t|m|d or re|ve|ll (any case) behind it
    uint32_t c1 = 0, c2 = 0;
    if (__ballot((AP & pLM) != 0)) {
        const uint32_t let = cm.asc & p6;
        const uint32_t X1 = let & ((p4 & ~p3 & ~p2 & p1 & p0) | (p4 & ~p3 & p2 & ~p1 & ~p0) | (~p4 & p3 & p2 & ~p1 & p0) | (~p4 & ~p3 & p2 & ~p1 & ~p0));  // s t m d
        const uint32_t X2 = let & p4 & ~p3 & p1 & ~p0;                  // r v
        const uint32_t XE = let & ~p4 & ~p3 & p2 & ~p1 & p0;            // e
        const uint32_t XL = let & ~p4 & p3 & p2 & ~p1 & ~p0;            // l
        const uint32_t Y = (X2 & cx.after(XE)) | (XL & cx.after(XL));
        c1 = AP & pLM & cx.after(X1);
        c2 = AP & pLM & cx.after(Y) & ~c1;
    }
    const uint32_t fire_cand = c1 | c2;
    // ---- the stretches: killers, the pieces they start, the tails of those
    const uint32_t next_word = cx.at_lead(cx.after(LM));   // (at a first byte) the character behind is a letter or a mark
    uint32_t Pz = 0, E_z = 0, tail = 0, E_tail = 0, X = Xo, sX = 0;
    const bool slashes = __ballot((SL & pNL) != 0) != 0;   // a slash behind a line break: the only kind a tail can take
    for (int round = 0;; ++round) {
        const uint32_t pX = cx.before(X);
        sX = X & ~pX;
        const uint32_t takes = sX & next_word & ~pSP & ~fire_cand;
        const uint32_t killer = sX & ~takes & ~fire_cand;
        const uint32_t St = X | M;
        Pz = 0; E_z = 0;
        if (__ballot(killer != 0)) Pz = flood_up(St & cx.before(St), killer, &E_z);
        E_z &= ~cx.rs;
        uint32_t t2 = 0, e2 = 0;
        const uint32_t seed = NL & E_z;
        if (__ballot(seed != 0)) t2 = flood_up((NL | SL) & ~cx.rs, seed, &e2);
        const bool same = __ballot(t2 != tail) == 0;
        tail = t2;
        E_tail = e2;
        if (same || !slashes) break;
        if (round == 3) return false;   // pieces of this kind chained four deep: the literal matcher's
        X = Xo & ~tail;
    }
    const uint32_t Mw = M & ~Pz;           // the marks that are word characters
    const uint32_t WD = L | Mw, bb = cm.B | Mw;
    const uint32_t pWD = cx.before(WD);
    const uint32_t sWD = WD & ~pWD;
    // (a mark in front that the stretch's piece took: no word there.)  A contraction right behind a contraction -- `it's's` -- is not one: the
    // letter in front of its apostrophe went with the first; the apostrophe starts a piece, its letter is a word, and a THIRD one
    // is that word's.  fire(p) = candidate(p) and not fire(the candidate whose letters end in front of p): settled from the left, a round
    // per link of such a chain (text has none)
    uint32_t fire1 = c1 & ~E_z, fire2 = c2 & ~E_z;
    if (__ballot(fire_cand != 0)) {
        const uint32_t ok1 = fire1, ok2 = fire2;
        const uint32_t chained = bm_before<2>(ok1, lane_prev(ok1)) | bm_before<3>(ok2, lane_prev(ok2));
        if (__ballot(((ok1 | ok2) & chained) != 0)) {
            fire1 = ok1 & ~chained;
            fire2 = ok2 & ~chained;
            for (int round = 0;; ++round) {
                const uint32_t blocked = bm_before<2>(fire1, lane_prev(fire1)) | bm_before<3>(fire2, lane_prev(fire2));
                const uint32_t n1 = ok1 & ~blocked, n2 = ok2 & ~blocked;
                const bool same = __ballot(((n1 ^ fire1) | (n2 ^ fire2)) != 0) == 0;
                fire1 = n1;
                fire2 = n2;
                if (same) break;
                if (round == 6) return false;
            }
        }
    }
    const uint32_t fire = fire1 | fire2;
    const uint32_t f1_p = lane_prev(fire1), f2_p = lane_prev(fire2);
    const uint32_t b_con = (bm_before<2>(fire1, f1_p) | bm_before<3>(fire2, f2_p));
    const uint32_t after_fire = bm_before<1>(fire, f1_p | f2_p);
    const uint32_t con_letters = after_fire | bm_before<2>(fire2, f2_p);
    // ---- X characters outside the killers' pieces start a piece (a killer, or a word's optional character), but behind U+0020 (that blank's)
    const uint32_t free_x = sX & ~E_z & ~fire;
    const uint32_t takes = free_x & next_word & ~pSP;
    const uint32_t supWD = sWD & ((pW & ~pNL) | cx.after_char(takes) | after_fire);
    // ---- inside the words
    uint32_t b_up = 0;
    if (__ballot(U != 0)) {
        uint32_t E1 = 0;
        flood_up(bb & ~cx.rs, Lw & ~con_letters, &E1);
        b_up = U & E1 & ~cx.rs;
        const uint32_t Uw = U & ~con_letters, WDw = WD & ~con_letters;   // (a contraction's letters are no part of the word behind them)
        const uint32_t ends = Uw & ~cx.after(WDw);
        const uint32_t Tr = flood_down(Uw & cx.after(Uw), ends);
        b_up |= Tr & ~cx.before(Tr) & cx.before(WDw);
    }
    // ---- digits: every third one from the start of its run
    uint32_t G = N;
    if (__ballot(N != 0)) {
        const int lm = l % 3;
        const uint32_t B0 = 0x49249249u, B1 = 0x92492492u, B2 = 0x24924924u;
        const uint32_t M0 = lm == 0 ? B0 : (lm == 1 ? B1 : B2), M1 = lm == 0 ? B1 : (lm == 1 ? B2 : B0), M2 = lm == 0 ? B2 : (lm == 1 ? B0 : B1);
        const uint32_t sN = N & ~pN, linkN = N & pN;
        const uint32_t F0 = flood_up(linkN, sN & M0), F1 = flood_up(linkN, sN & M1);
        G = (F0 & M0) | (F1 & M1) | (N & ~(F0 | F1) & M2);
    }
    const uint32_t ws = fam_space_starts(cx, W, NL, tail, E_tail);
    flags = ((sWD & ~supWD) | b_up | b_con | (free_x & ~pSP) | G | ws) & ~tail & V & ~INS;
    flags |= rs & V;
    if (l == 0) flags |= 1u;
    und = fam_und(cx, W, U, at_end, b_len);
    return true;
}

// The same by the literal matcher, on lane 0 (span_flags_l3_literal's contract).
template <int FAM>
__device__ __forceinline__ void span_flags_fam_literal(const uint32_t* rs_words, uint32_t* fl_words, const uint8_t* text, const SplitDev& sp,
                                                       bool at_end, int b_len, uint32_t& flags, int& und) {
    const int l = lane_id();
    fl_words[l] = 0;
    wave_sync();
    int und0 = b_len;
    if (l == 0) {
        for (int p = 0; p < b_len;) {
            int e = p + 1;
            while (e < b_len && !((rs_words[e >> 5] >> (e & 31)) & 1u)) ++e;
            const uint8_t* s = text + p;
            const int slen = e - p;
            for (int q = 0; q < slen;) {
                fl_words[(p + q) >> 5] |= 1u << ((p + q) & 31);
                q = fam_match_end<FAM>(sp, s, slen, q);
            }
            if (e == b_len && !at_end) {
                int nonw_end = 0, nonu_end = 0;
                for (int q = 0; q < slen - 3;) {
                    const SeqChar c = fam_char(sp, s, q, slen);
                    q += c.len;
                    if (c.cls != int(kC4Space)) nonw_end = q;
                    if (c.cls != int(kC4Upper)) nonu_end = q;
                }
                und0 = b_len > 8 ? b_len - 8 : 0;
                if (nonw_end < slen - 3 && p + nonw_end + 1 < und0) und0 = p + nonw_end + 1;
                if (FAM == kFamO200k && nonu_end < slen - 3 && p + nonu_end < und0) und0 = p + nonu_end > 1 ? p + nonu_end : 1;
            }
            p = e;
        }
    }
    wave_sync();
    flags = fl_words[l];
    und = wave_readlane(und0, 0);
}

}  // namespace ovtk
