// regex_device.hpp -- RegexSplit for patterns without a hand-written scanner: the DFA compiled by regex_compile.cpp,
// run ONE LANE PER ROW (src/regex_split.cpp:222-314 is a sequential loop over one string's matches; rows are
// independent).  A lane walks its strings character by character: class of the character (128-byte ASCII table,
// two-level table otherwise), one transition, "a match ends here" bit -- the transition table sits in LDS when it fits
// (address-divergent reads: LDS has no cache-line penalty), else it is read through L2.
// The loop around the matcher (gaps / matches, the five behaviours, invert, max_splits, skips) follows
// regex_split.cpp:240-309 statement by statement.
#pragma once

#include "device_common.hpp"
#include "encode_kernels.hpp"
#include "regex_compile.hpp"

namespace ovtk {

struct RegexDev {
    const uint16_t* trans;        // [n_states * n_syms]
    const uint8_t* ascii_class;   // [128]
    const uint16_t* cp_index;     // [0x110000 >> 7]
    const uint8_t* cp_blocks;     // [n_blocks * 128]
    const uint8_t* ctx_next;      // [n_ctx * n_classes]: the context automaton (RegexProgram::ctx_next)
    int32_t n_syms, n_states, sym_eot, sym_final_nl, n_ctx, behind_chars;
    int32_t cp_blocks_bytes;      // size of cp_blocks
    uint16_t start[kRegexMaxCtx];
    int32_t mode;                 // 0 removed, 1 isolated, 2 merged-with-previous, 3 merged-with-next
    int32_t invert;
    int32_t max_splits;
};

constexpr int kRegexLdsEntries = 12288;  // transitions kept in LDS (24 KB per block); larger tables stay in global memory

struct RegexTables {  // what a lane reads per character
    const uint16_t* trans;       // LDS or global
    const uint8_t* ascii_class;  // LDS
};

// Symbol at string position i (i < slen) and the character's length in bytes.
__device__ __forceinline__ int regex_symbol(const RegexDev& R, const RegexTables& T, const uint8_t* s, int slen, int i, int& len) {
    const uint32_t b = s[i];
    if (b < 0x80u) {
        len = 1;
        if (b == '\n' && i == slen - 1 && R.sym_final_nl >= 0) return R.sym_final_nl;
        return T.ascii_class[b];
    }
    uint32_t cp = b;
    len = 1;
    if (b >= 0xC0u) {  // (a stray continuation byte is taken as a character of its own: invalid UTF-8, parity undefined)
        int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
        if (i + n > slen) n = slen - i;
        cp = b & (0xFFu >> (n + 1));
        for (; len < n && (s[i + len] & 0xC0u) == 0x80u; ++len) cp = (cp << 6) | (s[i + len] & 0x3Fu);
    }
    if (cp > 0x10FFFFu) cp = 0x10FFFFu;
    return R.cp_blocks[uint32_t(R.cp_index[cp >> 7]) * 128u + (cp & 127u)];
}

// The character boundary `chars` characters in front of position i (not before `lo`).
__device__ __forceinline__ int regex_step_back(const uint8_t* s, int lo, int i, int chars) {
    for (; chars > 0 && i > lo; --chars) {
        const int e = i;
        --i;
        while (i > lo && e - i < 4 && (s[i] & 0xC0u) == 0x80u) --i;
    }
    return i;
}

// Context of start position p -- what the pattern's `^`, `\b` and look-behinds need to know of the text in front of it: the
// context automaton run over the last behind_chars characters (patterns with such assertions only).
__device__ __forceinline__ int regex_context(const RegexDev& R, const RegexTables& T, const uint8_t* s, int slen, int p) {
    if (R.n_ctx <= 1 || p <= 0) return 0;
    int q = regex_step_back(s, 0, p, R.behind_chars > 0 ? R.behind_chars : 1);
    int ctx = q == 0 ? 0 : 1;
    while (q < p) {
        int len = 0;
        int sym = regex_symbol(R, T, s, slen, q, len);
        if (sym == R.sym_final_nl) sym = T.ascii_class['\n'];
        ctx = R.ctx_next[ctx * R.sym_eot + sym];   // (sym_eot = the number of classes)
        q += len;
    }
    return ctx;
}

// Where the match that transition t reports at position i ended: i, or some characters back (a look-ahead decided late).
__device__ __forceinline__ int regex_match_end(uint32_t t, const uint8_t* s, int p, int i) {
    const int d = int((t >> kRegexDelayShift) & kRegexDelayMask);
    return d == 0 ? i : regex_step_back(s, p, i, d);
}

// The match PCRE2 finds at or after `start` (PCRE2Wrapper::match, src/utils.cpp:396-420): leftmost start position, at
// that position the first alternative / greediest repetition that lets the whole pattern match.  false: no match.
__device__ __forceinline__ bool regex_next_match(const RegexDev& R, const RegexTables& T, const uint8_t* s, int slen, int start,
                                                 int& mb, int& me) {
    int p = start;
    while (p <= slen) {
        int state = R.start[regex_context(R, T, s, slen, p)];
        int i = p, last = -1, first_len = 1;
        for (;;) {
            int len = 0;
            const int sym = i < slen ? regex_symbol(R, T, s, slen, i, len) : R.sym_eot;
            if (i == p) first_len = len;
            const uint32_t t = T.trans[state * R.n_syms + sym];
            if (t & kRegexMatchBit) last = regex_match_end(t, s, p, i);
            state = int(t & kRegexStateMask);
            if (state == 0 || i >= slen) break;
            i += len;
        }
        if (last >= 0) {
            mb = p;
            me = last;
            return true;
        }
        if (p >= slen) break;
        p += first_len;
    }
    return false;
}

// mode 0: row_cnt[row] = number of pieces.  mode 1: write begins / ends / skips at the row's output offset.
template <int WRITE>
static __global__ __launch_bounds__(kBlockThreads) void regex_split_kernel(RowsIn in, RegexDev R, EncodeWork w, int32_t* out_rb,
                                                                           int32_t* out_re, int32_t* out_begins,
                                                                           int32_t* out_ends, uint8_t* out_skips) {
    __shared__ uint16_t trans_lds[kRegexLdsEntries];
    __shared__ uint8_t ascii_lds[128];
    const int n_trans = R.n_states * R.n_syms;
    const bool in_lds = n_trans <= kRegexLdsEntries;
    if (in_lds)
        for (int i = int(threadIdx.x); i < n_trans; i += kBlockThreads) trans_lds[i] = R.trans[i];
    if (threadIdx.x < 128) ascii_lds[threadIdx.x] = R.ascii_class[threadIdx.x];
    __syncthreads();
    RegexTables T{in_lds ? trans_lds : R.trans, ascii_lds};
    if (w.status->flags & (kFlagRange | kFlagOutCapacity)) return;
    const int row = int(blockIdx.x) * kBlockThreads + int(threadIdx.x);
    const bool valid = row < in.n_rows;
    int o = 0;
    if (WRITE) {
        const int cnt = valid ? w.row_cnt[row] : 0;
        const int incl = wave_incl_sum(cnt);
        o = int(w.tile_off[row / kRowTile < (in.n_rows + kRowTile - 1) / kRowTile ? row / kRowTile : 0]) + incl - cnt;
        if (valid) {
            out_rb[row] = o;
            out_re[row] = o + cnt;
        }
    }
    if (!valid) return;
    int count = 0;
    auto put = [&](int b, int e, int skip) {
        if (WRITE) {
            out_begins[o + count] = b;
            out_ends[o + count] = e;
            if (out_skips) out_skips[o + count] = uint8_t(skip);
        }
        ++count;
    };
    for (int col = in.ragged_begins[row]; col < in.ragged_ends[row]; ++col) {
        const int sb = in.begins[col], se = in.ends[col];
        if (in.skips && in.skips[col]) {  // regex_split.cpp:231-234
            put(sb, se, 1);
            continue;
        }
        const uint8_t* s = in.chars + sb;
        const int len = se - sb;
        int start = 0;
        uint32_t num_splits = 0;
        long long last_begin = -1;  // size_t(-1) in the reference; travels through `int begin` as -1
        auto add_split = [&](int b, int e, bool flag) {  // regex_split.cpp:244-284
            switch (R.mode) {
                case 0:
                    if (flag) return;
                    break;
                case 1: break;
                case 2:
                    if (!flag && e != len) {
                        last_begin = b;
                        return;
                    } else if (flag) {
                        b = int(last_begin);
                    }
                    break;
                default:
                    if (!flag) {
                        if (last_begin != -1) b = int(last_begin);
                    } else {
                        last_begin = b;
                        return;
                    }
                    break;
            }
            b = b > 0 ? b : 0;
            e = e < len ? e : len;
            if (num_splits == uint32_t(R.max_splits)) e = len;  // uint32 vs int compare (:278): -1 never matches
            put(sb + b, sb + e, 0);
            ++num_splits;
        };
        int mb = 0, me = 0;
        // an empty match ends the loop like no match (regex_split.cpp:154-161)
        while (regex_next_match(R, T, s, len, start, mb, me) && me != mb) {  // :286-301
            if (mb != start) add_split(start, mb, R.invert != 0);
            add_split(mb, me, R.invert == 0);
            start = me;
        }
        if (start < len) add_split(start, len, R.invert != 0);  // :302-304
        else if (R.mode == 3 && last_begin != (long long)len) add_split(int(last_begin), len, R.invert != 0);  // :305-309
    }
    if (!WRITE) w.row_cnt[row] = count;
}

constexpr int kRegexSparseLdsMax = 64 * 1024;   // per block: two blocks of 256 lanes per CU
struct RegexSparseLds { int ascii_at, index_at, blocks_at, total; bool trans_in, cp_in; };
__host__ __device__ inline RegexSparseLds regex_sparse_layout(int n_trans, int cp_blocks_bytes) {
    RegexSparseLds y{};
    int at = 0;
    y.trans_in = n_trans * 2 <= 24 * 1024;
    if (y.trans_in) at = (n_trans * 2 + 3) & ~3;
    y.ascii_at = at;
    at += 128;
    const int cp = (0x110000 >> 7) * 2 + ((cp_blocks_bytes + 3) & ~3);
    y.cp_in = at + cp <= kRegexSparseLdsMax;
    y.index_at = at;
    y.blocks_at = at + (0x110000 >> 7) * 2;
    if (y.cp_in) at += cp;
    y.total = at;
    return y;
}

// ---- the same split in ONE pass, for the fused encode (round 4).  regex_split_kernel above is the op's contract: dense
// begins / ends behind a count pass, and its nested loops (matches inside strings inside rows, a matcher inside each) leave a
// wave's 64 lanes in 64 different places -- 1.2 ms per pass on 131 072 mixed-script rows.  Here:
//  * the pieces of a row go to a region of its own in buffers of the reference's capacity (n_chars + n_strings,
//    src/regex_split.cpp:182), taken by one atomic per wave; the row's ragged begin / end point there.  Nothing downstream
//    needs the pieces dense -- the BPE kernels only dereference offsets --, so there is no count pass, no second run of the
//    automaton, and no host wait between the split and the BPE stage;
//  * ONE loop per lane, one character per turn: symbol, transition, and -- when the attempt at this start position is over --
//    the bookkeeping of regex_next_match and of the loop around it (regex_split.cpp:286-309), as straight-line code under one
//    branch.  Lanes differ in data, not in where they are in the program;
//  * the text is read a dword at a time (one load per four characters).
// A row whose offsets leave their tensors gets a begin of -1: the BPE stage's own validation then reports OVTK_E_RANGE.
// PLAIN: behaviour `isolate` without max_splits (what every tokenizer.json Split step of a BPE model asks for): a piece is stored
// as it is found -- none of add_split's cases is compiled in.
template <bool PLAIN>
static __global__ __launch_bounds__(kBlockThreads) void regex_sparse_kernel(RowsIn in, RegexDev R, RunStatus* status, long long capacity,
                                                                            int32_t* out_rb, int32_t* out_re, int32_t* out_begins,
                                                                            int32_t* out_ends, uint8_t* out_skips) {
    // (out_skips: the RegexSplit op's third output -- a piece of a skipped string carries the flag, regex_split.cpp:231-234 -- or nullptr)
    int* bump = &status->n_out;
    // dynamic LDS (regex_sparse_lds_bytes): transitions | ASCII classes | code-point index | code-point blocks -- whatever of it
    // fits; with two waves per SIMD (a lane per row) every table read is on the critical path, and a non-ASCII character is two
    // dependent ones
#ifdef OVTK_SIMT_EMULATOR
    static uint32_t dyn_lds[kRegexSparseLdsMax / 4];
#else
    extern __shared__ uint32_t dyn_lds[];
#endif
    const int n_trans = R.n_states * R.n_syms;
    const RegexSparseLds lay = regex_sparse_layout(n_trans, R.cp_blocks_bytes);
    uint16_t* trans_lds = reinterpret_cast<uint16_t*>(dyn_lds);
    uint8_t* ascii_lds = reinterpret_cast<uint8_t*>(dyn_lds) + lay.ascii_at;
    uint16_t* index_lds = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(dyn_lds) + lay.index_at);
    uint8_t* blocks_lds = reinterpret_cast<uint8_t*>(dyn_lds) + lay.blocks_at;
    if (lay.trans_in)
        for (int i = int(threadIdx.x); i < n_trans; i += kBlockThreads) trans_lds[i] = R.trans[i];
    if (threadIdx.x < 128) ascii_lds[threadIdx.x] = R.ascii_class[threadIdx.x];
    if (lay.cp_in) {
        for (int i = int(threadIdx.x); i < (0x110000 >> 7); i += kBlockThreads) index_lds[i] = R.cp_index[i];
        for (int i = int(threadIdx.x); i < R.cp_blocks_bytes; i += kBlockThreads) blocks_lds[i] = R.cp_blocks[i];
    }
    __syncthreads();
    RegexTables T{lay.trans_in ? trans_lds : R.trans, ascii_lds};
    const uint16_t* cp_index = lay.cp_in ? index_lds : R.cp_index;
    const uint8_t* cp_blocks = lay.cp_in ? blocks_lds : R.cp_blocks;
    const int l = lane_id();
    const int row = int(blockIdx.x) * kBlockThreads + int(threadIdx.x);
    const bool valid = row < in.n_rows;
    // ---- the row's share of the output: a slot per character plus one per string
    int col = 0, col_end = 0;
    long long cap = 0;
    bool bad = false;
    if (valid) {
        col = in.ragged_begins[row];
        col_end = in.ragged_ends[row];
        if (col < col_end && (col < 0 || col_end > in.n_strings)) bad = true;
        for (int c = col; c < col_end && !bad; ++c) {
            const long long sb = in.begins[c], se = in.ends[c];
            if (sb < 0 || se < sb || se > in.n_chars) bad = true;
            else cap += se - sb + 1;
        }
        if (bad) cap = 0;
    }
    long long base = 0;
    {
        // (a 64-bit wave sum through two 32-bit ones: the low 31 bits and the rest)
        const int lo_incl = wave_incl_sum(int(cap & 0x3FFFFF)), hi_incl = wave_incl_sum(int(cap >> 22));
        const long long incl = (long long)lo_incl + ((long long)hi_incl << 22);
        const long long total = (long long)wave_readlane(lo_incl, kWave - 1) + ((long long)wave_readlane(hi_incl, kWave - 1) << 22);
        int b0 = 0;
        if (l == 0 && total > 0) b0 = atomicAdd(bump, int(total < INT32_MAX ? total : INT32_MAX));
        b0 = wave_readlane(b0, 0);
        base = (long long)b0 + incl - cap;
        if (b0 < 0 || (long long)b0 + total > capacity) {   // (overlapping strings can ask for more than n_chars + n_strings)
            // said as what it is (the rows' begin of -1 alone made the BPE stage report "offset outside tensor")
            if (l == 0) atomicOr(&status->flags, kFlagOutCapacity);
            bad = true;
        }
    }
    if (!valid) return;
    if (bad) {
        out_rb[row] = -1;
        out_re[row] = 0;
        return;
    }
    int count = 0;
    auto put = [&](int b, int e, int skip = 0) {
        out_begins[base + count] = b;
        out_ends[base + count] = e;
        if (out_skips) out_skips[base + count] = uint8_t(skip);
        ++count;
    };
    // ---- one loop: a character per turn
    bool need_string = true;
    int sb = 0, len = 0, start = 0, p = 0, i = 0, state = 0, last = -1, first_len = 1;
    uint32_t num_splits = 0;
    long long last_begin = -1;
    const uint8_t* s = in.chars;
    // the eight bytes of the chars tensor's dword grid around the current position: a character (up to four bytes) is cut out of
    // them with one funnel shift -- one or two loads per four characters, none per character
    uint32_t w0 = 0, w1 = 0;
    long long w_at = -8;   // tensor offset of w0 (a multiple of 4)
    auto load_dword = [&](long long at) -> uint32_t {
        if (at + 4 <= in.n_chars) return *reinterpret_cast<const uint32_t*>(in.chars + at);
        uint32_t v = 0;
        for (int k = 0; at + k < in.n_chars; ++k) v |= uint32_t(in.chars[at + k]) << (8 * k);
        return v;
    };
    auto add_split = [&](int b, int e, bool flag) {  // regex_split.cpp:244-284 (see regex_split_kernel)
        if (PLAIN) {
            put(sb + b, sb + e);
            return;
        }
        switch (R.mode) {
            case 0:
                if (flag) return;
                break;
            case 1: break;
            case 2:
                if (!flag && e != len) {
                    last_begin = b;
                    return;
                } else if (flag) {
                    b = int(last_begin);
                }
                break;
            default:
                if (!flag) {
                    if (last_begin != -1) b = int(last_begin);
                } else {
                    last_begin = b;
                    return;
                }
                break;
        }
        b = b > 0 ? b : 0;
        e = e < len ? e : len;
        if (num_splits == uint32_t(R.max_splits)) e = len;
        put(sb + b, sb + e);
        ++num_splits;
    };
    for (;;) {
        if (need_string) {
            if (col >= col_end) break;
            sb = in.begins[col];
            len = in.ends[col] - sb;
            if (in.skips && in.skips[col]) {  // regex_split.cpp:231-234 (BPETokenizer has no skips input: the string is one piece)
                put(sb, sb + len, 1);
                ++col;
                continue;
            }
            s = in.chars + sb;
            start = p = i = 0;
            num_splits = 0;
            last_begin = -1;
            state = R.start[0];   // (position 0 has no previous character)
            last = -1;
            need_string = false;
        }
        // the symbol at i
        int sym = R.sym_eot, clen = 0;
        if (i < len) {
            const long long g = (long long)sb + i;
            const long long ga = g & ~3ll;
            if (ga != w_at) {
                w0 = ga == w_at + 4 ? w1 : load_dword(ga);
                w1 = load_dword(ga + 4);
                w_at = ga;
            }
            const uint32_t c4 = funnel_shr(w0, w1, 8 * int(g & 3));   // the bytes at g, g + 1, g + 2, g + 3
            const uint32_t b = c4 & 0xFFu;
            if (b < 0x80u) {
                clen = 1;
                sym = (b == '\n' && i == len - 1 && R.sym_final_nl >= 0) ? R.sym_final_nl : int(T.ascii_class[b]);
            } else {   // regex_symbol(), on the four bytes in hand
                uint32_t cp = b;
                clen = 1;
                if (b >= 0xC0u) {
                    int n = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
                    if (i + n > len) n = len - i;
                    cp = b & (0xFFu >> (n + 1));
                    for (; clen < n && ((c4 >> (8 * clen)) & 0xC0u) == 0x80u; ++clen) cp = (cp << 6) | ((c4 >> (8 * clen)) & 0x3Fu);
                }
                if (cp > 0x10FFFFu) cp = 0x10FFFFu;
                sym = cp_blocks[uint32_t(cp_index[cp >> 7]) * 128u + (cp & 127u)];
            }
        }
        if (i == p) first_len = clen;
        const uint32_t t = T.trans[state * R.n_syms + sym];
        if (t & kRegexMatchBit) last = regex_match_end(t, s, p, i);
        state = int(t & kRegexStateMask);
        if (state == 0 || i >= len) {   // the attempt at p is over
            if (last >= 0 && last != p) {   // a match [p, last): the gap in front of it, then the match (regex_split.cpp:286-301)
                if (p != start) add_split(start, p, R.invert != 0);
                add_split(p, last, R.invert == 0);
                start = p = last;
            } else if (last == p || p >= len) {   // an empty match, or nothing from here to the end: the string is done (:302-309)
                if (start < len) add_split(start, len, R.invert != 0);
                else if (!PLAIN && R.mode == 3 && last_begin != (long long)len) add_split(int(last_begin), len, R.invert != 0);
                ++col;
                need_string = true;
                continue;
            } else {
                p += first_len;   // no match starts at p
            }
            i = p;
            state = R.start[regex_context(R, T, s, len, p)];
            last = -1;
        } else {
            i += clen;
        }
    }
    out_rb[row] = int(base);
    out_re[row] = int(base) + count;
}

}  // namespace ovtk
