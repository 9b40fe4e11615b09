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
    const uint8_t* ctx_of_class;  // [n_classes]
    int32_t n_syms, n_states, sym_eot, sym_final_nl, n_ctx;
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

// "Previous character" context of start position p (patterns with ^, \b or look-behind only).
__device__ __forceinline__ int regex_context(const RegexDev& R, const RegexTables& T, const uint8_t* s, int slen, int p) {
    if (R.n_ctx <= 1 || p <= 0) return 0;
    int q = p - 1;
    while (q > 0 && p - q < 4 && (s[q] & 0xC0u) == 0x80u) --q;
    int len = 0;
    int sym = regex_symbol(R, T, s, slen, q, len);
    if (sym == R.sym_final_nl) sym = T.ascii_class['\n'];
    return R.ctx_of_class[sym];
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
            if (t & kRegexMatchBit) last = i;
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

}  // namespace ovtk
