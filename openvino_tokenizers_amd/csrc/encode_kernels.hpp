// encode_kernels.hpp -- the gfx950 kernels of RegexSplit, BPETokenizer and their fusion.
//
// Work decomposition: ONE WAVEFRONT PER RAGGED ROW (= per input string in converted pipelines).
//   encode_kernel<kFused>   row -> strings -> split scanner -> pieces -> BPE  (RegexSplit + BPETokenizer)
//   encode_kernel<kPieces>  row -> pre-split pieces -> BPE                    (BPETokenizer op contract)
// Pieces are processed 64 at a time, one lane per piece (path F), long pieces by the whole wave
// (path W), ambiguous / oversized pieces are deferred to exact_kernel (path X) -- see bpe_device.hpp.
//
// Output placement.  The reference keeps one running `ragged_offset` across rows
// (bpe_tokenizer.cpp:141-161); here every row first writes into a staging region whose offset
// comes from an exclusive scan of per-row capacities, a second scan of the per-row token counts
// gives the final offsets, and compact_kernel moves the ids.  A row is staged "compact" (ids
// back to back) unless one of its pieces was deferred; from that piece on it is "slotted": each
// piece owns the stretch [bytepos*mul, (bytepos+len)*mul) of the row's region (mul = 1 +
// end_suffix length), unused entries hold kEmptyId and compact_kernel squeezes them out.
#pragma once

#include "bpe_device.hpp"
#include "device_common.hpp"
#include "split_device.hpp"

namespace ovtk {

struct RowsIn {
    const int32_t* ragged_begins;
    const int32_t* ragged_ends;
    int32_t n_rows;
    const int32_t* begins;
    const int32_t* ends;
    int32_t n_strings;
    const uint8_t* chars;
    int64_t n_chars;
    const uint8_t* skips;  // bool per string, or nullptr
};

struct DeferredPiece { int32_t begin, len, stage_pos, row; };

struct EncodeWork {
    int32_t* row_stage;     // [n_rows + 1] staging offset of each row (exclusive scan of capacities)
    int32_t* row_cnt;       // [n_rows]     elements produced by each row
    int32_t* row_out;       // [n_rows + 1] final offset of each row
    uint8_t* row_slotted;   // [n_rows]
    int32_t* stage;
    int32_t stage_cap;
    DeferredPiece* deferred;
    int32_t deferred_cap;
    uint8_t* scratch;
    uint32_t scratch_cap;
    RunStatus* status;
};

constexpr int kScanThreads = 1024;
constexpr int kScanPerThread = 4;

// Exclusive scan of f(0..n) by ONE block of THREADS threads: put(i, prefix) for every i, returns the
// total to every thread.  64-bit accumulation (callers clamp / flag).
template <int THREADS, class F, class Put>
__device__ __forceinline__ long long block_exclusive_scan(int n, F&& f, Put&& put) {
    __shared__ long long wave_tot[THREADS / kWave];
    __shared__ long long carry_s;
    const int tid = int(threadIdx.x), l = lane_id(), wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int tile = 0; tile < n; tile += THREADS * kScanPerThread) {
        const int i0 = tile + tid * kScanPerThread;
        long long v[kScanPerThread], s = 0;
#pragma unroll
        for (int j = 0; j < kScanPerThread; ++j) {
            v[j] = (i0 + j < n) ? (long long)f(i0 + j) : 0;
            s += v[j];
        }
        long long incl = s;  // inclusive scan of s over the wave
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            long long t = __shfl_up(incl, d);
            if (l >= d) incl += t;
        }
        if (l == kWave - 1) wave_tot[wv] = incl;
        __syncthreads();
        long long before = carry_s;
        for (int k = 0; k < wv; ++k) before += wave_tot[k];
        long long run = before + incl - s;
#pragma unroll
        for (int j = 0; j < kScanPerThread; ++j) {
            if (i0 + j < n) put(i0 + j, run);
            run += v[j];
        }
        __syncthreads();
        if (tid == THREADS - 1) carry_s = run;  // last thread's running sum = total so far
        __syncthreads();
    }
    return carry_s;
}

// ---- K1: per-row staging capacity -> offsets.  cap(row) = mul * sum over its strings of max(len, 1).
static __global__ __launch_bounds__(kScanThreads) void prepare_rows_kernel(RowsIn in, int mul, EncodeWork w) {
    __shared__ int bad_s;
    if (threadIdx.x == 0) bad_s = 0;
    __syncthreads();
    const long long total = block_exclusive_scan<kScanThreads>(
        in.n_rows,
        [&](int row) -> long long {
            long long cap = 0;
            const int b = in.ragged_begins[row], e = in.ragged_ends[row];
            if (b < e && (b < 0 || e > in.n_strings)) { bad_s = 1; return 0; }
            for (int col = b; col < e; ++col) {
                const long long sb = in.begins[col], se = in.ends[col];
                if (sb < 0 || se < sb || se > in.n_chars) { bad_s = 1; return 0; }
                cap += (se - sb > 0 ? se - sb : 1) * mul;
            }
            return cap;
        },
        [&](int row, long long off) { w.row_stage[row] = off > INT32_MAX ? INT32_MAX : int32_t(off); });
    if (threadIdx.x == 0) {
        w.row_stage[in.n_rows] = total > INT32_MAX ? INT32_MAX : int32_t(total);
        w.status->stage_need = total > INT32_MAX ? INT32_MAX : int32_t(total);
        uint32_t fl = 0;
        if (bad_s) fl |= kFlagRange;
        if (total > (long long)w.stage_cap) fl |= kFlagStageOverflow;
        if (fl) atomicOr(&w.status->flags, fl);
    }
}

// ---- shared by both encode modes -------------------------------------------------------------
struct RowState {
    int base;      // staging offset of the row
    int emitted;   // ids produced so far (deferred pieces excluded)
    int bytepos;   // slot units consumed so far (sum of max(len,1) of the pieces seen)
    bool slotted;
    int row;
};

__device__ __forceinline__ void make_slotted(RowState& st, const EncodeWork& w, int first_bytepos, int mul) {
    if (st.slotted) return;
    for (int k = st.emitted + lane_id(); k < first_bytepos * mul; k += kWave) w.stage[st.base + k] = kEmptyId;
    st.slotted = true;
}

__device__ __forceinline__ void push_deferred(const EncodeWork& w, int abs_begin, int len, int stage_pos, int row) {
    const int idx = atomicAdd(&w.status->n_deferred, 1);
    if (idx < w.deferred_cap) w.deferred[idx] = DeferredPiece{abs_begin, len, stage_pos, row};
    else atomicOr(&w.status->flags, kFlagDeferOverflow);
}

// A piece that never enters LDS (longer than the chunk): wave-uniform arguments.
__device__ __forceinline__ void defer_whole_piece(RowState& st, const EncodeWork& w, int abs_begin, int len, int mul) {
    make_slotted(st, w, st.bytepos, mul);
    if (lane_id() == 0) push_deferred(w, abs_begin, len, st.base + st.bytepos * mul, st.row);
    st.bytepos += len > 0 ? len : 1;
}

// One batch: lane l owns piece (toff, plen) of the LDS text (valid lanes form a prefix, their
// symbol needs fit kChunk).  abs_begin: offset of the piece in `chars` (for deferral).
__device__ __forceinline__ void process_batch(WaveScratch& ws, const BpeDev& T, const I2* root_lds, RowState& st,
                                              const EncodeWork& w, bool valid, int toff, int plen, int abs_begin) {
    const int l = lane_id();
    const int SL = T.suffix_len, mul = 1 + SL;
    const uint8_t* t = text_bytes(ws);
    const int need = valid ? plen + SL : 0;
    const int incl_need = wave_incl_sum(need);
    const int soff = incl_need - need;
    const int units = valid ? (plen > 0 ? plen : 1) : 0;
    const int incl_units = wave_incl_sum(units);
    const int bpos = st.bytepos + incl_units - units;
    uint32_t* id = ws.sym_id + soff;
    uint32_t* key = ws.sym_key + soff;

    int n = 0;
    if (valid)
        n = bpe_symbolize(T, root_lds,
                          [&](int i) -> uint32_t { return i < plen ? t[toff + i] : T.suffix[i - plen]; }, need,
                          [&](int k, int tok) { id[k] = uint32_t(tok); });
    const bool is_w = valid && n > kFastSyms;
    int res = n;
    if (valid && !is_w) res = bpe_merge_lane(T, id, key, n);
    unsigned long long wm = __ballot(is_w);
    while (wm) {
        const int src = __ffsll(wm) - 1;
        wm &= wm - 1;
        const int so = __shfl(soff, src), nn = __shfl(n, src);
        wave_sync();
        const int r = bpe_merge_wave(T, ws.sym_id + so, ws.sym_key + so, nn);
        if (l == src) res = r;
    }
    wave_sync();
    const bool defer = valid && res < 0;
    const int cnt = (valid && !defer) ? res : 0;

    if (__ballot(defer)) make_slotted(st, w, st.bytepos, mul);
    const int incl_cnt = wave_incl_sum(cnt);
    if (!st.slotted) {
        const int pos = st.base + st.emitted + incl_cnt - cnt;
        for (int k = 0; k < cnt; ++k) w.stage[pos + k] = int32_t(id[k]);
    } else if (valid) {
        const int slot = st.base + bpos * mul;
        if (defer) {
            push_deferred(w, abs_begin, plen, slot, st.row);
        } else {
            const int slot_len = (plen > 0 ? plen : 1) * mul;
            for (int k = 0; k < cnt; ++k) w.stage[slot + k] = int32_t(id[k]);
            for (int k = cnt; k < slot_len; ++k) w.stage[slot + k] = kEmptyId;
        }
    }
    st.emitted += __shfl(incl_cnt, kWave - 1);
    st.bytepos += __shfl(incl_units, kWave - 1);
}

// Number of leading lanes whose symbol needs fit the LDS arrays together (>= 1 when lane 0 is valid
// and fits; 0 when lane 0 alone does not fit -> caller defers that piece).
__device__ __forceinline__ int lanes_that_fit(bool valid, int need) {
    const int incl = wave_incl_sum(valid ? need : 0);
    const unsigned long long ok = __ballot(valid && incl <= kChunk);
    // valid lanes are a prefix and incl is monotone, so `ok` is a prefix mask
    return __popcll(ok);
}

enum EncodeMode : int { kFused = 0, kPieces = 1 };

// Whole strings as pieces (kPieces mode, and skipped strings of the fused mode): cols [c_begin, c_end).
__device__ __forceinline__ void encode_whole_strings(WaveScratch& ws, const BpeDev& T, const I2* root_lds, RowState& st,
                                                     const EncodeWork& w, const RowsIn& in, int c_begin, int c_end) {
    const int l = lane_id();
    const int SL = T.suffix_len, mul = 1 + SL;
    uint8_t* tb = reinterpret_cast<uint8_t*>(ws.text_w);
    int col = c_begin;
    while (col < c_end) {
        const int my = col + l;
        const bool cand = my < c_end;
        int sb = 0, plen = 0;
        if (cand) { sb = in.begins[my]; plen = in.ends[my] - sb; }
        const int take = lanes_that_fit(cand, plen + SL);
        if (take == 0) {  // lane 0's piece does not fit LDS: exact path
            defer_whole_piece(st, w, __shfl(sb, 0), __shfl(plen, 0), mul);
            col += 1;
            continue;
        }
        const bool valid = l < take;
        const int incl = wave_incl_sum(valid ? plen : 0);
        const int toff = incl - (valid ? plen : 0);
        wave_sync();  // the previous batch is done with the LDS text
        // short pieces: the owning lane copies; long ones: the wave copies them one by one
        const bool big = valid && plen > 32;
        if (valid && !big)
            for (int i = 0; i < plen; ++i) tb[toff + i] = in.chars[sb + i];
        unsigned long long bm = __ballot(big);
        while (bm) {
            const int src = __ffsll(bm) - 1;
            bm &= bm - 1;
            const int s_sb = __shfl(sb, src), s_len = __shfl(plen, src), s_to = __shfl(toff, src);
            for (int i = l; i < s_len; i += kWave) tb[s_to + i] = in.chars[s_sb + i];
        }
        wave_sync();
        process_batch(ws, T, root_lds, st, w, valid, toff, plen, sb);
        col += take;
    }
}

template <int MODE>
static __global__ __launch_bounds__(kBlockThreads) void encode_kernel(RowsIn in, SplitDev sp, BpeDev T, EncodeWork w) {
    __shared__ WaveScratch ws_all[kWavesPerBlock];
    __shared__ I2 root_lds[256];
    __shared__ uint8_t ascii_cls[128];
    for (int i = int(threadIdx.x); i < 256; i += kBlockThreads) root_lds[i] = T.trie.root[i];
    if (MODE == kFused && threadIdx.x < 128) ascii_cls[threadIdx.x] = uint8_t(uc_nibble(sp, threadIdx.x) & 7);
    __syncthreads();
    if (w.status->flags & (kFlagRange | kFlagStageOverflow)) return;
    WaveScratch& ws = ws_all[wave_in_block()];
    const int l = lane_id();
    const int mul = 1 + T.suffix_len;
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int row = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); row < in.n_rows; row += n_waves) {
        RowState st{w.row_stage[row], 0, 0, false, row};
        const int cb = in.ragged_begins[row], ce = in.ragged_ends[row];
        if (MODE == kPieces) {
            encode_whole_strings(ws, T, root_lds, st, w, in, cb, ce);
        } else {
            for (int col = cb; col < ce; ++col) {
                if (in.skips && in.skips[col]) {  // regex_split.cpp:231-234: passes through unsplit
                    encode_whole_strings(ws, T, root_lds, st, w, in, col, col + 1);
                    continue;
                }
                const int sb = in.begins[col], slen = in.ends[col] - sb;
                const int str_unit0 = st.bytepos;
                scan_string(
                    ws, sp, ascii_cls, in.chars + sb, slen,
                    [&](int np, int c0, int w0, int skew) {
                        for (int jb = 0; jb < np;) {
                            const int j = jb + l;
                            const bool cand = j < np;
                            int ps = 0, plen = 0;
                            if (cand) { ps = c0 + int(ws.pstart[j]); plen = c0 + int(ws.pstart[j + 1]) - ps; }
                            const int take = lanes_that_fit(cand, plen + T.suffix_len);
                            if (take == 0) {
                                const int b0 = __shfl(ps, 0), n0 = __shfl(plen, 0);
                                st.bytepos = str_unit0 + b0;
                                defer_whole_piece(st, w, sb + b0, n0, mul);
                                jb += 1;
                                continue;
                            }
                            st.bytepos = str_unit0 + __shfl(ps, 0);
                            process_batch(ws, T, root_lds, st, w, l < take, ps - w0 + skew, plen, sb + ps);
                            jb += take;
                        }
                    },
                    [&](int b, int e) {
                        st.bytepos = str_unit0 + b;
                        defer_whole_piece(st, w, sb + b, e - b, mul);
                    });
                if (slen <= 0 && st.slotted)  // an empty string produces no piece but owns one slot unit
                    for (int k = l; k < mul; k += kWave) w.stage[st.base + str_unit0 * mul + k] = kEmptyId;
                st.bytepos = str_unit0 + (slen > 0 ? slen : 1);
            }
        }
        if (l == 0) {
            w.row_cnt[row] = st.emitted;
            w.row_slotted[row] = st.slotted ? 1 : 0;
        }
    }
}

// ---- K3: path X, one lane per deferred piece.
static __global__ __launch_bounds__(kBlockThreads) void exact_kernel(RowsIn in, BpeDev T, EncodeWork w) {
    if (w.status->flags & (kFlagRange | kFlagStageOverflow)) return;
    int n = w.status->n_deferred;
    if (n > w.deferred_cap) n = w.deferred_cap;
    const int SL = T.suffix_len, mul = 1 + SL;
    const int stride = int(gridDim.x) * kBlockThreads;
    for (int i = int(blockIdx.x) * kBlockThreads + int(threadIdx.x); i < n; i += stride) {
        const DeferredPiece p = w.deferred[i];
        const int ntext = p.len + SL;
        const uint32_t bytes = (bpe_exact_scratch_bytes(uint32_t(ntext)) + 15u) & ~15u;
        const uint32_t off = atomicAdd(&w.status->scratch_used, bytes);
        if (off > w.scratch_cap || bytes > w.scratch_cap - off) {
            atomicOr(&w.status->flags, kFlagScratchOverflow);
            continue;
        }
        const uint8_t* text = in.chars + p.begin;
        int32_t* out = w.stage + p.stage_pos;
        const int cnt = bpe_exact_piece(
            T, [&](int k) -> uint32_t { return k < p.len ? text[k] : T.suffix[k - p.len]; }, ntext, w.scratch + off, out);
        const int slot_len = (p.len > 0 ? p.len : 1) * mul;
        for (int k = cnt; k < slot_len; ++k) out[k] = kEmptyId;
        atomicAdd(&w.row_cnt[p.row], cnt);
    }
}

// ---- K4: final offsets (one block).  out_begins/out_ends may be nullptr.
static __global__ __launch_bounds__(kScanThreads) void finalize_rows_kernel(int n_rows, EncodeWork w, int32_t* out_begins,
                                                                      int32_t* out_ends, long long out_cap) {
    if (w.status->flags & (kFlagRange | kFlagStageOverflow)) return;
    const long long total = block_exclusive_scan<kScanThreads>(
        n_rows, [&](int row) -> long long { return w.row_cnt[row]; },
        [&](int row, long long off) {
            w.row_out[row] = int32_t(off);
            if (out_begins) out_begins[row] = int32_t(off);
            if (out_ends) out_ends[row] = int32_t(off + w.row_cnt[row]);
        });
    if (threadIdx.x == 0) {
        w.row_out[n_rows] = int32_t(total);
        w.status->n_out = int32_t(total);
        if (total > out_cap) atomicOr(&w.status->flags, kFlagOutCapacity);
    }
}

// ---- K5: staging -> caller's buffer, one wave per row.
static __global__ __launch_bounds__(kBlockThreads) void compact_kernel(int n_rows, EncodeWork w, int32_t* out) {
    if (w.status->flags & (kFlagRange | kFlagStageOverflow | kFlagOutCapacity | kFlagDeferOverflow |
                           kFlagScratchOverflow))
        return;
    const int l = lane_id();
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int row = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); row < n_rows; row += n_waves) {
        const int base = w.row_stage[row], cap = w.row_stage[row + 1] - base;
        const int cnt = w.row_cnt[row], o = w.row_out[row];
        if (!w.row_slotted[row]) {
            for (int k = l; k < cnt; k += kWave) out[o + k] = w.stage[base + k];
        } else {
            int run = 0;
            for (int b = 0; b < cap; b += kWave) {
                const int v = (b + l < cap) ? w.stage[base + b + l] : kEmptyId;
                const unsigned long long m = __ballot(v != kEmptyId);
                if (v != kEmptyId) out[o + run + __popcll(m & lanemask_lt())] = v;
                run += __popcll(m);
            }
        }
    }
}

// ---- RegexSplit as its own op: count pass, then write pass (the scan is cheap enough to run twice).
// mode 0: row_cnt[row] = number of pieces.  mode 1: write begins/ends/skips at row_out[row].
template <int WRITE>
static __global__ __launch_bounds__(kBlockThreads) void split_kernel(RowsIn in, SplitDev sp, int max_splits, EncodeWork w,
                                                              int32_t* out_begins, int32_t* out_ends,
                                                              uint8_t* out_skips) {
    __shared__ WaveScratch ws_all[kWavesPerBlock];
    __shared__ uint8_t ascii_cls[128];
    if (threadIdx.x < 128) ascii_cls[threadIdx.x] = uint8_t(uc_nibble(sp, threadIdx.x) & 7);
    __syncthreads();
    if (w.status->flags & (kFlagRange | kFlagOutCapacity)) return;
    WaveScratch& ws = ws_all[wave_in_block()];
    const int l = lane_id();
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int row = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); row < in.n_rows; row += n_waves) {
        int count = 0;
        const int o = WRITE ? w.row_out[row] : 0;
        for (int col = in.ragged_begins[row]; col < in.ragged_ends[row]; ++col) {
            const int sb = in.begins[col], se = in.ends[col];
            if (in.skips && in.skips[col]) {
                if (WRITE && l == 0) {
                    out_begins[o + count] = sb;
                    out_ends[o + count] = se;
                    if (out_skips) out_skips[o + count] = 1;
                }
                ++count;
                continue;
            }
            int in_string = 0;  // pieces of this string so far (num_splits of regex_split.cpp:241)
            auto emit = [&](int k, int b, int e) {  // lane-local: k-th piece of the chunk
                if (!WRITE) return;
                const int idx = in_string + k;
                out_begins[o + count + idx] = sb + b;
                // regex_split.cpp:278-280: the piece whose index equals max_splits is stretched to the end
                out_ends[o + count + idx] = (idx == max_splits) ? se : sb + e;
                if (out_skips) out_skips[o + count + idx] = 0;
            };
            scan_string(
                ws, sp, ascii_cls, in.chars + sb, se - sb,
                [&](int np, int c0, int, int) {
                    for (int j = l; j < np; j += kWave) emit(j, c0 + int(ws.pstart[j]), c0 + int(ws.pstart[j + 1]));
                    in_string += np;
                },
                [&](int b, int e) {
                    if (l == 0) emit(0, b, e);
                    in_string += 1;
                });
            count += in_string;
        }
        if (!WRITE && l == 0) w.row_cnt[row] = count;
    }
}

}  // namespace ovtk
