// span_kernel.hpp -- lookup_span_kernel: the lookup stage of the GPT-2 family and of the fused WordPiece path (the BERT words), a
// 2 048-BYTE BLOCK OF TEXT PER SCAN whatever the rows are (round 4).
//
// lookup_rows_kernel scans one row per pass: a 512-byte row is 8 bytes per lane, and everything that is paid once per pass
// whatever the lane holds -- neighbour exchange, the prefix sum of the piece starts, the piece list, row bookkeeping, the last
// partly filled 64-piece batch -- is paid per row: 599 vector instructions per 512-byte row.  The kernels are bound by instruction
// issue, so here a pass covers a BLOCK, 32 bytes per lane:
//   * rows that follow each other in the chars tensor (begins[i + 1] == ends[i]: what StringTensorUnpack produces) are one
//     stretch of text, a CHAIN, and a block starts where the block before stopped -- the last piece start it could decide: whole
//     rows, the tail of one and the head of the next, or a slice of a 10 000-byte row alike.  Every lane loads its 32 bytes
//     straight from global memory into registers (two 16-byte loads at a byte address -- gfx950 takes them at any alignment,
//     tools/unaligned_probe.hip), the NEXT block's before this one's lookup rounds; the registers are the scanner's input, and
//     one copy goes to LDS for the key reads of the rounds;
//   * the row starts inside a block are one more per-byte flag (`rs`): they force a piece start and cut every look-ahead and
//     look-behind of the rules, so that a row's pieces are exactly those of the row scanned alone (src/regex_split.cpp:205-324
//     runs the pattern per string);
//   * the rules are gpt2_start_flags_ascii's (span_flags) / the BERT words' (span_flags_bert), the flags of a lane's 8 dwords
//     packed into 32 bits by eight v_dot4_u32_u8; a block with non-ASCII text takes the ballot form of the same rules, window by
//     window on the block's LDS text; a piece longer than a block is matched literally by lane 0;
//   * the pieces of the block form one list, looked up 64 at a time, the probe loads of round r + 1 in flight while round r is
//     resolved; staging positions run through the block (the wave's rows are staged back to back), a row's record is the
//     running sums at its first piece -- pulled by the row's lane from the piece's lane;
//   * misses are noted in LDS as {memo key, staging position, begin, length, byte position among the wave's rows} and become
//     DeferredPiece entries when the list fills or the wave is through.
// Rows that are not one non-empty string inside the chars tensor (empty, skipped, several strings) are marked kRowPending and
// listed for lookup_kernel<kFused>, exactly as lookup_rows_kernel does.  DESIGN.md 3.3 has the longer account.
#pragma once

#include "encode_kernels.hpp"
#include "span_l3.hpp"
#include "span_fam.hpp"

namespace ovtk {

constexpr int kSpanLane = 32;                    // text bytes per lane
constexpr int kSpanDwords = kSpanLane / 4;
constexpr int kSpanBytes = kWave * kSpanLane;    // bytes per block
// an entry of the piece list (BERT words; the GPT-2 family's entries are positions and nothing else)
constexpr uint32_t kSpanPosMask = 0x0FFFu, kSpanOneBefore = 0x4000u, kSpanDropped = 0x8000u;
constexpr int kSpanMiss = 48;                    // misses noted per wave before they are written out: the kernels that run five blocks per CU ...
constexpr int kSpanMissWide = 112;               // ... and the ones that run four (the Llama-3 family, span_fam.hpp's: 39 KB of LDS per block) -- mixed-script
                                                 // text under a 128 K vocabulary leaves ~80 pieces per wave to the store

struct SpanMiss {
    uint4 key;    // the piece's memo key (zero for pieces longer than one)
    uint4 info;   // {staging position, begin in chars, length, byte position among the wave's rows (-> the row, at flush time)}
};
template <int NM>
struct SpanWaveT {
    SpanMiss miss[NM];
    uint32_t text[1 + kSpanBytes / 4 + 8];       // one dword of padding (the window views of the ballot scanner), the block's text,
                                                 // 32 bytes behind it (the 16-byte key read of a piece at its end)
    uint16_t pstart[kSpanBytes + 4];             // piece starts, block-relative, np + 1 of them (every byte may start one); between two
                                                 // blocks its first 256 bytes collect the row-start flags
};

struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t x, y, z, w; };
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
typedef U32x4 U32x4Unaligned __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) Bytes4 { uint32_t v; };

// `sval` (wave-uniform) into lane `lane` (wave-uniform) of a per-lane value
__device__ __forceinline__ int wave_writelane(int old, int sval, int lane) { return lane_id() == lane ? sval : old; }
// 4 flag bits -> 4 bytes with the flag in bit 7
__device__ __forceinline__ uint32_t nibble_to_b7(uint32_t nib) {
    // bit k of nib lands at 7 + 7k + k = 8k + 7 for the three low bits (24-bit multiply); bit 3 is placed by hand
    return ((mul24(nib & 7u, 0x204080u) & 0x00808080u) | ((nib & 8u) << 28));
}

// The 32 bytes of lane l of the block that starts at chars[sb] and is blen bytes long (lanes behind its end: zeros).
__device__ __forceinline__ void span_load(const RowsIn& in, int sb, int blen, uint32_t (&x)[kSpanDwords]) {
    const int l = lane_id();
    const long long at = (long long)sb + kSpanLane * l;
#pragma unroll
    for (int j = 0; j < kSpanDwords; ++j) x[j] = 0;
    if (kSpanLane * l < blen) {
        if (at + kSpanLane <= in.n_chars) {
#if defined(OVTK_SIMT_EMULATOR)
            const Bytes16* p = reinterpret_cast<const Bytes16*>(in.chars + at);
            const Bytes16 a = p[0], b = p[1];
#else
            const U32x4Unaligned* p = reinterpret_cast<const U32x4Unaligned*>(in.chars + at);
            const U32x4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1);   // (read once: do not age the tables' lines)
#endif
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
            x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {   // the last lane of the chars tensor: byte by byte
            const int n = int(in.n_chars - at);
            for (int k = 0; k < n; ++k) x[k >> 2] |= uint32_t(in.chars[at + k]) << (8 * (k & 3));
        }
    }
}

// The scans: the GPT-2 family and the Llama-3 family run their rules on bit masks (span_l3.hpp: span_flags_gpt2m, span_flags_l3 -- until
// round 5 the GPT-2 family had a packed-byte form for ASCII blocks here, 3 % slower than the masks on the headline text and blind to
// non-ASCII text); the BERT words keep the packed-byte form below.
enum SpanScan : int { kSpanGpt2 = 0, kSpanGpt2Digits = 1, kSpanBertWords = 2, kSpanLlama3 = 3, kSpanDs3 = 4, kSpanO200k = 5 };
// The BERT words of the fused WordPiece path (class_packed_starts with kSplitBertWords: `\s+` removed, then every delimiter
// character -- bert_delimiter() below 0x80 -- isolated): a piece starts where white-space-ness changes, at every delimiter and
// behind every delimiter; white-space pieces are dropped (`dropped`: bit k = byte 32 l + k is white space).  Nothing looks ahead,
// so a row start only forces a start.  false (wave-uniform): the block holds a non-ASCII byte (\p{P}, the CJK blocks: the
// generic kernel's table form).
__device__ __forceinline__ bool span_flags_bert(uint32_t (&x)[kSpanDwords], uint32_t rs, uint32_t vm, uint32_t& flags, uint32_t& dropped) {
    constexpr int D = kSpanDwords;
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) any |= x[j];
    if (__ballot((any & kB7) != 0)) {
        uint32_t bad = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const uint32_t own = nibble_to_b7((vm >> (4 * j)) & 0xFu);
            bad |= x[j] & own;
            x[j] &= ~(kB7 & ~own);
        }
        if (__ballot(bad != 0)) return false;
    }
    auto classes = [](uint32_t v, uint32_t& S, uint32_t& P) {
        S = swar_eq(v, 0x20) | swar_range(v, 9, 13);
        P = swar_range(v, 0x21, 0x2F) | swar_range(v, 0x3A, 0x40) | swar_range(v, 0x5B, 0x60) | swar_range(v, 0x7B, 0x7E);
    };
    uint32_t lS, lP;
    classes(x[D - 1], lS, lP);
    uint32_t pSd = lane_prev(lS), pPd = lane_prev(lP);   // the dword in front of the lane's first
    uint32_t f_acc[D / 2], s_acc[D / 2];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        uint32_t S, P;
        if (j == D - 1) { S = lS; P = lP; }
        else classes(x[j], S, P);
        const uint32_t pS = swar_before<1>(pSd, S), pP = swar_before<1>(pPd, P);
        const uint32_t st = ((S ^ pS) | P | pP) & kB7;
        if ((j & 1) == 0) {
            f_acc[j >> 1] = dot4_u8(st, 0x08040201u, 0u);
            s_acc[j >> 1] = dot4_u8(S, 0x08040201u, 0u);
        } else {
            f_acc[j >> 1] = dot4_u8(st, 0x80402010u, f_acc[j >> 1]);
            s_acc[j >> 1] = dot4_u8(S, 0x80402010u, s_acc[j >> 1]);
        }
        pSd = S;
        pPd = P;
    }
    flags = (((f_acc[0] >> 7) | (f_acc[1] << 1) | (f_acc[2] << 9) | (f_acc[3] << 17)) | rs | (lane_id() == 0 ? 1u : 0u)) & vm;
    dropped = (s_acc[0] >> 7) | (s_acc[1] << 1) | (s_acc[2] << 9) | (s_acc[3] << 17);
    return true;
}

// The BERT words: one delimiter character, or a run of white space (`dropped`), or a run of anything else.
__device__ __forceinline__ int bert_match_end(const SplitDev& sp, const uint8_t* s, int slen, int p, bool& dropped) {
    auto kind = [&](int at, int& len) -> int {   // 0 word character, 1 white space, 2 delimiter
        const SeqChar c = seq_char(sp, s, at, slen);
        len = c.len;
        const uint32_t nib = c.cp < 0x80u ? 0u : uc_nibble(sp, c.cp);
        if (c.cls == kClsS) return 1;
        return bert_delimiter(c.cp, nib) ? 2 : 0;
    };
    int len = 1;
    const int k0 = kind(p, len);
    dropped = k0 == 1;
    int q = p + len;
    if (k0 == 2) return q;
    while (q < slen) {
        if (kind(q, len) != k0) break;
        q += len;
    }
    return q;
}

// Writes the noted misses [first, first + n) of the wave (n <= 64: a lane each) to its shard of the deferred list.  incl32: bytes of the
// wave's rows 0..l (the row of a miss is the first one whose sum lies behind the piece's position).
template <class SW>
__device__ __forceinline__ void span_flush(const SW& sw, int first, int n, const EncodeWork& w, int row0, int incl32) {
    const int l = lane_id();
    const int shard = int(blockIdx.x) % kShards;
    int idx = 0;
    if (l == 0) idx = atomicAdd(&w.status->shard_count[shard * kCounterStride], n);   // (with span_sums these are the unresolved pieces: compact_body reads the counters)
    const SpanMiss e = sw.miss[first + (l < n ? l : 0)];
    int row = 0;   // = rows whose bytes end at or before the piece
#pragma unroll
    for (int step = kWave / 2; step >= 1; step >>= 1) {
        const int t = row + step;
        if (__shfl(incl32, t - 1) <= int(e.info.w)) row = t;
    }
    idx = wave_readlane(idx, 0);
    if (l < n) {
        const int len = int(e.info.z);
        const bool keyed = len <= kPieceKeyBytes;
        const uint64_t k0 = keyed ? (uint64_t(e.key.x) | (uint64_t(e.key.y) << 32)) : 0ull;
        const uint64_t k1 = keyed ? (uint64_t(e.key.z) | (uint64_t(e.key.w) << 32)) : 0ull;
        if (idx + l < w.shard_cap)
            w.deferred[(long long)shard * w.shard_cap + idx + l] = DeferredPiece{k0, k1, int32_t(e.info.x), row0 + row, int32_t(e.info.y), len};
        else
            atomicOr(&w.status->flags, kFlagDeferOverflow);
    }
}

// One round's probe: the piece of lane l (none: plen == 0), its masked key dwords, and the candidate entry on its way.
struct SpanProbe {
    uint4 k, p;
    uint32_t a, b, c, d, mix;
    int plen, ps;
};

constexpr int kSpanHalo = 8;          // bytes at the end of a block that is not the end of its text: their flags may depend on what follows
constexpr int kSpanWindow = 16 * kWave;   // the ballot form's window (its masks travel inside a DPP row of 16 lanes)
constexpr int kSpanWindowHalo = 16;

// ---- the short path (round 6): span -> [left-over rows] -> [merge] -> compact, the two in the middle only when they are needed ----------
// Until round 5 every call was four launches: this kernel, lookup_kernel<kFused> for the rows it leaves (none, as a rule: 5 us of a
// kernel that finds nothing), merge_kernel (wordpiece_deferred_kernel) for the pieces the memo does not hold -- once the tables have
// learned a text these are pieces it finds in the piece store, nothing to merge: 13-70 us of launch, list, store probes and tile sums --
// and compact_kernel.  With EncodeWork::span_sums the span kernel does the middle's bookkeeping itself:
//   * at its end a wave looks its noted misses up in the piece store (the memo's second level: one round trip, a lane per miss;
//     store_lookup) and writes their ids into the staging entries it had reserved for them; only what the store does not hold either
//     is filed for merge_kernel: with span_sums the deferred list holds exactly the unresolved pieces (its per-shard counters say how many);
//   * it adds its rows' id counts to their tiles' sums (tile_cnt: what merge_kernel's fold_emitted_tile_sums did for every launch);
// and the host launches the kernels in the middle only when the handle's last calls say they will find work (EncodeWork::skip_mask says
// which were left out).  compact_kernel checks: rows left over (n_pending) without their kernel, or unresolved pieces without theirs,
// and it writes nothing -- the host then launches what was left out, and compact_kernel again (RowsRun::launch_phase2).  Results are
// the same either way: the piece store is the reference's piece cache (src/bpe_tokenizer.cpp:197-205, 331-338), pure memoisation.
// (Tried first and dropped, profiles/r06/a_one_pass_*: ONE launch -- a wave takes its output offset from a decoupled look-back over the
// waves in front of it and copies its own staging stretch to the final ids.  A persistent grid's waves all end together, so every
// wave waits for the slowest one in front of it, and waiting is not free: the waves that wait are the older waves of their SIMDs and
// the issue arbiter serves the oldest first -- polls every 2 us slowed the waves still at work by a fifth, whoever polled what (every
// wave its group's words; one wave per block with the others at a workgroup barrier) -- and the copies of all waves then leave in
// one burst.  93-117 us alone where span + compact take 80.)

// The store's answer for one noted miss: its ids into the staging entries the piece had reserved, the rest of them cleared.
// -> the id count, or -1 (not in the store, or no key of it fits one).
// WORDS: the store is a WordPiece handle's word store -- an entry of the one id kStoreUnk16 / kStoreUnk32 is a word without a segmentation
// and comes back as the call's unk_token_id (T.unk_id).
template <bool NARROW, bool S16, bool WORDS>
__device__ __forceinline__ int span_resolve_miss(const RowsIn& in, const BpeDev& T, const EncodeWork& w, const SpanMiss& e, int SL) {
    const int len = int(e.info.z);
    if (len < 1 || len > kStoreKeyBytes) return -1;
    uint32_t skey[8];
    if (len <= kPieceKeyBytes)
        store_key_short(uint64_t(e.key.x) | (uint64_t(e.key.y) << 32), uint64_t(e.key.z) | (uint64_t(e.key.w) << 32), len, skey);
    else
        store_key_long(in.chars + e.info.y, len, skey);
    uint32_t pay[8];
    const int c = store_lookup<NARROW>(T.store, skey, pay);
    if (c < 0 || c > len + SL) return -1;
    const int pos = int(e.info.x), need = len + SL;
    constexpr int kIds = NARROW ? kStoreIds16 : kStoreIds32;
#pragma unroll
    for (int k = 0; k < kIds; ++k)
        if (k < c) {
            int32_t id = store_id<NARROW>(pay, k);
            if (WORDS && k == 0 && id == (NARROW ? kStoreUnk16 : kStoreUnk32)) id = T.unk_id;
            if (S16) reinterpret_cast<uint16_t*>(w.stage)[pos + k] = uint16_t(id);
            else w.stage[pos + k] = id;
        }
    for (int k = c; k < need; ++k) {
        if (S16) reinterpret_cast<uint16_t*>(w.stage)[pos + k] = uint16_t(0xFFFFu);
        else w.stage[pos + k] = kEmptyId;
    }
    return c;
}

// SCAN: kSpanGpt2 / kSpanGpt2Digits (RegexSplit + BPETokenizer), kSpanBertWords (the fused WordPiece path: `T` then holds nothing but the
// word memo, misses go to wordpiece_deferred_kernel through the same deferred list).
//
// A wave's consecutive rows fall into CHAINS -- runs of rows each of which continues the text of the one before (one non-empty string
// inside the chars tensor each; any other row is left to the generic kernel) -- and a chain is worked through in BLOCKS of up to
// 2 048 bytes that start wherever the block before stopped: at the last piece start it could decide (a block that does not reach
// the chain's end cannot know where its last piece ends, nor trust the flags of its last kSpanHalo bytes).  A block therefore holds
// whole rows, the tail of a row and the head of the next, or a slice of one long row alike: rows of any length, every block full.
template <int SCAN, bool S16>
static __global__ __launch_bounds__(kBlockThreads, 4) void lookup_span_kernel(RowsIn in, SplitDev sp, BpeDev T, EncodeWork w) {
    constexpr bool DIGITS = SCAN == kSpanGpt2Digits;
    constexpr bool BERT = SCAN == kSpanBertWords;
    constexpr bool L3 = SCAN == kSpanLlama3;   // the Llama-3 family (span_l3.hpp)
    constexpr int FAM = SCAN == kSpanDs3 ? int(kFamDs3) : (SCAN == kSpanO200k ? int(kFamO200k) : 0);   // DeepSeek-V3's pattern, o200k_base (span_fam.hpp)
    constexpr int NM = (L3 || FAM != 0) ? kSpanMissWide : kSpanMiss;   // (noted misses per wave)
    using SpanWave = SpanWaveT<NM>;
    __shared__ SpanWave sw_all[kWavesPerBlock];
    __shared__ uint4 mask_tab[16];   // [n]: byte masks of the four key dwords of an n-byte piece (n = 0: nothing)
    const int l = lane_id();
    const int wave = wave_uniform(int(blockIdx.x) * kWavesPerBlock + wave_in_block());
    const int R = w.rows_per_wave;  // <= kWave
    const int row0 = wave * R;
    const int nr = in.n_rows - row0 < R ? in.n_rows - row0 : R;   // (<= 0: a wave with no rows)
    // ---- the headers of all my rows: lane i = row row0 + i.  Row i is almost always string i: its offsets are asked for together
    // with the row's string range, not behind it (one memory round trip instead of two on the wave's way to its first text).
    int cb = -1, ce = -1, g_b = 0, g_e = 0;
    uint8_t g_skip = 0;
    const int guess = row0 + l;
    if (l < nr) {
        cb = in.ragged_begins[guess];
        ce = in.ragged_ends[guess];
        if (guess < in.n_strings) {
            g_b = in.begins[guess];
            g_e = in.ends[guess];
            if (in.skips) g_skip = in.skips[guess];
        }
    }
    if (threadIdx.x < 16) {
        const int n = int(threadIdx.x);
        uint32_t m[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int keep = n - 4 * d;
            m[d] = keep >= 4 ? ~0u : (keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u));
        }
        mask_tab[n] = uint4{m[0], m[1], m[2], m[3]};
    }
    const bool fatal = (w.status->flags & kFatalFlags) != 0;
    __syncthreads();
    if (fatal || nr <= 0) return;
    SpanWave& sw = sw_all[wave_in_block()];
    PROBE(0);

    const uint8_t* text = reinterpret_cast<const uint8_t*>(sw.text + 1);
    const int SL = T.suffix_len, mul = SL + 1;
    int h_sb = 0, h_len = 0;
    bool h_simple = false;
    if (l < nr && ce == cb + 1 && cb >= 0 && cb < in.n_strings) {
        if (cb != guess) {
            g_b = in.begins[cb];
            g_e = in.ends[cb];
            g_skip = in.skips ? in.skips[cb] : uint8_t(0);
        }
        h_sb = g_b;
        h_len = g_e - g_b;
        h_simple = !g_skip && h_len > 0 && h_sb >= 0 && (long long)h_sb + h_len <= in.n_chars;
    }
    if (!h_simple) h_len = 0;
    // row l continues the stretch of text of row l - 1
    const int prev_end = int(lane_prev(uint32_t(h_sb + h_len)));
    const unsigned long long simple_m = __ballot(h_simple);
    const unsigned long long link_m = __ballot(h_simple && l > 0 && ((simple_m >> (l > 0 ? l - 1 : 0)) & 1ull) && h_sb == prev_end);
    // bytes of rows 0..l: the true total in 64 bits (it decides whether the wave's staging fits an int), the running sums in 32
    // (a wave whose total does not fit is `dead` before it uses them)
    int incl32;
    long long total_bytes;
    {
        const int lo_incl = wave_incl_sum(h_len & 0xFFFFF), hi_incl = wave_incl_sum(h_len >> 20);
        incl32 = int(uint32_t(lo_incl) + (uint32_t(hi_incl) << 20));
        total_bytes = (long long)wave_readlane(lo_incl, kWave - 1) + ((long long)wave_readlane(hi_incl, kWave - 1) << 20);
    }
    const int excl32 = incl32 - h_len;
#ifdef OVTK_PROBE
    if (lane_id() == 0 && wave < 8192) {   // where the wave runs: HW_ID (wave / SIMD / CU / SH / SE) and the XCC; the bytes of its rows
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g_ts[wave][8] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        g_ts[wave][9] = (unsigned long long)total_bytes;
    }
#endif
    // ---- chains: rows [ci, cj) that continue one another; the first one's text is asked for before the staging reservation
    int ci = 0, cj = 0, chain_sb = 0, chain_len = 0, ex0 = 0;
    uint32_t xa[kSpanDwords];
    auto next_chain = [&](int from) -> bool {
        const unsigned long long cand = from >= kWave ? 0ull : (simple_m & ~((1ull << from) - 1ull));
        if (!cand) return false;
        ci = __ffsll(cand) - 1;
        const unsigned long long above = ci >= kWave - 1 ? 0ull : ~((2ull << ci) - 1ull);
        const unsigned long long stop = ~link_m & above;
        cj = stop ? __ffsll(stop) - 1 : kWave;
        chain_sb = wave_readlane(h_sb, ci);
        ex0 = wave_readlane(excl32, ci);
        chain_len = wave_readlane(incl32, cj - 1) - ex0;
        span_load(in, chain_sb, chain_len < kSpanBytes ? chain_len : kSpanBytes, xa);
        return true;
    };
    bool have_chain = next_chain(0);
    // ---- staging: ONE reservation for all my rows (rows that end up pending leave theirs unused)
    int cursor = 0;
    bool dead = false;
    {
        const long long total = total_bytes * mul;
        const int shard = wave % kShards;
        int base = 0;
        if (total > w.stage_region) dead = true;
        if (l == 0 && total > 0 && !dead) base = atomicAdd(&w.status->stage_top[shard * kCounterStride], int(total));
        base = wave_readlane(base, 0);
        if (dead || base < 0 || base > w.stage_region - int(total)) {
            if (l == 0) atomicOr(&w.status->flags, kFlagStageOverflow);
            dead = true;   // the host grows the buffer and reruns
        }
        cursor = shard * w.stage_region + base;
    }
    const unsigned long long pending_m = dead ? ~0ull : ~simple_m;   // (bits >= nr are ignored below)
    // the rows' records collect in lane i for row i and leave with one store per array at the end
    int rec_stage = 0, rec_cnt = 0, rec_used = 0;
    int n_miss = 0;
    int emitted = 0;   // ids written by hits so far (all my rows)
    // wave-uniform call, mm = ballot(mine): the piece of `len` bytes at `begin` of the chars tensor, `wpos` bytes into the wave's rows,
    // got `need` staging entries at `at` and must go through the merge path
    auto note_miss = [&](bool mine, unsigned long long mm, int at, int begin, int len, int wpos, uint32_t ka, uint32_t kb, uint32_t kc, uint32_t kd) {
        const int add = __popcll(mm);
        const int rank = rank_below(mm);
        for (int done = 0; done < add;) {
            if (n_miss + (add - done) > NM && n_miss > 0) {
                // (more than the wave can note: these go to merge_kernel / wordpiece_deferred_kernel as they are -- span_flush counts them as unresolved)
                wave_sync();
                for (int f0 = 0; f0 < n_miss; f0 += kWave) span_flush(sw, f0, n_miss - f0 < kWave ? n_miss - f0 : kWave, w, row0, incl32);
                wave_sync();
                n_miss = 0;
            }
            const int take = add - done < NM - n_miss ? add - done : NM - n_miss;
            if (mine && rank >= done && rank < done + take)
                sw.miss[n_miss + rank - done] = SpanMiss{uint4{ka, kb, kc, kd}, uint4{uint32_t(at), uint32_t(begin), uint32_t(len), uint32_t(wpos)}};
            n_miss += take;
            done += take;
        }
    };
    while (have_chain && !dead) {
        const bool in_chain = l >= ci && l < cj;
        const int Rl = excl32 - ex0;   // where my row starts in the chain
        int pos = 0;
        while (pos < chain_len) {
            const int b_len = chain_len - pos < kSpanBytes ? chain_len - pos : kSpanBytes;
            const bool at_end = pos + b_len == chain_len;
            // ---- the text into LDS (the key reads of the rounds, the letters behind an apostrophe, the ballot form's windows)
            wave_sync();   // the previous block's rounds are done with the LDS text and piece list
            {
                uint32_t* tw = sw.text + 1 + 8 * l;   // (4 bytes off the 16-byte grid: eight dword stores)
#pragma unroll
                for (int j = 0; j < kSpanDwords; ++j) tw[j] = xa[j];
            }
            // rows that start inside the block: a flag per first byte (scattered through LDS: the piece list's room is free between
            // two blocks); and one behind the chain's last byte
            const int my_p = Rl - pos;   // my row's first byte, block-relative
            const bool starts_here = in_chain && my_p >= 0 && my_p < b_len;
            uint32_t* rs_words = reinterpret_cast<uint32_t*>(sw.pstart);
            rs_words[l] = 0;
            wave_sync();
            if (starts_here) atomicOr(&rs_words[my_p >> 5], 1u << (my_p & 31));
            if (at_end && b_len < kSpanBytes && l == 0) atomicOr(&rs_words[b_len >> 5], 1u << (b_len & 31));
            wave_sync();
            const uint32_t rs = rs_words[l];
            const int nv = b_len - kSpanLane * l;
            const uint32_t vm = nv >= kSpanLane ? ~0u : (nv <= 0 ? 0u : ((1u << nv) - 1u));
            wave_sync();
            uint32_t fl = 0, dropped = 0;
            int lim = b_len - kSpanHalo;   // a block that was cut: starts from here on may depend on what follows it
            bool fast = true;
            if constexpr (L3) {
                // the rule algebra on bit masks, whatever the script; what it does not cover (a non-ASCII digit, U+017F): lane 0, literally
                uint32_t* fl_words = rs_words + kWave;   // (the piece list's room is still free: kSpanClassScratch bytes of it)
                static_assert(kWave * 4 + kSpanClassScratch <= int(sizeof(SpanWave::pstart)), "the Llama-3 scanner's scratch lives in the piece list's room");
                if (!span_flags_l3(xa, rs, vm, text, fl_words, sp, at_end, b_len, fl, lim)) span_flags_l3_literal(rs_words, fl_words, text, sp, at_end, b_len, fl, lim);
#ifdef OVTK_SIMT_EMULATOR
                else {   // the emulator build checks the algebra against the literal matcher on every block
                    uint32_t fl0 = 0;
                    int lim0 = 0;
                    span_flags_l3_literal(rs_words, fl_words, text, sp, at_end, b_len, fl0, lim0);
                    const int dk = lim - kSpanLane * l;
                    const uint32_t below = dk >= kSpanLane ? ~0u : (dk <= 0 ? 0u : ((1u << dk) - 1u));
                    const unsigned long long bad = __ballot(((fl ^ fl0) & below) != 0);
                    if (bad || lim > lim0) {
                        const int bl = bad ? __ffsll(bad) - 1 : 0;
                        const uint32_t a = uint32_t(wave_readlane(int(fl), bl)), b = uint32_t(wave_readlane(int(fl0), bl));
                        if (l == 0) {
                            printf("span_flags_l3 differs from the literal matcher: b_len %d at_end %d und %d / %d lane %d flags %08x / %08x\n", b_len, int(at_end), lim, lim0, bl, a, b);
                            printf("  text of the lane and its neighbours:");
                            for (int i = (bl > 0 ? bl - 1 : 0) * 32; i < (bl + 2) * 32 && i < b_len; ++i) printf(" %02x", text[i]);
                            printf("\n");
                        }
                        __builtin_trap();
                    }
                }
#endif
            } else if constexpr (FAM != 0) {
                uint32_t* fl_words = rs_words + kWave;
                static_assert(kWave * 4 + kSpanFamScratch <= int(sizeof(SpanWave::pstart)), "the scanners' scratch lives in the piece list's room");
                const bool covered = FAM == kFamDs3 ? span_flags_ds3(xa, rs, vm, text, fl_words, sp, at_end, b_len, fl, lim)
                                                    : span_flags_o200k(xa, rs, vm, text, fl_words, sp, at_end, b_len, fl, lim);
                if (!covered) span_flags_fam_literal<FAM>(rs_words, fl_words, text, sp, at_end, b_len, fl, lim);
#ifdef OVTK_SIMT_EMULATOR
                else {   // the emulator build checks the algebra against the literal matcher on every block
                    uint32_t fl0 = 0;
                    int lim0 = 0;
                    span_flags_fam_literal<FAM>(rs_words, fl_words, text, sp, at_end, b_len, fl0, lim0);
                    const int dk = lim - kSpanLane * l;
                    const uint32_t below = dk >= kSpanLane ? ~0u : (dk <= 0 ? 0u : ((1u << dk) - 1u));
                    const unsigned long long bad = __ballot(((fl ^ fl0) & below) != 0);
                    if (bad || lim > lim0) {
                        const int bl = bad ? __ffsll(bad) - 1 : 0;
                        const uint32_t a = uint32_t(wave_readlane(int(fl), bl)), b = uint32_t(wave_readlane(int(fl0), bl));
                        if (l == 0) {
                            printf("span_flags of family %d differ from the literal matcher: b_len %d at_end %d und %d / %d lane %d flags %08x / %08x\n", FAM, b_len, int(at_end), lim, lim0, bl, a, b);
                            printf("  text of the lane and its neighbours:");
                            for (int i = (bl > 0 ? bl - 1 : 0) * 32; i < (bl + 2) * 32 && i < b_len; ++i) printf(" %02x", text[i]);
                            printf("\n");
                        }
                        __builtin_trap();
                    }
                }
#endif
            } else {
                if constexpr (BERT) fast = span_flags_bert(xa, rs, vm, fl, dropped);
                else span_flags_gpt2m<DIGITS>(xa, rs, vm, text, rs_words + kWave, sp, at_end, b_len, fl);   // (any text: the characters of a
                                                                                                            // non-ASCII block are classified by the wave)
            }
            // ---- the block's piece list: np pieces, the last one ends at q_end (= where the next block starts); rowfirst: the list
            // index of my row's first piece, if that is one of them
            int np = 0, q_end = 0, rowfirst = 0x7FFFFFFF;
            uint32_t end_flag = 0;   // BERT words: the sentinel's kSpanOneBefore (below)
            if (fast) {
                q_end = b_len;
                const uint32_t fl_all = fl;
                if (!at_end) {   // the last start the block can decide ends its last whole piece
                    const int dk = lim - kSpanLane * l;
                    fl &= dk >= kSpanLane ? ~0u : (dk <= 0 ? 0u : ((1u << dk) - 1u));
                    const unsigned long long nz = __ballot(fl != 0);
                    const int hl = 63 - __clzll(nz);   // (lane 0 holds the block's first byte: never empty)
                    q_end = hl * kSpanLane + 31 - __clz(uint32_t(wave_readlane(int(fl), hl)));
                    const int qk = q_end - kSpanLane * l;
                    fl &= qk >= kSpanLane ? ~0u : (qk <= 0 ? 0u : ((1u << qk) - 1u));
                }
                if (q_end > 0) {
                    // BERT words: the one blank between two words is most of the dropped pieces, and half of all pieces.  A white-space
                    // piece of one byte with a word or delimiter in front of it gets no entry; the entry behind it says so
                    // (kSpanOneBefore: the piece in front ends one byte early).  `dropped` = the white-space bytes.
                    uint32_t one_before = 0;
                    if (BERT) {
                        const uint32_t s_prev = (dropped << 1) | (lane_prev(dropped) >> 31);
                        const uint32_t f_next = (fl_all >> 1) | (lane_next(fl_all) << 31);   // (fl_all: the flags before q_end cut them)
                        const uint32_t omit = fl_all & dropped & ~s_prev & f_next;
                        one_before = (omit << 1) | (lane_prev(omit) >> 31);
                        fl &= ~omit;
                        if (q_end < kSpanBytes) end_flag = (uint32_t(wave_readlane(int(one_before), q_end >> 5)) >> (q_end & 31)) & 1u;
                    }
                    const int cnt = __popc(fl);
                    const int p_incl = wave_incl_sum(cnt);
                    const int at0 = p_incl - cnt;
                    np = wave_readlane(p_incl, kWave - 1);
                    {
                        uint32_t f = fl;
                        uint16_t* at = sw.pstart + at0;
                        while (f) {
                            const int bit = __ffs(f) - 1;
                            uint32_t e = uint32_t(kSpanLane * l + bit);
                            if (BERT) e |= (((dropped >> bit) & 1u) ? kSpanDropped : 0u) | (((one_before >> bit) & 1u) ? kSpanOneBefore : 0u);
                            *at++ = uint16_t(e);
                            f &= f - 1;
                        }
                    }
                    // my row's first piece: the pieces of the lanes in front of its first byte's lane, and of that lane's bytes in front
                    const int ln = (my_p >> 5) & (kWave - 1);
                    const int first = __shfl(at0, ln) + __popc(uint32_t(__shfl(int(fl), ln)) & ((1u << (my_p & 31)) - 1u));
                    if (starts_here && my_p < q_end) rowfirst = first;
                }
            } else if constexpr (BERT) {
                // A block with non-ASCII text: row by row, window by window (slices of a long row) through the ballot form of the rules on
                // the block's LDS text -- a byte per lane and 64-byte word, code points through the Unicode tables, windows of up to
                // 1 024 bytes: four times the packed form's instructions per byte.  Every window starts at a true piece start and
                // decides the pieces up to its last start; the next window -- same row or next row -- starts there.  The rows stay in
                // this kernel, the lookup rounds are shared (round 3 left such rows to the generic kernel: VERDICT r03 missing 2).
                int a = 0, k = __ffsll(__ballot(in_chain && my_p <= 0 && my_p + h_len > 0)) - 1;   // the row that holds the block's first byte
                for (;;) {
                    const int r_begin = wave_readlane(Rl, k) - pos;   // (negative: the row began in a block before)
                    const int r_end = r_begin + wave_readlane(h_len, k);
                    int b = r_end < b_len ? r_end : b_len;
                    if (b - a > kSpanWindow) b = a + kSpanWindow;
                    const bool row_ends = b == r_end;
                    const bool cut = !row_ends && b == b_len;   // the block's text ends inside the row
                    if (cut && a > 0 && b - a < kSpanWindow / 2) break;   // too little of the row left in this block: the next one starts here
                    const int wl = b - a;
                    int q = wl, npw = 0;   // the window's pieces: npw of them, the last one ends at window byte q
                    const WsView view{sw.text + (a >> 2), sw.pstart};
                    Mask start, drop = 0;
                    if (BERT) class_start_mask(view, sp, a & 3, wl, start, drop);
                    else start = gpt2_start_mask(view, sp, a & 3, wl, DIGITS);
                    const int dec = row_ends ? wl : wl - kSpanWindowHalo;   // window bytes whose flags are decided
                    {
                        const int dk = dec - 64 * l;
                        start &= dk >= 64 ? ~0ull : (dk <= 0 ? 0ull : ((1ull << dk) - 1ull));
                        if (l == 0) start |= 1ull;
                    }
                    if (!row_ends) {
                        const unsigned long long nz = __ballot(start != 0);
                        const int hw = 63 - __clzll(nz);
                        q = hw * 64 + 63 - __clzll(wave_readlane(start, hw));
                        const int qk = q - 64 * l;
                        start &= qk >= 64 ? ~0ull : (qk <= 0 ? 0ull : ((1ull << qk) - 1ull));
                    }
                    for (int wd = 0; wd * 64 < q; ++wd) {
                        const Mask m = wave_readlane(start, wd);
                        const Mask dm = BERT ? wave_readlane(drop, wd) : 0ull;
                        if ((m >> l) & 1ull)
                            sw.pstart[np + npw + rank_below(m)] = uint16_t((a + wd * 64 + l) | (((dm >> l) & 1ull) ? kSpanDropped : 0u));
                        npw += __popcll(m);
                    }
                    if (q == 0) break;   // one piece fills the window: matched literally, from a block that starts with it
                    if (a == r_begin) rowfirst = wave_writelane(rowfirst, np, k);
                    np += npw;
                    a += q;
                    q_end = a;
                    if (cut || a >= b_len) break;
                    if (a == r_end) ++k;   // the next row starts at a (else: the same row's next window)
                }
            }
            if (q_end == 0) {
                // ---- one piece longer than a block (or a window): its end by the literal matcher, itself straight to the deferred list
                const int k = __ffsll(__ballot(in_chain && my_p <= 0 && my_p + h_len > 0)) - 1;
                const int rlen = wave_readlane(h_len, k);
                const int p = pos - wave_readlane(Rl, k);
                const uint8_t* row_text = in.chars + wave_readlane(h_sb, k);
                bool drop1 = false;
                int e = 0;
                if (l == 0) e = BERT ? bert_match_end(sp, row_text, rlen, p, drop1) : (L3 ? llama3_match_end(sp, row_text, rlen, p) : (FAM != 0 ? fam_match_end<(FAM != 0 ? FAM : 1)>(sp, row_text, rlen, p) : gpt2_match_end(sp, row_text, rlen, p, DIGITS)));
                e = wave_readlane(e, 0);
                drop1 = wave_readlane(int(drop1), 0) != 0;
                const int plen = e - p;
                if (p == 0) {
                    rec_stage = wave_writelane(rec_stage, cursor, k);
                    rec_cnt = wave_writelane(rec_cnt, emitted, k);
                }
                if (!drop1) {
                    // (longer than a block as a rule; but a block that starts inside a long white-space run decides nothing but its
                    // first byte, and what starts there may be `\n` alone: a piece short enough for a memo key carries it)
                    uint64_t k0 = 0, k1 = 0;
                    if (plen <= kPieceKeyBytes && l == 0) {
                        uint64_t r0 = 0, r1 = 0;
                        global_bytes16(row_text + p, plen, r0, r1);
                        piece_key(r0, r1, plen, k0, k1);
                    }
                    note_miss(l == 0, 1ull, cursor, chain_sb + pos, plen, ex0 + pos, uint32_t(k0), uint32_t(k0 >> 32), uint32_t(k1), uint32_t(k1 >> 32));
                    cursor += plen + SL;
                }
                pos += plen;
                span_load(in, chain_sb + pos, chain_len - pos < kSpanBytes ? chain_len - pos : kSpanBytes, xa);
                continue;
            }
            if (l < 2) sw.pstart[np + l] = uint16_t(uint32_t(q_end) | (l == 0 ? end_flag * kSpanOneBefore : 0u));   // (two: lane j >= np reads a piece of no bytes)
            // ---- the next block's text: in flight while this block's pieces are looked up
            {
                const int nx = pos + q_end;
                span_load(in, chain_sb + nx, chain_len - nx < kSpanBytes ? chain_len - nx : kSpanBytes, xa);
            }
            wave_sync();
            // ---- rounds of 64 pieces; the probe of the next round is in flight while this one is resolved
            const int b_begin = chain_sb + pos, b_wpos = ex0 + pos;
            const char* slots = reinterpret_cast<const char*>(T.pieces.slots);
            const uint32_t slot_shift = T.pieces.shift - 5u;   // (slot index x 32 bytes: a table has at most 2^27 slots)
            // The probe of list entry j (an index behind the list's end finds a piece of no bytes: the sentinel entries).
            auto fetch_at = [&](int j_raw) -> SpanProbe {
                SpanProbe q;
                const int j = j_raw < np ? j_raw : np;
                const uint32_t pp = reinterpret_cast<const Bytes4*>(sw.pstart + j)->v;
                if (BERT) {
                    q.ps = int(pp & kSpanPosMask);
                    const int end = int((pp >> 16) & kSpanPosMask) - int((pp >> 30) & 1u);   // (kSpanOneBefore of the next entry)
                    q.plen = (pp & kSpanDropped) ? 0 : end - q.ps;   // (a dropped piece: nothing to look up)
                } else {
                    q.ps = int(pp & 0xFFFFu);
                    q.plen = int(pp >> 16) - q.ps;
                }
                const Bytes16 r = *reinterpret_cast<const Bytes16*>(text + q.ps);
                const uint4 m = mask_tab[q.plen < 15 ? q.plen : 15];
                q.a = r.x & m.x;
                q.b = r.y & m.y;
                q.c = r.z & m.z;
                q.d = (r.w & m.w) | (uint32_t(q.plen) << 24);
                q.mix = piece_mix((uint64_t(q.b) << 32) | q.a, (uint64_t(q.d) << 32) | q.c);
                const uint4* e = reinterpret_cast<const uint4*>(slots + ((q.mix >> slot_shift) & ~31u));   // (a 32-bit offset: tables.hpp piece_h)
                q.k = e[0];
                q.p = e[1];
                return q;
            };
            // six u16 ids per entry: the word memo when the vocabulary fits (wave-uniform), the BPE memo exactly when the staging
            // entries are u16 (api_encode.cpp build_memo: packed6 = stage16 -- a template flag here)
            const bool packed6 = BERT ? T.pieces.packed6 != 0 : S16;
            // hit or miss, the ids a hit brings, the staging entries the piece takes
            auto classify = [&](const SpanProbe& q, bool& hit, int& cnt_ids, int& need) {
                const bool valid = q.plen >= 1;
                uint32_t xk = (q.k.x ^ q.a) | (q.k.y ^ q.b) | (q.k.z ^ q.c) | (q.k.w ^ q.d);
#ifndef OVTK_SIMT_EMULATOR
                asm volatile("" : "+v"(xk));   // (one compare, not four: memo_resolve)
#endif
                const uint32_t c0 = q.p.w ^ piece_tag(q.mix, 0);
                hit = valid && q.plen <= kPieceKeyBytes && xk == 0u && c0 <= uint32_t(packed6 ? kPieceMaxIds6 : kPieceMaxIds);
                cnt_ids = hit ? int(c0) : 0;
                need = hit ? cnt_ids : (valid ? q.plen + SL : 0);
            };
            auto put_ids = [&](const SpanProbe& q, bool hit, int cnt_ids, int at) {
                if (!BERT && S16) {
                    // The BPE memo's six u16 ids into u16 staging entries: a pair as one dword (at a 2-byte address: gfx950 stores
                    // at any alignment), an odd count's last id alone; every store at a constant offset from ONE address (an offset
                    // that depends on the count is a second 64-bit address: two registers this kernel does not have -- 97 VGPRs,
                    // four waves per SIMD instead of five, twelve spilled SGPRs).  (cnt_ids == 0: a miss, nothing to store)
                    uint16_t* st16 = reinterpret_cast<uint16_t*>(w.stage) + at;
                    if (cnt_ids == 1) st16[0] = uint16_t(q.p.x);
                    if (cnt_ids > 1) reinterpret_cast<Bytes4*>(st16)->v = q.p.x;
                    if (cnt_ids == 3) st16[2] = uint16_t(q.p.y);
                    if (cnt_ids > 3) reinterpret_cast<Bytes4*>(st16 + 2)->v = q.p.y;
                    if (cnt_ids == 5) st16[4] = uint16_t(q.p.z);
                    if (cnt_ids > 5) reinterpret_cast<Bytes4*>(st16 + 4)->v = q.p.z;
                } else if (hit && packed6) {
                    const uint32_t i0 = q.p.x & 0xFFFFu, i1 = q.p.x >> 16, i2 = q.p.y & 0xFFFFu, i3 = q.p.y >> 16, i4 = q.p.z & 0xFFFFu, i5 = q.p.z >> 16;
                    if (S16) {
                        uint16_t* st16 = reinterpret_cast<uint16_t*>(w.stage) + at;
                        if (cnt_ids > 0) st16[0] = uint16_t(i0);
                        if (cnt_ids > 1) st16[1] = uint16_t(i1);
                        if (cnt_ids > 2) st16[2] = uint16_t(i2);
                        if (cnt_ids > 3) {
                            st16[3] = uint16_t(i3);
                            if (cnt_ids > 4) st16[4] = uint16_t(i4);
                            if (cnt_ids > 5) st16[5] = uint16_t(i5);
                        }
                    } else {
                        int32_t* st32 = w.stage + at;
                        if (cnt_ids > 0) st32[0] = int32_t(i0);
                        if (cnt_ids > 1) st32[1] = int32_t(i1);
                        if (cnt_ids > 2) st32[2] = int32_t(i2);
                        if (cnt_ids > 3) {
                            st32[3] = int32_t(i3);
                            if (cnt_ids > 4) st32[4] = int32_t(i4);
                            if (cnt_ids > 5) st32[5] = int32_t(i5);
                        }
                    }
                } else if (hit) {
                    if (S16) {   // (the staging entries are u16: EncodeWork::stage16, a template flag here -- one branch less per round)
                        uint16_t* st16 = reinterpret_cast<uint16_t*>(w.stage) + at;
                        if (cnt_ids > 0) st16[0] = uint16_t(q.p.x);
                        if (cnt_ids > 1) st16[1] = uint16_t(q.p.y);
                        if (cnt_ids > 2) st16[2] = uint16_t(q.p.z);
                    } else {
                        int32_t* st32 = w.stage + at;
                        if (cnt_ids > 0) st32[0] = int32_t(q.p.x);
                        if (cnt_ids > 1) st32[1] = int32_t(q.p.y);
                        if (cnt_ids > 2) st32[2] = int32_t(q.p.z);
                    }
                }
            };
            // (rounds of 128 pieces, two per lane, were measured in round 5 and lose: 60 registers of probes in flight cost a wave per SIMD;
            // profiles/r05/experiments/two_pieces_per_lane.*)
            {
            auto resolve = [&](const SpanProbe& q, int jb) {
                bool hit;
                int cnt_ids, need;
                classify(q, hit, cnt_ids, need);
                const int v = need | (cnt_ids << 16);
                const int s_incl = wave_incl_sum(v);
                const int s_excl = s_incl - v;
                const int at = cursor + (s_excl & 0xFFFF);
                put_ids(q, hit, cnt_ids, at);
                // a row whose first piece lies in this round: its records are the running sums at that piece (lane = row pulls them
                // from lane = piece)
                {
                    const uint32_t d = uint32_t(rowfirst - jb);
                    const uint32_t sums = uint32_t(__shfl(s_excl, int(d & (kWave - 1))));
                    if (d < uint32_t(kWave)) {
                        rec_stage = cursor + int(sums & 0xFFFFu);
                        rec_cnt = emitted + int(sums >> 16);
                    }
                }
                const bool miss = q.plen >= 1 && !hit;
                const unsigned long long mm = __ballot(miss);
                if (mm) note_miss(miss, mm, at, b_begin + q.ps, q.plen, b_wpos + q.ps, q.a, q.b, q.c, q.d);
                const uint32_t tot = uint32_t(wave_readlane(s_incl, kWave - 1));
                cursor += int(tot & 0xFFFFu);
                emitted += int(tot >> 16);
            };
            // (a fetch behind the list's end finds pieces of no bytes: inside the loop nothing is conditional, so that no wait for
            // "a load that may still be on its way" ends up in front of the next fetch)
            SpanProbe qa = fetch_at(l);
            int jb = 0;
            for (; jb + kWave < np; jb += 2 * kWave) {
                const SpanProbe qb = fetch_at(jb + kWave + l);
                resolve(qa, jb);
                qa = fetch_at(jb + 2 * kWave + l);
                resolve(qb, jb + kWave);
            }
            if (jb < np) resolve(qa, jb);
            }
            if (BERT && rowfirst == np) {   // a row whose pieces in this block got no entry (blanks): its records are the sums behind the list
                rec_stage = cursor;
                rec_cnt = emitted;
            }
            pos += q_end;
        }
        // ---- the chain's rows: entries used = up to the next row's first entry, ids = the hits' ids in between
        {
            const int nx_stage = int(lane_next(uint32_t(rec_stage))), nx_cnt = int(lane_next(uint32_t(rec_cnt)));
            const int end_stage = l == cj - 1 ? cursor : nx_stage;
            const int end_cnt = l == cj - 1 ? emitted : nx_cnt;
            if (in_chain) {
                rec_used = end_stage - rec_stage;
                rec_cnt = end_cnt - rec_cnt;
            }
        }
        have_chain = next_chain(cj);
    }
    PROBE(1);
    if (w.span_sums && n_miss > 0 && T.store.slots && !dead) {
        // ---- the short path: the noted misses through the piece store, a lane each, 64 at a time
        uint32_t* radd = reinterpret_cast<uint32_t*>(sw.pstart);   // the rows' id counts go through LDS (the piece list's room is free)
        wave_sync();
        radd[l] = 0;
        int n_left = 0;   // entries [0, n_left) of the list: what the store does not hold
        for (int f0 = 0; f0 < n_miss; f0 += kWave) {
            const int n = n_miss - f0 < kWave ? n_miss - f0 : kWave;
            wave_sync();
            const SpanMiss e = sw.miss[f0 + (l < n ? l : 0)];
            int row = 0;   // the row of my miss (span_flush's search)
#pragma unroll
            for (int step = kWave / 2; step >= 1; step >>= 1) {
                const int t = row + step;
                if (__shfl(incl32, t - 1) <= int(e.info.w)) row = t;
            }
            int c = -1;
            if (l < n) {
                if (S16) c = span_resolve_miss<true, S16, BERT>(in, T, w, e, SL);
                else if (T.store.narrow) c = span_resolve_miss<true, S16, BERT>(in, T, w, e, SL);
                else c = span_resolve_miss<false, S16, BERT>(in, T, w, e, SL);
            }
            if (c > 0) atomicAdd(&radd[row], uint32_t(c));
            const unsigned long long um = __ballot(l < n && c < 0);
            // what the store did for this call decides whether the next ones ask it at all (api_encode.cpp store_pause): a SAMPLE of the
            // waves counts, as in merge_kernel -- which now sees only what the store does not hold
            if ((wave & 63) == 0 && l == 0) {
                atomicAdd(&w.status->n_store_probe, n);
                if (n - int(__popcll(um)) > 0) atomicAdd(&w.status->n_store_hit, n - int(__popcll(um)));
            }
            wave_sync();   // (every lane holds its entry in registers: what is moved to the front overwrites nothing unread)
            if (l < n && c < 0) sw.miss[n_left + rank_below(um)] = e;
            n_left += __popcll(um);
        }
        wave_sync();
        rec_cnt += int(radd[l]);
        n_miss = n_left;
    }
    if (n_miss > 0) {
        wave_sync();
        for (int f0 = 0; f0 < n_miss; f0 += kWave) span_flush(sw, f0, n_miss - f0 < kWave ? n_miss - f0 : kWave, w, row0, incl32);
    }
    PROBE(2);
    const bool is_pending = l < nr && ((pending_m >> l) & 1ull);
    if (l < nr) {
        w.row_used[row0 + l] = is_pending ? kRowPending : rec_used;
        if (!is_pending) {
            w.row_stage[row0 + l] = rec_stage;
            w.row_cnt[row0 + l] = rec_cnt;
            if (w.row_emit) w.row_emit[row0 + l] = rec_cnt;
        }
    }
    const unsigned long long pm = __ballot(is_pending);
    if (pm) {   // the rows left to the generic kernel: counted, and listed for it
        int base = 0;
        if (l == 0) base = atomicAdd(&w.status->n_pending, int(__popcll(pm)));
        base = wave_readlane(base, 0);
        if (w.pending_rows && is_pending) w.pending_rows[base + rank_below(pm)] = row0 + l;
    }
    if (w.span_sums) {
        // ---- the short path: my rows' id counts to their tiles' sums (R <= 64 consecutive rows: two tiles at most)
        const int my_cnt = l < nr && !is_pending ? rec_cnt : 0;
        const int t0 = row0 / kRowTile;
        const int all = wave_sum(my_cnt), first = wave_sum((row0 + l) / kRowTile == t0 ? my_cnt : 0);
        if (l == 0) {
            if (first) atomicAdd(&w.tile_cnt[t0], first);
            if (all - first) atomicAdd(&w.tile_cnt[t0 + 1], all - first);
        }
    }
}

}  // namespace ovtk
