// api_ops.cpp -- C-ABI entry points of WordpieceTokenizer, VocabEncoder, RaggedToDense, VocabDecoder,
// ByteFallback, FuzeRagged and the fused detokenizer.  Compiled as HIP (hipcc -x hip).
// Reference behaviour replaced: src/wordpiece_tokenizer.cpp:49-133, src/vocab_encoder.cpp:55-94,
// src/ragged_to_dense.cpp:70-174, src/vocab_decoder.cpp:23-87, src/byte_fallback.cpp:16-50, src/fuze.cpp:20-40.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "api_common.hpp"
#include "ops_kernels.hpp"
#include "span_kernel.hpp"
#include "runtime.hpp"
#include "tables.hpp"

using namespace ovtk;

namespace {

struct TrieBufs {
    DevBuf root, edges;
    int upload(const TrieHost& t, TrieDev& d) {
        int e = 0;
        e = e ? e : root.upload(t.root.data(), t.root.size() * sizeof(I2));
        e = e ? e : edges.upload(t.edges.data(), t.edges.size() * sizeof(TrieEdge));
        if (e) return e;
        d.root = root.as<I2>();
        d.edges = edges.as<TrieEdge>();
        d.edge_mask = t.edge_mask;
        d.edge_shift = t.edge_shift;
        return OVTK_OK;
    }
};

int check_strings_arg(const ovtk_strings* s, const char* what) {
    if (!s) return set_error(OVTK_E_ARG, std::string(what) + ": null argument");
    if (s->n < 0 || s->n_chars < 0) return set_error(OVTK_E_ARG, std::string(what) + ": negative size");
    if (s->n >= INT32_MAX || s->n_chars >= INT32_MAX)
        return set_error(OVTK_E_ARG, std::string(what) + ": tensor sizes must fit int32 offsets");
    return OVTK_OK;
}

// Status block handling of the ops that do not go through run_rows_to_ids.
int begin_status(Workspace& ws, hipStream_t s, RunStatus** st) {
    if (!ws.host_status) return set_error(OVTK_E_HIP, "pinned host allocation failed");
    if (int rc = ws.status.ensure(sizeof(RunStatus))) return rc;
    *st = ws.status.as<RunStatus>();
    OVTK_HIP(hipMemsetAsync(*st, 0, sizeof(RunStatus), s));
    return OVTK_OK;
}

// Exclusive scan of len(i), i < n, then apply(i, offset, len).
template <class LenF, class ApplyF>
int scan_and_apply(Workspace& ws, hipStream_t s, long long n, LenF len, ApplyF apply, long long cap, RunStatus* st,
                   const char* tag) {
    if ((n + kTileElems - 1) / kTileElems > INT32_MAX) return set_error(OVTK_E_UNSUPPORTED, "too many elements for one call; split it");
    if (int rc = ws.tiles.ensure(scan_tiles_bytes(n))) return rc;
    launch_scan(ws.marks, tag, s, n, len, apply, CharsFin{st, cap}, ws.tiles.as<long long>(), st,
                kFlagOutCapacity | kFlagRange);
    return OVTK_OK;
}

// VocabDecoder / fused detokenizer: count pass -> scan -> write pass over (row, segment) units.
int decode_passes(Workspace& ws, hipStream_t s, int device, const DecodeDev& d, int64_t batch, int64_t seq, int32_t* tok_begins,
                  int32_t* tok_ends, int32_t* row_begins, int32_t* row_ends, uint8_t* chars, long long cap, RunStatus* st,
                  const char* tag) {
    const int n_seg = int((seq + kSegTokens - 1) / kSegTokens);
    const long long n_units = batch * n_seg;
    if (int rc = ws.gen[2].ensure(size_t(n_units) * sizeof(long long))) return rc;
    if (int rc = ws.gen[3].ensure(size_t(n_units) * sizeof(long long))) return rc;
    if (int rc = ws.tiles.ensure(scan_tiles_bytes(n_units))) return rc;
    long long* unit_bytes = ws.gen[2].as<long long>();
    long long* unit_off = ws.gen[3].as<long long>();
    const int grid = int(std::min<long long>((n_units + kWavesPerBlock - 1) / kWavesPerBlock, (long long)device_cu_count(device) * 8));
    if (d.vocab_size <= kLenLdsTokens)
        OVTK_LAUNCH(ws.marks, "decode_count", decode_count_lds_kernel,
                    int(std::min<long long>((n_units + kCountLdsThreads / kWave - 1) / (kCountLdsThreads / kWave), (long long)device_cu_count(device))),
                    kCountLdsThreads, s, d, int(seq), n_seg, n_units, unit_bytes);
    else
        OVTK_LAUNCH(ws.marks, "decode_count", decode_count_kernel, grid, kBlockThreads, s, d, int(seq), n_seg, n_units, unit_bytes);
    launch_scan(ws.marks, "decode_scan", s, n_units, UnitLen{unit_bytes}, UnitApply{unit_off, n_seg, row_begins, row_ends},
                CharsFin{st, cap}, ws.tiles.as<long long>(), st, kFlagOutCapacity | kFlagRange);
    OVTK_LAUNCH(ws.marks, tag, decode_write_kernel, grid, kBlockThreads, s, d, int(seq), n_seg, n_units,
                (const long long*)unit_off, (const long long*)unit_bytes, tok_begins, tok_ends, chars, (const RunStatus*)st);
    return OVTK_OK;
}

}  // namespace

// ------------------------------------------------------------------------------- handles
struct ovtk_wordpiece {
    int device = 0;
    WordpieceDev dev{};
    TrieBufs root, sub;
    DevBuf memo_buf, memo_room, store, store_room;
    int32_t store_capacity = 0;
    PieceTableDev memo{nullptr, 30, nullptr, 0, 0};  // word -> ids of every vocabulary word (the fused path's first-level lookup)
    int64_t n_vocab = 0;   // ids are vocabulary indices: below 65535 (and unk_token_id too), a call stages u16 entries
    std::atomic<int> expect_pending{0}, expect_merge{kShortPathKeep}, last_unresolved{-1};   // the short path's predictors (api_encode.cpp ovtk_bpe says how they count)
};

struct ovtk_vocab_encoder {
    int device = 0;
    int value_size = 4;
    StringMapDev dev{};
    long long n_key_chars = 0;
    DevBuf slots, kb, ke, kc, values;
};

struct ovtk_vocab_decoder {
    int device = 0;
    int32_t vocab_size = 0;
    int32_t max_token_len = 0;
    DevBuf vb, vc, len_plain, pack_plain, len_bf, pack_bf;  // len_*: output bytes per token, 0 for the attribute's skip_tokens
    std::vector<uint16_t> len_plain_host, len_bf_host;      // without any skips: input 4 of a call replaces the attribute
};

extern "C" {

// ------------------------------------------------------------------------------- WordpieceTokenizer
int ovtk_wordpiece_create(const ovtk_wordpiece_params* p, ovtk_wordpiece** out) {
    if (!p || !out) return set_error(OVTK_E_ARG, "wordpiece: null argument");
    if (int rc = check_strings_arg(&p->vocab, "wordpiece vocab")) return rc;
    if (int rc = use_device(p->device)) return rc;
    auto h = std::make_unique<ovtk_wordpiece>();
    h->device = p->device;
    TrieHost root, sub;
    std::string err;
    const std::string si(p->suffix_indicator ? p->suffix_indicator : "", size_t(p->suffix_indicator ? p->suffix_indicator_len : 0));
    if (int rc = build_wordpiece(view_of(p->vocab), si, root, sub, err)) return set_error(rc, err);
    if (int rc = h->root.upload(root, h->dev.root)) return rc;
    if (int rc = h->sub.upload(sub, h->dev.sub)) return rc;
    OVTK_HIP(hipStreamSynchronize(nullptr));
    h->dev.max_bytes = p->max_bytes_per_word;
    h->n_vocab = p->vocab.n;
    // Word memo for ovtk_wordpiece_encode_run: WordPiece(t) of every vocabulary string t taken as a word, computed by the
    // device op itself (one word per row).  No entry depends on unk_token_id: a word that needs unk is simply not stored.
    if (p->vocab.n > 0) {
        const size_t V = size_t(p->vocab.n);
        std::vector<int32_t> rb(V), re(V), ob(V), oe(V);
        for (size_t i = 0; i < V; ++i) {
            rb[i] = int32_t(i);
            re[i] = int32_t(i + 1);
        }
        const int64_t cap = p->vocab.n_chars + p->vocab.n;
        std::vector<int32_t> ids(static_cast<size_t>(cap) + 1);
        ovtk_ragged_strings in{rb.data(), re.data(), p->vocab.n, p->vocab};
        ovtk_ragged_i32_out o{ob.data(), oe.data(), ids.data(), cap, 0, 0};
        const int32_t marker = INT32_MIN + 1;  // stands for unk while the memo is built
        if (int rc = ovtk_wordpiece_run(h.get(), &in, marker, &o, OVTK_MEM_HOST, nullptr)) return rc;
        const bool packed6 = p->vocab.n <= 65535;   // ids are vocabulary indices: six u16 ids per memo entry instead of three i32
        for (size_t i = 0; i < V; ++i)  // drop the words that produced unk (an entry with more ids than the table takes is never stored)
            for (int32_t k = ob[i]; k < oe[i]; ++k)
                if (ids[size_t(k)] == marker) { oe[i] = ob[i] + kPieceMaxIds6 + 1; break; }
        // ... and room for as many words again that the text brings along (up to three ids, up to 15 bytes: wordpiece_deferred_kernel
        // files them the first time it resolves them; the reference has no counterpart -- it walks its tries for every word,
        // wordpiece_tokenizer.cpp:94-130 -- and the result is the same function of the word either way)
        const int32_t learn = int32_t(std::min<int64_t>(std::max<int64_t>(4 * p->vocab.n, 32768), 1 << 20));   // (as many as the word store may hold: a lexicon has more words of several pieces than the vocabulary has words)
        PieceTableHost host;
        build_piece_table(view_of(p->vocab), ob.data(), oe.data(), ids.data(), host, size_t(learn), packed6);
        if (int rc = h->memo_buf.upload(host.slots.data(), host.slots.size() * sizeof(PieceEntry))) return rc;
        std::vector<int32_t> rooms(size_t(kRoomShards) * kRoomStride, 0);
        for (int k = 0; k < kRoomShards; ++k) rooms[size_t(k) * kRoomStride] = learn / kRoomShards + (k < learn % kRoomShards ? 1 : 0);
        if (int rc = h->memo_room.upload(rooms.data(), rooms.size() * sizeof(int32_t))) return rc;
        OVTK_HIP(hipStreamSynchronize(nullptr));
        h->memo = PieceTableDev{h->memo_buf.as<PieceEntry>(), host.shift, h->memo_room.as<int32_t>(), uint32_t(kRoomShards - 1), packed6 ? 1u : 0u};
        h->dev.memo = h->memo;
        // the words the fused path has to walk the tries for are filed in a store of the handle's own (tables.hpp "piece store")
        if (int rc = alloc_piece_store(h->store, h->store_room, p->vocab.n, p->vocab.n <= 65535, h->dev.store, h->store_capacity, resolve_memo_store(p->memo_store))) return rc;
    }
    *out = h.release();
    return OVTK_OK;
}

int ovtk_wordpiece_run(ovtk_wordpiece* h, const ovtk_ragged_strings* in, int32_t unk_token_id, ovtk_ragged_i32_out* out,
                       int mem, void* stream) {
    if (int rc = check_rows(in)) return rc;
    if (!h || !out) return set_error(OVTK_E_ARG, "null argument");
    if (out->data_capacity < 0 || out->data_capacity >= INT32_MAX) return set_error(OVTK_E_ARG, "bad output capacity");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    out->n_data = 0;
    out->n_rows = in->n_rows;
    if (in->n_rows == 0) return OVTK_OK;
    return run_rows_to_ids(h->device, "WordpieceTokenizer", in, nullptr, 1, out, mem, s,
                           [&](Workspace& ws, const RowsIn& d_in, const EncodeWork& w, int grid) {
                               OVTK_LAUNCH(ws.marks, "wordpiece", wordpiece_kernel, grid, kBlockThreads, s, d_in, h->dev,
                                           unk_token_id, w);
                           });
}

}  // extern "C"
namespace {
// Launches the fused BERT split + WordPiece kernels; `run` stays empty when the result was complete without any.
int start_wordpiece_encode(ovtk_wordpiece* h, ovtk_regex_split* whitespace, ovtk_regex_split* delimiters,
                           const ovtk_ragged_strings* in, int32_t unk_token_id, ovtk_ragged_i32_out* out, int mem,
                           void* stream, std::unique_ptr<PendingRun>& run) {
    if (int rc = check_rows(in)) return rc;
    if (!h || !whitespace || !delimiters || !out) return set_error(OVTK_E_ARG, "null argument");
    if (whitespace->dev.kind != kSplitWhitespace || whitespace->dev.drop != 1 || whitespace->max_splits != -1 ||
        delimiters->dev.kind != kSplitBertPunct || delimiters->dev.drop != 0 || delimiters->max_splits != -1)
        return set_error(OVTK_E_UNSUPPORTED,
                         "fused WordPiece encode needs RegexSplit(\\s+, remove) followed by RegexSplit(BERT delimiters, isolate)");
    if (whitespace->device != h->device || delimiters->device != h->device)
        return set_error(OVTK_E_ARG, "split and wordpiece handles live on different devices");
    if (out->data_capacity < 0 || out->data_capacity >= INT32_MAX) return set_error(OVTK_E_ARG, "bad output capacity");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    out->n_data = 0;
    out->n_rows = in->n_rows;
    if (in->strings.n_chars == 0) {  // regex_split.cpp:129-143: the first split leaves one empty row
        const int32_t zero = 0;
        if (mem == OVTK_MEM_HOST) {
            out->begins[0] = 0;
            out->ends[0] = 0;
        } else {
            OVTK_HIP(hipMemcpyAsync(out->begins, &zero, 4, hipMemcpyHostToDevice, s));
            OVTK_HIP(hipMemcpyAsync(out->ends, &zero, 4, hipMemcpyHostToDevice, s));
            OVTK_HIP(hipStreamSynchronize(s));
        }
        out->n_rows = 1;
        return OVTK_OK;
    }
    if (in->n_rows == 0) return OVTK_OK;
    SplitDev sp = whitespace->dev;
    sp.kind = kSplitBertWords;  // whitespace runs dropped, every delimiter char its own word
    sp.drop = 1;
    BpeDev memo_only{};        // the lookup kernel reads nothing but the memo and the (absent) end suffix
    memo_only.pieces = h->memo;
    memo_only.store = h->dev.store;   // (the short path: lookup_span_kernel looks the words the memo does not hold up in the word store itself)
    memo_only.suffix_len = 0;
    memo_only.unk_id = unk_token_id;   // (what a word-store entry without a segmentation comes back as)
    const int dev = h->device;
    const WordpieceDev wdev = h->dev;
    auto r = make_rows_run(dev, "WordpieceTokenizer", in, nullptr, 1, out, mem, s,
                           [=](Workspace& ws, const RowsIn& d_in, const EncodeWork& w, int grid) {
                               // rows that are one ASCII window: lookup_rows_kernel (consecutive rows per wave, the next row's text
                                   // requested ahead: encode_kernels.hpp); what it leaves -- marked in row_used -- goes through the
                                   // generic kernel
                               EncodeWork w1 = w;
                               const int grid1 = rows_grid(d_in.n_rows, grid, w1.rows_per_wave);
                               if (!w.rows_per_ticket) {
                                   // several rows per scan block (span_kernel.hpp) where there is a word memo to probe
                                   if (!(w.launch_mask & kLaunchSpan)) {}   // (the short path's second set of launches: the span kernel has run)
                                   else if (memo_only.pieces.slots && w1.stage16)
                                       OVTK_LAUNCH(ws.marks, "lookup_words", (lookup_span_kernel<kSpanBertWords, true>), grid1, kBlockThreads, s, d_in, sp,
                                                   memo_only, w1);
                                   else if (memo_only.pieces.slots)
                                       OVTK_LAUNCH(ws.marks, "lookup_words", (lookup_span_kernel<kSpanBertWords, false>), grid1, kBlockThreads, s, d_in, sp,
                                                   memo_only, w1);
                                   else
                                       OVTK_LAUNCH(ws.marks, "lookup_words", lookup_rows_kernel<kRowsBertWords>, grid1, kBlockThreads, s, d_in, sp,
                                                   memo_only, w1);
                                   EncodeWork w2 = w;
                                   w2.only_pending = 1;
                                   if (w.launch_mask & kLaunchPending)
                                       OVTK_LAUNCH(ws.marks, "lookup_fused", lookup_kernel<kFused>, grid, kBlockThreads, s, d_in, sp, memo_only, w2);
                               } else {
                                   OVTK_LAUNCH(ws.marks, "lookup_words", lookup_kernel<kFused>, grid, kBlockThreads, s, d_in, sp, memo_only, w);
                               }
                               if (!(w.launch_mask & kLaunchMerge)) return;   // (the short path: no word is expected to be left for the tries)
                               const dim3 dgrid(grid_deferred_hinted(grid_deferred_per_shard(d_in.n_chars, d_in.n_strings, device_cu_count(dev) * 8 / kShards), w.span_sums, w.merge_hint), kShards);
                               if (w.stage16)
                                   OVTK_LAUNCH(ws.marks, "wordpiece_deferred", wordpiece_deferred_kernel<true>, dgrid, kBlockThreads, s, d_in, wdev,
                                               unk_token_id, w, w.fold_tail ? d_in.n_rows : 0, w.out_cap);
                               else
                                   OVTK_LAUNCH(ws.marks, "wordpiece_deferred", wordpiece_deferred_kernel<false>, dgrid, kBlockThreads, s, d_in, wdev,
                                               unk_token_id, w, w.fold_tail ? d_in.n_rows : 0, w.out_cap);
                           },
                           /*self_alloc=*/true,
                           memo_only.pieces.slots ? resident_blocks_per_cu(lookup_span_kernel<kSpanBertWords, true>) : resident_blocks_per_cu(lookup_kernel<kFused>),
                           /*tail_in_middle=*/true);
    if (h->n_vocab > 0 && h->n_vocab <= 65535 && unk_token_id >= 0 && unk_token_id <= 65534) r->enable_stage16();
    if (memo_only.pieces.slots) r->stage_twice();   // (lookup_span_kernel in front of the generic kernel)
    // The short path (span_kernel.hpp): lookup_kernel<kFused> / wordpiece_deferred_kernel only when the handle's last calls had rows /
    // words for them (as a BPE handle does: api_encode.cpp).
    if (memo_only.pieces.slots && !row_tickets().load(std::memory_order_relaxed) && short_path_mode().load(std::memory_order_relaxed) != 0) {
        r->enable_short_path(h->expect_pending.load(std::memory_order_relaxed) > 0, h->expect_merge.load(std::memory_order_relaxed) > 0 || !memo_only.store.slots,
                             memo_only.store.slots ? h->last_unresolved.load(std::memory_order_relaxed) : -1);
        r->on_status([h](const RunStatus& st) {
            if (!st.short_path) return;
            auto note = [](std::atomic<int>& expect, bool had_work) {
                if (had_work) expect.store(kShortPathKeep, std::memory_order_relaxed);
                else if (expect.load(std::memory_order_relaxed) > 0) expect.fetch_sub(1, std::memory_order_relaxed);
            };
            note(h->expect_pending, st.n_pending > 0);
            if (st.short_path != 3) {   // (3: the long way, nothing was counted)
                note(h->expect_merge, st.n_unresolved > 0);
                h->last_unresolved.store(st.n_unresolved, std::memory_order_relaxed);
            }
        });
    }
    if (int rc = r->start()) return rc;
    run = std::move(r);
    return OVTK_OK;
}
}  // namespace
extern "C" {

int ovtk_wordpiece_encode_run(ovtk_wordpiece* h, ovtk_regex_split* whitespace, ovtk_regex_split* delimiters,
                              const ovtk_ragged_strings* in, int32_t unk_token_id, ovtk_ragged_i32_out* out, int mem,
                              void* stream) {
    std::unique_ptr<PendingRun> run;
    if (int rc = start_wordpiece_encode(h, whitespace, delimiters, in, unk_token_id, out, mem, stream, run)) return rc;
    return run ? run->finish(out) : OVTK_OK;
}

int ovtk_wordpiece_encode_enqueue(ovtk_wordpiece* h, ovtk_regex_split* whitespace, ovtk_regex_split* delimiters,
                                  const ovtk_ragged_strings* in, int32_t unk_token_id, const ovtk_ragged_i32_out* out,
                                  void* stream, ovtk_pending** pending) {
    if (!pending || !out) return set_error(OVTK_E_ARG, "null argument");
    auto p = std::make_unique<ovtk_pending>();
    p->out = *out;
    if (int rc = start_wordpiece_encode(h, whitespace, delimiters, in, unk_token_id, &p->out, OVTK_MEM_DEVICE, stream, p->run)) return rc;
    *pending = p.release();
    return OVTK_OK;
}

void ovtk_wordpiece_destroy(ovtk_wordpiece* h) { delete h; }

// ------------------------------------------------------------------------------- VocabEncoder
int ovtk_vocab_encoder_create(const ovtk_vocab_encoder_params* p, ovtk_vocab_encoder** out) {
    if (!p || !out) return set_error(OVTK_E_ARG, "vocab_encoder: null argument");
    if (int rc = check_strings_arg(&p->keys, "vocab_encoder keys")) return rc;
    if (p->value_size != 4 && p->value_size != 8)  // vocab_encoder.cpp:38-53
        return set_error(OVTK_E_ARG, "VocabEncoder: unsupported element type (i32 and i64 values only)");
    if (p->keys.n > 0 && !p->values) return set_error(OVTK_E_ARG, "vocab_encoder: null values");
    if (int rc = use_device(p->device)) return rc;
    auto h = std::make_unique<ovtk_vocab_encoder>();
    h->device = p->device;
    h->value_size = p->value_size;
    StringMapHost host;
    std::string err;
    if (int rc = build_string_map(view_of(p->keys), host, err)) return set_error(rc, err);
    int e = 0;
    e = e ? e : h->slots.upload(host.slots.data(), host.slots.size() * sizeof(uint64_t));
    e = e ? e : h->kb.upload(host.key_begins.data(), host.key_begins.size() * 4);
    e = e ? e : h->ke.upload(host.key_ends.data(), host.key_ends.size() * 4);
    e = e ? e : h->kc.upload(host.key_chars.data(), host.key_chars.size());
    e = e ? e : h->values.upload(p->values, size_t(p->keys.n) * size_t(p->value_size));
    if (e) return e;
    OVTK_HIP(hipStreamSynchronize(nullptr));
    h->dev.slots = h->slots.as<uint64_t>();
    h->dev.mask = host.mask;
    h->dev.key_begins = h->kb.as<int32_t>();
    h->dev.key_ends = h->ke.as<int32_t>();
    h->dev.key_chars = h->kc.as<uint8_t>();
    h->n_key_chars = (long long)host.key_chars.size();
    h->dev.values = h->values.as<void>();
    h->dev.value_size = p->value_size;
    *out = h.release();
    return OVTK_OK;
}

int ovtk_vocab_encoder_run(ovtk_vocab_encoder* h, const ovtk_strings* in, const void* default_value, void* out, int mem,
                           void* stream) {
    if (!h || !default_value) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_strings_arg(in, "vocab_encoder input")) return rc;
    if (in->n == 0) return OVTK_OK;
    if (!out) return set_error(OVTK_E_ARG, "null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    WorkspaceLease ws(h->device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    const int32_t *b = nullptr, *e = nullptr;
    const uint8_t* c = nullptr;
    if (int rc = in_source(ws->in_begins, in->begins, size_t(in->n) * 4, mem, s, &b)) return rc;
    if (int rc = in_source(ws->in_ends, in->ends, size_t(in->n) * 4, mem, s, &e)) return rc;
    if (int rc = in_source(ws->in_chars, in->chars, size_t(in->n_chars), mem, s, &c)) return rc;
    const size_t out_bytes = size_t(in->n) * size_t(h->value_size);
    uint8_t* d_out = nullptr;
    if (int rc = out_target(ws->out_a, static_cast<uint8_t*>(out), out_bytes, mem, &d_out)) return rc;
    const int n = int(in->n);
    if (h->value_size == 4) {
        int32_t d;
        std::memcpy(&d, default_value, 4);
        OVTK_LAUNCH(ws->marks, "vocab_encoder", vocab_encoder_kernel<int32_t>, grid_for_elems(n), kBlockThreads, s, b, e, c,
                    (long long)in->n_chars, n, h->dev, h->n_key_chars, d, reinterpret_cast<int32_t*>(d_out), st);
    } else {
        long long d;
        std::memcpy(&d, default_value, 8);
        OVTK_LAUNCH(ws->marks, "vocab_encoder", vocab_encoder_kernel<long long>, grid_for_elems(n), kBlockThreads, s, b, e, c,
                    (long long)in->n_chars, n, h->dev, h->n_key_chars, d, reinterpret_cast<long long*>(d_out), st);
    }
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside the chars tensor");
    if (int rc = copy_back(out, d_out, out_bytes, mem, s)) return rc;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

void ovtk_vocab_encoder_destroy(ovtk_vocab_encoder* h) { delete h; }

// ------------------------------------------------------------------------------- RaggedToDense
int ovtk_ragged_to_dense(const int32_t* begins, const int32_t* ends, int64_t n_rows, const void* data, int64_t n_data,
                         int elem_size, int64_t inner_elems, int32_t target_dim, const void* default_value, int pad_right,
                         int pad_max_length, void* out_dense, uint8_t* out_mask, int mem, int device, void* stream) {
    if (n_rows < 0 || n_data < 0 || inner_elems < 1 || target_dim < 0) return set_error(OVTK_E_ARG, "ragged_to_dense: bad size");
    if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8)
        return set_error(OVTK_E_ARG, "ragged_to_dense: element size must be 1, 2, 4 or 8 bytes (POD types)");
    if (!default_value) return set_error(OVTK_E_ARG, "ragged_to_dense: null default value");
    if (n_rows >= INT32_MAX || inner_elems * elem_size >= (1 << 20))
        return set_error(OVTK_E_UNSUPPORTED, "ragged_to_dense: shape too large for one call");
    if (int rc = use_device(device)) return rc;
    if (n_rows == 0 || target_dim == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    DenseArgs a{};
    a.n_rows = int32_t(n_rows);
    a.n_data = n_data;
    a.elem_size = elem_size;
    a.inner = int32_t(inner_elems);
    a.cell = int32_t(elem_size * inner_elems);
    a.target = target_dim;
    a.pad_right = pad_right != 0;
    a.pad_max_length = pad_max_length != 0;
    std::memset(a.dflt, 0, sizeof a.dflt);
    std::memcpy(a.dflt, default_value, size_t(elem_size));
    a.status = st;
    const uint8_t* d_data = nullptr;
    if (int rc = in_source(ws->in_begins, begins, size_t(n_rows) * 4, mem, s, &a.begins)) return rc;
    if (int rc = in_source(ws->in_ends, ends, size_t(n_rows) * 4, mem, s, &a.ends)) return rc;
    if (int rc = in_source(ws->in_chars, static_cast<const uint8_t*>(data), size_t(n_data) * size_t(a.cell), mem, s, &d_data)) return rc;
    a.data = d_data;
    const size_t dense_bytes = size_t(n_rows) * size_t(target_dim) * size_t(a.cell);
    const size_t mask_bytes = size_t(n_rows) * size_t(target_dim) * size_t(inner_elems);
    if (int rc = out_target(ws->out_a, static_cast<uint8_t*>(out_dense), dense_bytes, mem, &a.out)) return rc;
    a.mask = nullptr;
    if (out_mask)
        if (int rc = out_target(ws->out_b, out_mask, mask_bytes, mem, &a.mask)) return rc;
    const unsigned long long total = (unsigned long long)n_rows * (unsigned long long)target_dim;
    // (dword cells only when ONE 4-byte element is the cell: 2 x 2-byte or 4 x 1-byte cells have a mask byte per inner element
    // and a default that repeats per element -- the generic path)
    const bool flat = a.elem_size == 4 && a.inner == 1 && total < (1ull << 31) && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                      (!a.mask || (reinterpret_cast<uintptr_t>(a.mask) & 3) == 0) && (reinterpret_cast<uintptr_t>(a.data) & 3) == 0;
    if (flat)
        OVTK_LAUNCH(ws->marks, "ragged_to_dense", ragged_to_dense_flat4_kernel,
                    int(std::min<unsigned long long>((total / 4 + kBlockThreads) / kBlockThreads, 1ull << 22)),  // one group of four per thread
                    kBlockThreads, s, a);
    else
        OVTK_LAUNCH(ws->marks, "ragged_to_dense", ragged_to_dense_kernel, grid_for_rows(a.n_rows), kBlockThreads, s, a);
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "ragged_to_dense: a row reads past the data tensor");
    if (int rc = copy_back(out_dense, a.out, dense_bytes, mem, s)) return rc;
    if (out_mask)
        if (int rc = copy_back(out_mask, a.mask, mask_bytes, mem, s)) return rc;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- VocabDecoder
int ovtk_vocab_decoder_create(const ovtk_vocab_decoder_params* p, ovtk_vocab_decoder** out) {
    if (!p || !out) return set_error(OVTK_E_ARG, "vocab_decoder: null argument");
    if (int rc = check_strings_arg(&p->vocab, "vocab_decoder vocab")) return rc;
    if (p->n_skip_tokens < 0 || (p->n_skip_tokens > 0 && !p->skip_tokens)) return set_error(OVTK_E_ARG, "vocab_decoder: bad skip_tokens");
    if (int rc = use_device(p->device)) return rc;
    auto h = std::make_unique<ovtk_vocab_decoder>();
    h->device = p->device;
    const int64_t V = p->vocab.n;
    h->vocab_size = int32_t(V);
    if (V > 0 && (!p->vocab.begins || !p->vocab.ends)) return set_error(OVTK_E_ARG, "vocab_decoder: null vocab offsets");
    const size_t nv = (size_t(std::max<int64_t>(V, 1)) + 1) & ~size_t(1);  // even: the length tables are also read as dwords
    std::vector<uint16_t> len_plain(nv, 0), len_bf(nv, 0);
    std::vector<TokenPack> pack_plain(nv, TokenPack{{0, 0, 0, 0}}), pack_bf(nv, TokenPack{{0, 0, 0, 0}});
    for (int64_t i = 0; i < V; ++i) {
        const int64_t b = p->vocab.begins[i], e = p->vocab.ends[i];
        if (b < 0 || e < b || e > p->vocab.n_chars) return set_error(OVTK_E_RANGE, "vocab_decoder: vocab begins/ends outside chars");
        if (e - b > 0xFFFF) return set_error(OVTK_E_UNSUPPORTED, "vocab_decoder: tokens longer than 65535 bytes are not supported");
        h->max_token_len = std::max<int32_t>(h->max_token_len, int32_t(e - b));
        const uint8_t* t = p->vocab.chars + b;
        const int len = int(e - b);
        len_plain[size_t(i)] = uint16_t(len);
        std::memcpy(pack_plain[size_t(i)].w, t, size_t(std::min(len, 15)));
        pack_plain[size_t(i)].w[3] |= uint32_t(len <= 15 ? len : 0xFF) << 24;  // byte 15: the length
        // ByteFallback applied to this token (byte_fallback.cpp:37-41), precomputed once per vocabulary
        int v = -1;
        if (len == 6 && t[0] == '<' && t[5] == '>') {
            bool only_first = true;
            for (int k = 1; k < 6; ++k) only_first = only_first && t[k] != '<';
            if (only_first) {
                v = 255;  // not a "<0x%02X>" spelling: PieceToByte returns -1, stored as (uint8_t)0xFF
                auto hex = [](uint8_t c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1); };
                if (t[1] == '0' && t[2] == 'x' && hex(t[3]) >= 0 && hex(t[4]) >= 0) v = hex(t[3]) * 16 + hex(t[4]);
            }
        }
        if (v >= 0) {
            len_bf[size_t(i)] = 1;
            pack_bf[size_t(i)].w[0] = uint32_t(v);
            pack_bf[size_t(i)].w[3] = 1u << 24;
        } else {
            len_bf[size_t(i)] = len_plain[size_t(i)];
            pack_bf[size_t(i)] = pack_plain[size_t(i)];
        }
    }
    h->len_plain_host = len_plain;
    h->len_bf_host = len_bf;
    for (int64_t k = 0; k < p->n_skip_tokens; ++k) {  // a skipped token decodes to "" (vocab_decoder.cpp:70-81): length 0 in the tables
        const int32_t t = p->skip_tokens[k];
        if (t >= 0 && t < V) {
            len_plain[size_t(t)] = len_bf[size_t(t)] = 0;
            pack_plain[size_t(t)].w[3] &= 0x00FFFFFFu;  // length byte 0; the text stays (a call's own skip list may keep the token)
            pack_bf[size_t(t)].w[3] &= 0x00FFFFFFu;
        }
    }
    int e = 0;
    e = e ? e : h->vb.upload(p->vocab.begins, size_t(V) * 4);
    e = e ? e : h->vc.upload(p->vocab.chars, size_t(p->vocab.n_chars));
    e = e ? e : h->len_plain.upload(len_plain.data(), nv * sizeof(uint16_t));
    e = e ? e : h->pack_plain.upload(pack_plain.data(), nv * sizeof(TokenPack));
    e = e ? e : h->len_bf.upload(len_bf.data(), nv * sizeof(uint16_t));
    e = e ? e : h->pack_bf.upload(pack_bf.data(), nv * sizeof(TokenPack));
    if (e) return e;
    OVTK_HIP(hipStreamSynchronize(nullptr));
    *out = h.release();
    return OVTK_OK;
}

void ovtk_vocab_decoder_destroy(ovtk_vocab_decoder* h) { delete h; }

}  // extern "C"

namespace {

// Common front of VocabDecoder and the fused detokenizer: ids on the device + the skip bitmap of this call.
int decoder_inputs(ovtk_vocab_decoder* h, Workspace& ws, const int32_t* ids, int64_t n_ids, const int32_t* skip_in,
                   int64_t n_skip_in, int mem, hipStream_t s, bool byte_fallback, DecodeDev& d) {
    d = DecodeDev{};
    if (int rc = in_source(ws.gen[0], ids, size_t(n_ids) * 4, mem, s, &d.ids)) return rc;
    d.v_begins = h->vb.as<int32_t>();
    d.v_chars = h->vc.as<uint8_t>();
    d.v_len = (byte_fallback ? h->len_bf : h->len_plain).as<uint16_t>();
    d.v_pack = (byte_fallback ? h->pack_bf : h->pack_plain).as<TokenPack>();
    d.vocab_size = h->vocab_size;
    d.len_in_pack = skip_in ? 0 : 1;
    if (skip_in) {  // input 4 replaces the attribute for this call (vocab_decoder.cpp:36-41): its own length table
        std::vector<uint16_t> lens = byte_fallback ? h->len_bf_host : h->len_plain_host;
        for (int64_t k = 0; k < n_skip_in; ++k) {
            const int32_t t = skip_in[k];
            if (t >= 0 && t < h->vocab_size) lens[size_t(t)] = 0;
        }
        if (int rc = ws.gen[1].upload(lens.data(), lens.size() * sizeof(uint16_t), s)) return rc;
        OVTK_HIP(hipStreamSynchronize(s));  // `lens` dies at return
        d.v_len = ws.gen[1].as<uint16_t>();
    }
    return OVTK_OK;
}

int check_decoder_args(ovtk_vocab_decoder* h, const int32_t* ids, int64_t batch, int64_t seq_len, const int32_t* skip_in,
                       int64_t n_skip_in, ovtk_strings_out* out) {
    if (!h || !out) return set_error(OVTK_E_ARG, "null argument");
    if (batch < 0 || seq_len < 0 || n_skip_in < 0) return set_error(OVTK_E_ARG, "negative size");
    if (batch * std::max<int64_t>(seq_len, 1) >= INT32_MAX)
        return set_error(OVTK_E_ARG, "batch * seq_len must fit int32 offsets (vocab_decoder.cpp:45-46); split the call");
    if (batch * seq_len > 0 && !ids) return set_error(OVTK_E_ARG, "null ids");
    if (n_skip_in > 0 && !skip_in) return set_error(OVTK_E_ARG, "null skip tokens");
    if (out->chars_capacity < 0) return set_error(OVTK_E_ARG, "bad output capacity");
    return OVTK_OK;
}

}  // namespace

extern "C" {

int ovtk_vocab_decoder_run(ovtk_vocab_decoder* h, const int32_t* ids, int64_t batch, int64_t seq_len,
                           const int32_t* skip_in, int64_t n_skip_in, int32_t* out_rb, int32_t* out_re,
                           ovtk_strings_out* out, int mem, void* stream) {
    if (int rc = check_decoder_args(h, ids, batch, seq_len, skip_in, n_skip_in, out)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    out->n_chars = 0;
    if (batch == 0) return OVTK_OK;
    const int64_t sp = seq_len > 0 ? seq_len : 1, n_tok = batch * sp;
    WorkspaceLease ws(h->device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    int32_t *d_rb = nullptr, *d_re = nullptr, *d_b = nullptr, *d_e = nullptr;
    uint8_t* d_c = nullptr;
    if (int rc = out_target(ws->out_a, out_rb, size_t(batch) * 4, mem, &d_rb)) return rc;
    if (int rc = out_target(ws->out_b, out_re, size_t(batch) * 4, mem, &d_re)) return rc;
    if (int rc = out_target(ws->out_c, out->begins, size_t(n_tok) * 4, mem, &d_b)) return rc;
    if (int rc = out_target(ws->out_d, out->ends, size_t(n_tok) * 4, mem, &d_e)) return rc;
    if (int rc = out_target(ws->out_e, out->chars, size_t(out->chars_capacity), mem, &d_c)) return rc;
    OVTK_LAUNCH(ws->marks, "decoder_rows", decoder_rows_kernel, grid_for_elems(batch), kBlockThreads, s, int(batch), int(sp), d_rb, d_re);
    if (seq_len == 0) {  // vocab_decoder.cpp:61-65: one empty string per row
        OVTK_HIP(hipMemsetAsync(d_b, 0, size_t(n_tok) * 4, s));
        OVTK_HIP(hipMemsetAsync(d_e, 0, size_t(n_tok) * 4, s));
    } else {
        DecodeDev d;
        if (int rc = decoder_inputs(h, *ws.ws, ids, n_tok, skip_in, n_skip_in, mem, s, false, d)) return rc;
        if (int rc = decode_passes(*ws.ws, s, h->device, d, batch, seq_len, d_b, d_e, nullptr, nullptr, d_c,
                                   (long long)std::min<int64_t>(out->chars_capacity, INT32_MAX - 1), st, "vocab_decoder"))
            return rc;
    }
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagOutCapacity)
        return set_error(OVTK_E_CAPACITY, "VocabDecoder: output chars buffer too small or beyond int32 offsets (" +
                                              std::to_string(ws->host_status->n_out) + " bytes needed)");
    out->n_chars = ws->host_status->n_out;
    int e = 0;
    e = e ? e : copy_back(out_rb, d_rb, size_t(batch) * 4, mem, s);
    e = e ? e : copy_back(out_re, d_re, size_t(batch) * 4, mem, s);
    e = e ? e : copy_back(out->begins, d_b, size_t(n_tok) * 4, mem, s);
    e = e ? e : copy_back(out->ends, d_e, size_t(n_tok) * 4, mem, s);
    e = e ? e : copy_back(out->chars, d_c, size_t(out->n_chars), mem, s);
    if (e) return e;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

int ovtk_detokenize_run(ovtk_vocab_decoder* h, const int32_t* ids, int64_t batch, int64_t seq_len, const int32_t* skip_in,
                        int64_t n_skip_in, int byte_fallback, ovtk_strings_out* out, int mem, void* stream) {
    if (int rc = check_decoder_args(h, ids, batch, seq_len, skip_in, n_skip_in, out)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    out->n_chars = 0;
    if (batch == 0) return OVTK_OK;
    WorkspaceLease ws(h->device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    int32_t *d_b = nullptr, *d_e = nullptr;
    uint8_t* d_c = nullptr;
    if (int rc = out_target(ws->out_c, out->begins, size_t(batch) * 4, mem, &d_b)) return rc;
    if (int rc = out_target(ws->out_d, out->ends, size_t(batch) * 4, mem, &d_e)) return rc;
    if (int rc = out_target(ws->out_e, out->chars, size_t(out->chars_capacity), mem, &d_c)) return rc;
    if (seq_len == 0) {
        OVTK_HIP(hipMemsetAsync(d_b, 0, size_t(batch) * 4, s));
        OVTK_HIP(hipMemsetAsync(d_e, 0, size_t(batch) * 4, s));
    } else {
        DecodeDev d;
        if (int rc = decoder_inputs(h, *ws.ws, ids, batch * seq_len, skip_in, n_skip_in, mem, s, byte_fallback != 0, d)) return rc;
        // per-token offsets are not materialised: the scan writes the row bounds FuzeRagged would pick
        if (int rc = decode_passes(*ws.ws, s, h->device, d, batch, seq_len, nullptr, nullptr, d_b, d_e, d_c,
                                   (long long)std::min<int64_t>(out->chars_capacity, INT32_MAX - 1), st, "detokenize"))
            return rc;
    }
    if (int rc = finish_status(*ws.ws, s)) return rc;
    out->n_chars = ws->host_status->n_out;  // (on E_CAPACITY: what the call needs, INT32_MAX = more than int32 offsets reach)
    if (ws->host_status->flags & kFlagOutCapacity)
        return set_error(OVTK_E_CAPACITY, "detokenize: output chars buffer too small or beyond int32 offsets (" +
                                              std::to_string(ws->host_status->n_out) + " bytes needed)");
    int e = 0;
    e = e ? e : copy_back(out->begins, d_b, size_t(batch) * 4, mem, s);
    e = e ? e : copy_back(out->ends, d_e, size_t(batch) * 4, mem, s);
    e = e ? e : copy_back(out->chars, d_c, size_t(out->n_chars), mem, s);
    if (e) return e;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

// The same call in two halves for device buffers: enqueue launches the three passes and a copy of the status block,
// finish waits for that call's own event and reports.  Every call in flight leases its own workspace.
namespace {
struct DetokRun final : ovtk::PendingStrings {
    explicit DetokRun(int device) : ws(device) {}
    WorkspaceLease ws;
    int finish(ovtk_strings_out* out) override {
        OVTK_HIP(hipEventSynchronize(ws->done));
        ws->marks.settled();
        Profiler::get().resolve(ws->marks);
        OVTK_HIP(hipGetLastError());
        out->n_chars = ws->host_status->n_out;  // (on E_CAPACITY: what the call needs, INT32_MAX = more than int32 offsets reach)
        if (ws->host_status->flags & kFlagOutCapacity)
            return set_error(OVTK_E_CAPACITY, "detokenize: output chars buffer too small or beyond int32 offsets (" +
                                                  std::to_string(ws->host_status->n_out) + " bytes needed)");
        return OVTK_OK;
    }
};
}  // namespace

int ovtk_detokenize_enqueue(ovtk_vocab_decoder* h, const int32_t* ids, int64_t batch, int64_t seq_len, const int32_t* skip_in,
                            int64_t n_skip_in, int byte_fallback, ovtk_strings_out* out, void* stream, ovtk_pending** pending) {
    if (!pending) return set_error(OVTK_E_ARG, "null argument");
    *pending = nullptr;
    if (int rc = check_decoder_args(h, ids, batch, seq_len, skip_in, n_skip_in, out)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    auto p = std::make_unique<ovtk_pending>();
    p->strings_out = *out;
    p->strings_out.n_chars = 0;
    if (batch > 0) {
        auto run = std::make_unique<DetokRun>(h->device);
        Workspace& ws = *run->ws.ws;
        if (!ws.done) OVTK_HIP(hipEventCreateWithFlags(&ws.done, hipEventDisableTiming));
        RunStatus* st = nullptr;
        if (int rc = begin_status(ws, s, &st)) return rc;
        if (seq_len == 0) {
            OVTK_HIP(hipMemsetAsync(out->begins, 0, size_t(batch) * 4, s));
            OVTK_HIP(hipMemsetAsync(out->ends, 0, size_t(batch) * 4, s));
        } else {
            DecodeDev d;
            if (int rc = decoder_inputs(h, ws, ids, batch * seq_len, skip_in, n_skip_in, OVTK_MEM_DEVICE, s, byte_fallback != 0, d)) return rc;
            if (int rc = decode_passes(ws, s, h->device, d, batch, seq_len, nullptr, nullptr, out->begins, out->ends, out->chars,
                                       (long long)std::min<int64_t>(out->chars_capacity, INT32_MAX - 1), st, "detokenize"))
                return rc;
        }
        OVTK_HIP(hipMemcpyAsync(ws.host_status, ws.status.as<RunStatus>(), sizeof(RunStatus), hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipEventRecord(ws.done, s));
        p->strings = std::move(run);
    }
    *pending = p.release();
    return OVTK_OK;
}

int ovtk_detokenize_finish(ovtk_pending* pending, ovtk_strings_out* out) {
    if (!pending) return set_error(OVTK_E_ARG, "null argument");
    std::unique_ptr<ovtk_pending> p(pending);  // released whatever happens
    const int rc = p->strings ? p->strings->finish(&p->strings_out) : OVTK_OK;
    if (out) *out = p->strings_out;
    return rc;
}

// ------------------------------------------------------------------------------- ByteFallback
int ovtk_byte_fallback(const ovtk_strings* in, ovtk_strings_out* out, int mem, int device, void* stream) {
    if (int rc = check_strings_arg(in, "byte_fallback input")) return rc;
    if (!out || out->chars_capacity < 0) return set_error(OVTK_E_ARG, "byte_fallback: bad output");
    if (int rc = use_device(device)) return rc;
    out->n_chars = 0;
    if (in->n == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    const int32_t *b = nullptr, *e = nullptr;
    const uint8_t* c = nullptr;
    if (int rc = in_source(ws->in_begins, in->begins, size_t(in->n) * 4, mem, s, &b)) return rc;
    if (int rc = in_source(ws->in_ends, in->ends, size_t(in->n) * 4, mem, s, &e)) return rc;
    if (int rc = in_source(ws->in_chars, in->chars, size_t(in->n_chars), mem, s, &c)) return rc;
    int32_t *d_b = nullptr, *d_e = nullptr;
    uint8_t* d_c = nullptr;
    if (int rc = out_target(ws->out_c, out->begins, size_t(in->n) * 4, mem, &d_b)) return rc;
    if (int rc = out_target(ws->out_d, out->ends, size_t(in->n) * 4, mem, &d_e)) return rc;
    if (int rc = out_target(ws->out_e, out->chars, size_t(out->chars_capacity), mem, &d_c)) return rc;
    OVTK_LAUNCH(ws->marks, "check_strings", check_strings_kernel, grid_for_elems(in->n), kBlockThreads, s, b, e,
                (long long)in->n, (long long)in->n_chars, st);
    if (int rc = scan_and_apply(*ws.ws, s, in->n, FallbackLen{b, e, c, (long long)in->n_chars}, FallbackApply{b, e, c, d_b, d_e, d_c},
                                (long long)out->chars_capacity, st, "byte_fallback"))
        return rc;
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside the chars tensor");
    if (ws->host_status->flags & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "ByteFallback: output chars buffer too small");
    out->n_chars = ws->host_status->n_out;
    int err = 0;
    err = err ? err : copy_back(out->begins, d_b, size_t(in->n) * 4, mem, s);
    err = err ? err : copy_back(out->ends, d_e, size_t(in->n) * 4, mem, s);
    err = err ? err : copy_back(out->chars, d_c, size_t(out->n_chars), mem, s);
    if (err) return err;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- FuzeRagged
int ovtk_fuze_ragged(const int32_t* ragged_begins, const int32_t* ragged_ends, int64_t n_rows, const int32_t* begins,
                     const int32_t* ends, int64_t n, int32_t* out_begins, int32_t* out_ends, int mem, int device, void* stream) {
    if (n_rows < 0 || n < 0 || n_rows >= INT32_MAX || n >= INT32_MAX) return set_error(OVTK_E_ARG, "fuze: bad size");
    if (int rc = use_device(device)) return rc;
    if (n_rows == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    const int32_t *rb = nullptr, *re = nullptr, *b = nullptr, *e = nullptr;
    if (int rc = in_source(ws->in_rb, ragged_begins, size_t(n_rows) * 4, mem, s, &rb)) return rc;
    if (int rc = in_source(ws->in_re, ragged_ends, size_t(n_rows) * 4, mem, s, &re)) return rc;
    if (int rc = in_source(ws->in_begins, begins, size_t(n) * 4, mem, s, &b)) return rc;
    if (int rc = in_source(ws->in_ends, ends, size_t(n) * 4, mem, s, &e)) return rc;
    int32_t *d_b = nullptr, *d_e = nullptr;
    if (int rc = out_target(ws->out_a, out_begins, size_t(n_rows) * 4, mem, &d_b)) return rc;
    if (int rc = out_target(ws->out_b, out_ends, size_t(n_rows) * 4, mem, &d_e)) return rc;
    OVTK_LAUNCH(ws->marks, "fuze", fuze_kernel, grid_for_elems(n_rows), kBlockThreads, s, rb, re, int(n_rows), b, e, int(n), d_b, d_e, st);
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "fuze: ragged begins/ends index outside the string tensor");
    int err = 0;
    err = err ? err : copy_back(out_begins, d_b, size_t(n_rows) * 4, mem, s);
    err = err ? err : copy_back(out_ends, d_e, size_t(n_rows) * 4, mem, s);
    if (err) return err;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- TrieTokenizer
struct ovtk_trie_tokenizer {
    int device = 0;
    TrieBucketsDev dev{};
    DevBuf root, buckets;
};

int ovtk_trie_tokenizer_create(const ovtk_strings* vocab, const int32_t* indices, int device, ovtk_trie_tokenizer** out) {
    if (int rc = check_strings_arg(vocab, "trie tokenizer vocab")) return rc;
    if (!indices || !out) return set_error(OVTK_E_ARG, "trie tokenizer: null argument");
    if (int rc = use_device(device)) return rc;
    auto h = std::make_unique<ovtk_trie_tokenizer>();
    h->device = device;
    TrieHost t;
    for (int64_t i = 0; i < vocab->n; ++i) {  // trie_tokenizer.cpp:40-43: a later entry with the same bytes overwrites
        const int64_t b = vocab->begins[i], e = vocab->ends[i];
        if (b < 0 || e < b || e > vocab->n_chars) return set_error(OVTK_E_RANGE, "trie tokenizer: vocab begins/ends outside the chars tensor");
        t.add(vocab->chars + b, size_t(e - b), indices[i]);
    }
    TrieBucketsHost tb;
    if (!tb.build(t)) return set_error(OVTK_E_UNSUPPORTED, "trie tokenizer: the vocabulary's trie has more than 8 million nodes");
    if (int rc = h->root.upload(tb.root.data(), tb.root.size() * sizeof(I2))) return rc;
    if (int rc = h->buckets.upload(tb.buckets.data(), tb.buckets.size() * sizeof(TrieBucket))) return rc;
    h->dev = TrieBucketsDev{h->root.as<I2>(), h->buckets.as<TrieBucket>(), tb.bucket_mask, tb.bucket_shift};
    OVTK_HIP(hipStreamSynchronize(nullptr));
    *out = h.release();
    return OVTK_OK;
}

void ovtk_trie_tokenizer_destroy(ovtk_trie_tokenizer* h) { delete h; }

int ovtk_trie_tokenizer_run(ovtk_trie_tokenizer* h, const ovtk_ragged_strings* in, ovtk_ragged_i32_out* out, int mem, void* stream) {
    if (!h || !in || !out) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_strings_arg(&in->strings, "trie tokenizer input")) return rc;
    if (in->n_rows < 0 || in->n_rows >= INT32_MAX || out->data_capacity < 0) return set_error(OVTK_E_ARG, "trie tokenizer: bad size");
    if (int rc = use_device(h->device)) return rc;
    out->n_rows = in->n_rows;
    out->n_data = 0;
    if (in->n_rows == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(h->device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    TrieRows r{};
    if (int rc = in_source(ws->in_rb, in->ragged_begins, size_t(in->n_rows) * 4, mem, s, &r.ragged_begins)) return rc;
    if (int rc = in_source(ws->in_re, in->ragged_ends, size_t(in->n_rows) * 4, mem, s, &r.ragged_ends)) return rc;
    if (int rc = in_source(ws->in_begins, in->strings.begins, size_t(in->strings.n) * 4, mem, s, &r.begins)) return rc;
    if (int rc = in_source(ws->in_ends, in->strings.ends, size_t(in->strings.n) * 4, mem, s, &r.ends)) return rc;
    if (int rc = in_source(ws->in_chars, in->strings.chars, size_t(in->strings.n_chars), mem, s, &r.chars)) return rc;
    r.n_strings = in->strings.n;
    r.n_chars = in->strings.n_chars;
    r.trie = h->dev;
    r.status = st;
    int32_t *d_b = nullptr, *d_e = nullptr, *d_i = nullptr;
    if (int rc = out_target(ws->out_a, out->begins, size_t(in->n_rows) * 4, mem, &d_b)) return rc;
    if (int rc = out_target(ws->out_b, out->ends, size_t(in->n_rows) * 4, mem, &d_e)) return rc;
    if (int rc = out_target(ws->out_c, out->data, size_t(out->data_capacity) * 4, mem, &d_i)) return rc;
    // the rows' bytes in whole segments -> scan: staging offsets, the segments' rows; a lane per segment walks its guess; a lane per row
    // stitches them and files the count; scan of the counts: the rows' offsets in the output; a wave per row gathers its kept entries
    if (int rc = ws->gen[6].ensure(size_t(in->n_rows) * 4)) return rc;
    if (int rc = ws->gen[7].ensure(size_t(in->n_rows) * 8)) return rc;
    int32_t* lens = ws->gen[6].as<int32_t>();
    long long* stage_off = ws->gen[7].as<long long>();
    const int wave_grid = int(std::min<long long>((in->n_rows + kTileThreads / kWave - 1) / (kTileThreads / kWave), (long long)device_cu_count(h->device) * 32));
    // (a token takes at least a byte, a row's stretch is its bytes in whole segments; rows that share strings need more: second attempt)
    int64_t stage_cap = (std::max<int64_t>(in->strings.n_chars, 1) + int64_t(kTrieSeg) * in->n_rows + kTrieSeg - 1) / kTrieSeg * kTrieSeg;
    uint32_t f = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt) {
            if (int rc = begin_status(*ws.ws, s, &st)) return rc;
            r.status = st;
        }
        if (stage_cap / kTrieSeg > INT32_MAX / 2) return set_error(OVTK_E_UNSUPPORTED, "TrieTokenizer: too much text for one call; split it");
        const size_t n_seg_cap = size_t(stage_cap / kTrieSeg);
        if (int rc = ws->stage.ensure(size_t(stage_cap) * 4)) return rc;
        if (int rc = ws->gen[3].ensure(n_seg_cap * 4)) return rc;
        if (int rc = ws->gen[4].ensure(n_seg_cap * 8)) return rc;
        if (int rc = ws->gen[5].ensure(n_seg_cap * 4)) return rc;
        int32_t* stage = ws->stage.as<int32_t>();
        int32_t* seg_row = ws->gen[3].as<int32_t>();
        unsigned long long* seg_bits = ws->gen[4].as<unsigned long long>();
        int32_t* seg_exit = ws->gen[5].as<int32_t>();
        if ((in->n_rows + kTileElems - 1) / kTileElems > INT32_MAX) return set_error(OVTK_E_UNSUPPORTED, "too many rows for one call; split it");
        if (int rc = ws->tiles.ensure(scan_tiles_bytes(in->n_rows))) return rc;
        launch_scan(ws->marks, "trie_tokenizer", s, in->n_rows, TrieRowStretch{r}, TrieStageOffsets{stage_off, seg_row, (long long)stage_cap},
                    TrieStageFin{st, (long long)stage_cap}, ws->tiles.as<long long>(), st, kFlagRange);
        const unsigned seg_grid = unsigned((n_seg_cap + kTileThreads - 1) / kTileThreads);
        OVTK_LAUNCH(ws->marks, "trie_segments", trie_segments_kernel, seg_grid, kTileThreads, s, r, (const long long*)stage_off, (const int32_t*)seg_row, stage, seg_bits,
                    seg_exit);
        int spread = 1;   // rows on every spread-th lane (trie_rows_kernel): as many waves as give every SIMD four
        while (spread < 8 && (long long)in->n_rows * spread < 4ll * 4 * device_cu_count(h->device) * kWave) spread *= 2;
        const unsigned rows_grid_n = unsigned(((long long)in->n_rows * spread + kTileThreads - 1) / kTileThreads);
        OVTK_LAUNCH(ws->marks, "trie_rows", trie_rows_kernel, rows_grid_n, kTileThreads, s, (long long)in->n_rows, r, (const long long*)stage_off, stage, seg_bits,
                    (const int32_t*)seg_exit, lens, spread);
        if (int rc = scan_and_apply(*ws.ws, s, in->n_rows, FiledLen{lens}, RowOffsets{d_b, d_e, 0},
                                    (long long)std::min<int64_t>(out->data_capacity, INT32_MAX - 1), st, "trie_tokenizer"))
            return rc;
        OVTK_LAUNCH(ws->marks, "trie_gather", each_wave_kernel<TrieGather>, wave_grid, kTileThreads, s, (long long)in->n_rows,
                    TrieGather{stage_off, stage, seg_bits, lens, d_b, d_i, st, (long long)in->n_rows}, (const RunStatus*)st,
                    kFlagOutCapacity | kFlagRange | kFlagItemsOverflow | kFlagStageOverflow);
        if (int rc = finish_status(*ws.ws, s)) return rc;
        f = ws->host_status->flags;
        if (!(f & kFlagStageOverflow) || (f & kFlagRange)) break;
        if (ws->host_status->stage_need >= INT32_MAX - 1) return set_error(OVTK_E_UNSUPPORTED, "TrieTokenizer: the rows' strings add up to 2^31 bytes or more; split the call");
        stage_cap = ws->host_status->stage_need;
    }
    if (f & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside their tensors");
    if (f & kFlagItemsOverflow)
        return set_error(OVTK_E_VOCAB, "TrieTokenizer: no vocabulary entry matches at some byte (the reference does not terminate on this input)");
    if (f & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "TrieTokenizer: output ids buffer too small");
    out->n_data = ws->host_status->n_out;
    int err = 0;
    err = err ? err : copy_back(out->begins, d_b, size_t(in->n_rows) * 4, mem, s);
    err = err ? err : copy_back(out->ends, d_e, size_t(in->n_rows) * 4, mem, s);
    err = err ? err : copy_back(out->data, d_i, size_t(out->n_data) * 4, mem, s);
    if (err) return err;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- string tensor wire format
int64_t ovtk_string_tensor_packed_bytes(int64_t n, int64_t n_chars) { return 8 + 4 * n + n_chars; }

int ovtk_string_tensor_unpack(const uint8_t* packed, int64_t n_bytes, int packed_mem, ovtk_strings_out* out,
                              int64_t rows_capacity, int64_t* n, int device, void* stream) {
    if (!packed || !out || !n || rows_capacity < 0 || out->chars_capacity < 0) return set_error(OVTK_E_ARG, "string_tensor_unpack: bad arguments");
    if (int rc = use_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto peek = [&](int64_t byte_off, int32_t* v) -> int {  // one header word
        if (packed_mem == OVTK_MEM_HOST) { std::memcpy(v, packed + byte_off, 4); return OVTK_OK; }
        OVTK_HIP(hipMemcpyAsync(v, packed + byte_off, 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipStreamSynchronize(s));
        return OVTK_OK;
    };
    // the reference's format checks (src/utils.cpp:21-25)
    if (n_bytes < 4) return set_error(OVTK_E_ARG, "Incorrect packed string tensor format: no batch size in the packed string tensor");
    int32_t batch = 0, total = 0;
    if (int rc = peek(0, &batch)) return rc;
    if (batch < 0 || n_bytes < 8 + 4 * int64_t(batch))
        return set_error(OVTK_E_ARG, "Incorrect packed string tensor format: the packed string tensor must contain first string offset and end indices");
    *n = batch;
    out->n_chars = 0;
    if (batch == 0) return OVTK_OK;
    if (int rc = peek(4 + 4 * int64_t(batch), &total)) return rc;  // end_ids[batch - 1] (string_tensor_unpack.cpp:59)
    if (total < 0 || 8 + 4 * int64_t(batch) + total > n_bytes) return set_error(OVTK_E_RANGE, "packed string tensor: end offsets exceed the buffer");
    if (batch > rows_capacity || total > out->chars_capacity) return set_error(OVTK_E_CAPACITY, "string_tensor_unpack: output buffers too small");
    const hipMemcpyKind kind = packed_mem == OVTK_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    OVTK_HIP(hipMemcpyAsync(out->begins, packed + 4, size_t(batch) * 4, kind, s));  // begin_ids = words 1.., end_ids = words 2.. (:26-27)
    OVTK_HIP(hipMemcpyAsync(out->ends, packed + 8, size_t(batch) * 4, kind, s));
    if (total) OVTK_HIP(hipMemcpyAsync(out->chars, packed + 8 + 4 * size_t(batch), size_t(total), kind, s));
    if (packed_mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));  // the host buffer may be reused on return
    out->n_chars = total;
    return OVTK_OK;
}

int ovtk_string_tensor_pack(const ovtk_strings* in, uint8_t* packed, int64_t capacity, int packed_mem, int64_t* n_bytes,
                            int device, void* stream) {
    if (int rc = check_strings_arg(in, "string_tensor_pack input")) return rc;
    if (!packed || !n_bytes || capacity < 8 + 4 * in->n) return set_error(OVTK_E_CAPACITY, "string_tensor_pack: packed buffer too small for the header");
    if (int rc = use_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    uint8_t* d_packed = packed;
    if (packed_mem == OVTK_MEM_HOST) {
        if (int rc = ws->out_e.ensure(size_t(capacity))) return rc;
        d_packed = ws->out_e.as<uint8_t>();
    }
    int32_t* header = reinterpret_cast<int32_t*>(d_packed);
    const long long cap = std::min<long long>(capacity - 8 - 4 * in->n, INT32_MAX - 1);
    OVTK_LAUNCH(ws->marks, "pack_header", pack_header_kernel, 1, kWave, s, header, int32_t(in->n));
    if (in->n) {
        OVTK_LAUNCH(ws->marks, "check_strings", check_strings_kernel, grid_for_elems(in->n), kBlockThreads, s, in->begins, in->ends,
                    (long long)in->n, (long long)in->n_chars, st);
        if (int rc = scan_and_apply(*ws.ws, s, in->n, PackLen{in->begins, in->ends, (long long)in->n_chars}, PackApply{header}, cap, st, "string_pack"))
            return rc;
        // the bytes: a wave per string, 16 bytes per lane (until round 5 a lane per string, byte by byte, inside the scan's apply pass)
        OVTK_LAUNCH(ws->marks, "string_pack_copy", each_wave_kernel<PackCopy>,
                    int(std::min<long long>((in->n + kTileThreads / kWave - 1) / (kTileThreads / kWave), (long long)device_cu_count(device) * 16)), kTileThreads, s,
                    (long long)in->n, PackCopy{in->begins, in->chars, header, d_packed + 8 + 4 * in->n}, (const RunStatus*)st, kFlagOutCapacity | kFlagRange);
    }
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside the chars tensor");
    if (ws->host_status->flags & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "string_tensor_pack: packed buffer too small");
    *n_bytes = 8 + 4 * in->n + ws->host_status->n_out;
    if (packed_mem == OVTK_MEM_HOST) {
        OVTK_HIP(hipMemcpyAsync(packed, d_packed, size_t(*n_bytes), hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipStreamSynchronize(s));
    }
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- row-shard exchange
struct ovtk_shard_exchange {
    int device = 0;
    ShardGeom g{};
    DevBuf in_a, in_b, in_c, out_a, out_b, out_c;  // staging of host-memory calls (the CPU tests)
    std::vector<Profiler::Mark> marks;
};

int ovtk_shard_exchange_create(int world, int64_t n_rows, int id_bytes, int64_t max_shard_rows, int device,
                               ovtk_shard_exchange** out) {
    if (!out || world < 1 || n_rows < 0 || n_rows >= INT32_MAX || (id_bytes != 2 && id_bytes != 4) || max_shard_rows > n_rows)
        return set_error(OVTK_E_ARG, "shard exchange: bad geometry");
    if (int rc = use_device(device)) return rc;
    auto h = std::make_unique<ovtk_shard_exchange>();
    h->device = device;
    ShardGeom& g = h->g;
    g.n_rows = n_rows; g.world = world; g.id_bytes = id_bytes;
    if (max_shard_rows <= 0) max_shard_rows = (n_rows + world - 1) / world;  // rows balanced by count
    g.max_rows = (max_shard_rows + 3) / 4 * 4;  // ids start 16-byte aligned in the wire
    *out = h.release();
    return OVTK_OK;
}

int64_t ovtk_shard_max_rows(const ovtk_shard_exchange* h) { return h ? h->g.max_rows : -1; }
int64_t ovtk_shard_wire_bytes(const ovtk_shard_exchange* h, int64_t pad_ids) {
    return h ? kShardHeaderBytes + h->g.max_rows * 4 + pad_ids * int64_t(h->g.id_bytes) : -1;
}
void ovtk_shard_exchange_destroy(ovtk_shard_exchange* h) { delete h; }

namespace {
int shard_pad(ovtk_shard_exchange* h, int64_t pad_ids, ShardGeom* g) {
    if (!h) return set_error(OVTK_E_ARG, "shard exchange: null handle");
    if (pad_ids < 0 || pad_ids % 8 || pad_ids >= INT32_MAX) return set_error(OVTK_E_ARG, "shard exchange: pad_ids must be a multiple of 8");
    *g = h->g;
    g->pad_ids = pad_ids;
    g->stride = kShardHeaderBytes + g->max_rows * 4 + pad_ids * g->id_bytes;
    return use_device(h->device);
}
}  // namespace

int ovtk_shard_pack(ovtk_shard_exchange* h, const int32_t* begins, const int32_t* ends, const int32_t* ids, int64_t rows,
                    int64_t n_ids, int64_t pad_ids, void* wire, int mem, void* stream) {
    ShardGeom g{};
    if (int rc = shard_pad(h, pad_ids, &g)) return rc;
    if (rows < 0 || rows > g.max_rows || n_ids < 0 || n_ids >= INT32_MAX || !wire) return set_error(OVTK_E_ARG, "shard_pack: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Profiler::get().resolve(h->marks);  // marks of earlier calls (complete once the caller consumed their results)
    const int32_t *b = nullptr, *e = nullptr, *d = nullptr;
    if (int rc = in_source(h->in_a, begins, size_t(rows) * 4, mem, s, &b)) return rc;
    if (int rc = in_source(h->in_b, ends, size_t(rows) * 4, mem, s, &e)) return rc;
    if (int rc = in_source(h->in_c, ids, size_t(n_ids) * 4, mem, s, &d)) return rc;
    uint8_t* w = nullptr;
    if (int rc = out_target(h->out_a, static_cast<uint8_t*>(wire), size_t(g.stride), mem, &w)) return rc;
    const long long work = std::max<long long>(g.max_rows, std::min<long long>(n_ids, pad_ids));
    const int grid = int(std::min<long long>((work + kBlockThreads - 1) / kBlockThreads + 1, (long long)device_cu_count(h->device) * 8));
    OVTK_LAUNCH(h->marks, "shard_pack", shard_pack_kernel, grid, kBlockThreads, s, b, e, d, (long long)rows, (long long)n_ids, g, w);
    if (int rc = copy_back(static_cast<uint8_t*>(wire), w, size_t(g.stride), mem, s)) return rc;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

int ovtk_shard_unpack(ovtk_shard_exchange* h, const void* wires, int64_t pad_ids, int32_t* out_begins, int32_t* out_ends,
                      int32_t* out_ids, int64_t out_capacity, ovtk_shard_result* result, int mem, void* stream) {
    ShardGeom g{};
    if (int rc = shard_pad(h, pad_ids, &g)) return rc;
    if (!wires || !result || out_capacity < 0) return set_error(OVTK_E_ARG, "shard_unpack: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Profiler::get().resolve(h->marks);
    const uint8_t* w = nullptr;
    if (int rc = in_source(h->in_c, static_cast<const uint8_t*>(wires), size_t(g.stride) * g.world, mem, s, &w)) return rc;
    int32_t *d_b = nullptr, *d_e = nullptr, *d_i = nullptr;
    const size_t rows_bytes = size_t(std::max<int64_t>(g.n_rows, 1)) * 4;
    if (int rc = out_target(h->out_a, out_begins, rows_bytes, mem, &d_b)) return rc;
    if (int rc = out_target(h->out_b, out_ends, rows_bytes, mem, &d_e)) return rc;
    if (int rc = out_target(h->out_c, out_ids, size_t(out_capacity) * 4 + sizeof(ovtk_shard_result), mem, &d_i)) return rc;
    ovtk_shard_result* d_res =
        mem == OVTK_MEM_HOST ? reinterpret_cast<ovtk_shard_result*>(h->out_c.as<uint8_t>() + size_t(out_capacity) * 4) : result;
    const long long share = std::max<long long>(pad_ids, g.max_rows);
    const int chunks = int(std::max<long long>(1, std::min<long long>((share + kBlockThreads * 8 - 1) / (kBlockThreads * 8),
                                                                       (long long)device_cu_count(h->device) * 8 / g.world + 1)));
    OVTK_LAUNCH(h->marks, "shard_unpack", shard_unpack_kernel, dim3(chunks, g.world), kBlockThreads, s, w, g, d_b, d_e, d_i,
                (long long)out_capacity, d_res);
    if (mem == OVTK_MEM_HOST) {
        OVTK_HIP(hipMemcpyAsync(result, d_res, sizeof(ovtk_shard_result), hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipStreamSynchronize(s));
        if (result->status == OVTK_OK) {
            int err = 0;
            err = err ? err : copy_back(out_begins, d_b, size_t(g.n_rows) * 4, mem, s);
            err = err ? err : copy_back(out_ends, d_e, size_t(g.n_rows) * 4, mem, s);
            err = err ? err : copy_back(out_ids, d_i, size_t(result->n_ids) * 4, mem, s);
            if (err) return err;
            OVTK_HIP(hipStreamSynchronize(s));
        }
    }
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- UTF8Validate
int ovtk_utf8_validate(const ovtk_strings* in, int replace_mode, ovtk_strings_out* out, int mem, int device, void* stream) {
    if (int rc = check_strings_arg(in, "utf8_validate input")) return rc;
    if (!out || out->chars_capacity < 0) return set_error(OVTK_E_ARG, "utf8_validate: bad output");
    if (int rc = use_device(device)) return rc;
    out->n_chars = 0;
    if (in->n == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    const int32_t *b = nullptr, *e = nullptr;
    const uint8_t* c = nullptr;
    if (int rc = in_source(ws->in_begins, in->begins, size_t(in->n) * 4, mem, s, &b)) return rc;
    if (int rc = in_source(ws->in_ends, in->ends, size_t(in->n) * 4, mem, s, &e)) return rc;
    if (int rc = in_source(ws->in_chars, in->chars, size_t(in->n_chars), mem, s, &c)) return rc;
    int32_t base = 0;  // begins[0]
    if (mem == OVTK_MEM_HOST) base = in->begins[0];
    else {
        OVTK_HIP(hipMemcpyAsync(&base, b, 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipStreamSynchronize(s));
    }
    if (base < 0 || base > in->n_chars) return set_error(OVTK_E_RANGE, "input begins/ends index outside the chars tensor");
    int32_t *d_b = nullptr, *d_e = nullptr;
    uint8_t* d_c = nullptr;
    if (int rc = out_target(ws->out_c, out->begins, size_t(in->n) * 4, mem, &d_b)) return rc;
    if (int rc = out_target(ws->out_d, out->ends, size_t(in->n) * 4, mem, &d_e)) return rc;
    if (int rc = out_target(ws->out_e, out->chars, size_t(out->chars_capacity), mem, &d_c)) return rc;
    OVTK_LAUNCH(ws->marks, "check_strings", check_strings_kernel, grid_for_elems(in->n), kBlockThreads, s, b, e,
                (long long)in->n, (long long)in->n_chars, st);
    const long long cap = std::min<long long>(out->chars_capacity, INT32_MAX - 1) - base;
    // a wave per string counts (the walk as mask algebra, ops_kernels.hpp) -> scan of the filed lengths -> a wave per string writes
    if (int rc = ws->gen[6].ensure(size_t(in->n) * 4)) return rc;
    int32_t* lens = ws->gen[6].as<int32_t>();
    const int wave_grid = int(std::min<long long>((in->n + kTileThreads / kWave - 1) / (kTileThreads / kWave), (long long)device_cu_count(device) * 16));
    OVTK_LAUNCH(ws->marks, "utf8_count", each_wave_kernel<Utf8WaveCount>, wave_grid, kTileThreads, s, (long long)in->n,
                Utf8WaveCount{b, e, c, (long long)in->n_chars, replace_mode != 0, lens}, (const RunStatus*)nullptr, 0u);
    if (int rc = scan_and_apply(*ws.ws, s, in->n, FiledLen{lens}, RowOffsets{d_b, d_e, (long long)base}, cap, st, "utf8_validate")) return rc;
    OVTK_LAUNCH(ws->marks, "utf8_write", each_wave_kernel<Utf8WaveWrite>, wave_grid, kTileThreads, s, (long long)in->n,
                Utf8WaveWrite{b, e, c, d_b, d_c, replace_mode != 0}, (const RunStatus*)st, kFlagOutCapacity | kFlagRange);
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside the chars tensor");
    if (ws->host_status->flags & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "UTF8Validate: output chars buffer too small");
    out->n_chars = base + ws->host_status->n_out;
    int err = 0;
    err = err ? err : copy_back(out->begins, d_b, size_t(in->n) * 4, mem, s);
    err = err ? err : copy_back(out->ends, d_e, size_t(in->n) * 4, mem, s);
    err = err ? err : copy_back(out->chars, d_c, size_t(out->n_chars), mem, s);
    if (err) return err;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- Truncate
int ovtk_truncate(int n_inputs, const int32_t* begins0, const int32_t* ends0, const int32_t* begins1, const int32_t* ends1,
                  int64_t n, int32_t max_length, const char* side, const char* mode, int32_t* out_begins0,
                  int32_t* out_ends0, int32_t* out_begins1, int32_t* out_ends1, int mem, int device, void* stream) {
    if (n_inputs != 1 && n_inputs != 2)
        return set_error(OVTK_E_ARG, "Only single or pair inputs are supported in Truncation op");  // truncate.cpp:147
    if (n < 0 || n >= INT32_MAX || !side) return set_error(OVTK_E_ARG, "truncate: bad size");
    const std::string sd(side), md(mode ? mode : "");
    if (sd != "left" && sd != "right") return set_error(OVTK_E_ARG, "Unknown truncation side: " + sd);
    int m = 2;
    if (n_inputs == 2) {
        if (md == "only_first") m = 0;
        else if (md == "only_second") m = 1;
        else if (md == "longest_first") m = 2;
        else return set_error(OVTK_E_ARG, "Unknown truncation mode: " + md);  // truncate.cpp:129
    }
    if (int rc = use_device(device)) return rc;
    if (n == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    TruncateArgs a{};
    a.n = n; a.max_length = max_length; a.left = sd == "left"; a.mode = m;
    const size_t bytes = size_t(n) * 4;
    if (int rc = in_source(ws->in_begins, begins0, bytes, mem, s, &a.b0)) return rc;
    if (int rc = in_source(ws->in_ends, ends0, bytes, mem, s, &a.e0)) return rc;
    if (int rc = out_target(ws->out_a, out_begins0, bytes, mem, &a.ob0)) return rc;
    if (int rc = out_target(ws->out_b, out_ends0, bytes, mem, &a.oe0)) return rc;
    if (n_inputs == 2) {
        if (int rc = in_source(ws->in_rb, begins1, bytes, mem, s, &a.b1)) return rc;
        if (int rc = in_source(ws->in_re, ends1, bytes, mem, s, &a.e1)) return rc;
        if (int rc = out_target(ws->out_c, out_begins1, bytes, mem, &a.ob1)) return rc;
        if (int rc = out_target(ws->out_d, out_ends1, bytes, mem, &a.oe1)) return rc;
    }
    OVTK_LAUNCH(ws->marks, "truncate", truncate_kernel, grid_for_elems(n), kBlockThreads, s, a);
    int err = 0;
    err = err ? err : copy_back(out_begins0, a.ob0, bytes, mem, s);
    err = err ? err : copy_back(out_ends0, a.oe0, bytes, mem, s);
    if (n_inputs == 2) {
        err = err ? err : copy_back(out_begins1, a.ob1, bytes, mem, s);
        err = err ? err : copy_back(out_ends1, a.oe1, bytes, mem, s);
    }
    if (err) return err;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    ws->marks.settled();  // host memory: waited above; device memory: the kernel touches the caller's buffers only
    Profiler::get().resolve(ws->marks);  // empty unless profiling (then it waits for the kernel)
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- CombineSegments
namespace {
// Validation + (host memory) staging shared by the ops that take k ragged i32 segments.
int check_segments(const ovtk_ragged_i32* segs, int n_segs, const int32_t* segment_ids, const char* op, int64_t* rows) {
    if (!segs || !segment_ids || n_segs < 1) return set_error(OVTK_E_ARG, std::string(op) + ": bad arguments");
    if (n_segs > kMaxSegments) return set_error(OVTK_E_UNSUPPORTED, std::string(op) + ": more than 16 inputs");
    *rows = 0;
    for (int j = 0; j < n_segs; ++j) {
        if (segs[j].n < 0 || segs[j].n_data < 0 || segs[j].n >= INT32_MAX || segs[j].n_data >= INT32_MAX)
            return set_error(OVTK_E_ARG, std::string(op) + ": bad size");
        *rows = std::max(*rows, segs[j].n);
    }
    for (int j = 0; j < n_segs; ++j)  // combine_segments.cpp:110-116 indexes row i of every non-scalar input
        if (segs[j].n != 1 && segs[j].n != *rows)
            return set_error(OVTK_E_ARG, std::string(op) + ": inputs must have one row or the common number of rows");
    return OVTK_OK;
}

int stage_segments(Workspace& ws, const ovtk_ragged_i32* segs, int n_segs, const int32_t* segment_ids, int mem, hipStream_t s,
                   CombineDev& d) {
    d.n_segs = n_segs;
    if (mem == OVTK_MEM_HOST) {  // one staging buffer for all segments
        size_t total = 0;
        for (int j = 0; j < n_segs; ++j) total += (size_t(segs[j].n) * 2 + size_t(segs[j].n_data)) * 4;
        if (int rc = ws.in_chars.ensure(total)) return rc;
    }
    size_t cur = 0;
    auto place = [&](const int32_t* src, int64_t count, const int32_t** dst) -> int {
        if (mem != OVTK_MEM_HOST) { *dst = src; return OVTK_OK; }
        int32_t* p = reinterpret_cast<int32_t*>(ws.in_chars.as<uint8_t>() + cur);
        if (count) OVTK_HIP(hipMemcpyAsync(p, src, size_t(count) * 4, hipMemcpyHostToDevice, s));
        cur += size_t(count) * 4;
        *dst = p;
        return OVTK_OK;
    };
    for (int j = 0; j < n_segs; ++j) {
        if (int rc = place(segs[j].begins, segs[j].n, &d.begins[j])) return rc;
        if (int rc = place(segs[j].ends, segs[j].n, &d.ends[j])) return rc;
        if (int rc = place(segs[j].data, segs[j].n_data, &d.data[j])) return rc;
        d.n_rows[j] = int32_t(segs[j].n);
        d.n_data[j] = int32_t(segs[j].n_data);
        d.ids[j] = segment_ids[j];
    }
    return OVTK_OK;
}
}  // namespace

int ovtk_combine_segments(const ovtk_ragged_i32* segs, int n_segs, const int32_t* segment_ids, int32_t* out_begins,
                          int32_t* out_ends, int32_t* out_data, int32_t* out_ids, int64_t out_capacity, int64_t* n_out,
                          int mem, int device, void* stream) {
    if (!n_out || out_capacity < 0) return set_error(OVTK_E_ARG, "combine_segments: bad arguments");
    int64_t rows = 0;
    if (int rc = check_segments(segs, n_segs, segment_ids, "combine_segments", &rows)) return rc;
    if (int rc = use_device(device)) return rc;
    *n_out = 0;
    if (rows == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    CombineDev d{};
    if (int rc = stage_segments(*ws.ws, segs, n_segs, segment_ids, mem, s, d)) return rc;
    int32_t *d_b = nullptr, *d_e = nullptr, *d_d = nullptr, *d_i = nullptr;
    if (int rc = out_target(ws->out_a, out_begins, size_t(rows) * 4, mem, &d_b)) return rc;
    if (int rc = out_target(ws->out_b, out_ends, size_t(rows) * 4, mem, &d_e)) return rc;
    if (int rc = out_target(ws->out_c, out_data, size_t(out_capacity) * 4, mem, &d_d)) return rc;
    if (int rc = out_target(ws->out_d, out_ids, size_t(out_capacity) * 4, mem, &d_i)) return rc;
    if (int rc = scan_and_apply(*ws.ws, s, rows, CombineLen{d, st}, CombineApply{d_b, d_e},
                                (long long)std::min<int64_t>(out_capacity, INT32_MAX - 1), st, "combine_rows"))
        return rc;
    const int grid = int(std::min<long long>((rows + kWavesPerBlock - 1) / kWavesPerBlock, (long long)device_cu_count(device) * 8));
    OVTK_LAUNCH(ws->marks, "combine_segments", combine_copy_kernel, grid, kBlockThreads, s, d, (long long)rows,
                (const int32_t*)d_b, d_d, d_i, (const RunStatus*)st);
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "combine_segments: a row reads past its data tensor");
    if (ws->host_status->flags & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "combine_segments: output buffer too small");
    *n_out = ws->host_status->n_out;
    int err = 0;
    err = err ? err : copy_back(out_begins, d_b, size_t(rows) * 4, mem, s);
    err = err ? err : copy_back(out_ends, d_e, size_t(rows) * 4, mem, s);
    err = err ? err : copy_back(out_data, d_d, size_t(*n_out) * 4, mem, s);
    err = err ? err : copy_back(out_ids, d_i, size_t(*n_out) * 4, mem, s);
    if (err) return err;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

// ------------------------------------------------------------------------------- Truncate + CombineSegments + RaggedToDense
int ovtk_encode_tail_run(const ovtk_encode_tail_params* p, int32_t* out_ids, uint8_t* out_mask, int32_t* out_type_ids,
                         int64_t out_capacity, int32_t* out_target_dim, int mem, int device, void* stream) {
    if (!p || !out_ids || !out_target_dim || out_capacity < 0) return set_error(OVTK_E_ARG, "encode_tail: bad arguments");
    int64_t rows = 0;
    if (int rc = check_segments(p->segs, p->n_segs, p->segment_ids, "encode_tail", &rows)) return rc;
    if (p->trunc_a >= p->n_segs || p->trunc_b >= p->n_segs || (p->trunc_b >= 0 && (p->trunc_a < 0 || p->trunc_a == p->trunc_b)))
        return set_error(OVTK_E_ARG, "encode_tail: bad truncated segment indices");
    TailDev t{};
    t.trunc_a = p->trunc_a < 0 ? -1 : p->trunc_a;
    t.trunc_b = p->trunc_b < 0 ? -1 : p->trunc_b;
    t.max_length = p->max_length;
    t.mode = 2;
    if (t.trunc_a >= 0) {
        const std::string sd(p->trunc_side ? p->trunc_side : ""), md(p->trunc_mode ? p->trunc_mode : "");
        if (sd != "left" && sd != "right") return set_error(OVTK_E_ARG, "Unknown truncation side: " + sd);
        t.left = sd == "left";
        if (t.trunc_b >= 0) {
            if (md == "only_first") t.mode = 0;
            else if (md == "only_second") t.mode = 1;
            else if (md == "longest_first") t.mode = 2;
            else return set_error(OVTK_E_ARG, "Unknown truncation mode: " + md);
        }
    }
    t.pad_value = p->pad_value;
    t.type_pad = p->type_pad_value;
    t.pad_right = p->pad_right != 0;
    if (int rc = use_device(device)) return rc;
    *out_target_dim = p->target_dim < 0 ? 0 : p->target_dim;
    if (rows == 0) return OVTK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WorkspaceLease ws(device);
    RunStatus* st = nullptr;
    if (int rc = begin_status(*ws.ws, s, &st)) return rc;
    if (int rc = stage_segments(*ws.ws, p->segs, p->n_segs, p->segment_ids, mem, s, t.c)) return rc;
    int32_t T = p->target_dim;
    if (T < 0) {  // the PaddingStep's ReduceMax: the longest combined row
        OVTK_LAUNCH(ws->marks, "tail_measure", tail_measure_kernel, grid_for_elems(rows), kBlockThreads, s, t, (long long)rows, st);
        if (int rc = finish_status(*ws.ws, s)) return rc;
        if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "encode_tail: a row reads past its data tensor");
        T = ws->host_status->n_out;
    }
    *out_target_dim = T;
    const int64_t need = rows * int64_t(T);
    if (need > out_capacity) return set_error(OVTK_E_CAPACITY, "encode_tail: output buffers too small for [rows, " + std::to_string(T) + "]");
    if (need == 0) return OVTK_OK;
    t.target = T;
    int32_t *d_ids = nullptr, *d_types = nullptr;
    uint8_t* d_mask = nullptr;
    if (int rc = out_target(ws->out_a, out_ids, size_t(need) * 4, mem, &d_ids)) return rc;
    if (out_mask)
        if (int rc = out_target(ws->out_b, out_mask, size_t(need), mem, &d_mask)) return rc;
    if (out_type_ids)
        if (int rc = out_target(ws->out_c, out_type_ids, size_t(need) * 4, mem, &d_types)) return rc;
    const int grid = int(std::min<long long>((rows + kWavesPerBlock - 1) / kWavesPerBlock, (long long)device_cu_count(device) * 8));
    OVTK_LAUNCH(ws->marks, "encode_tail", tail_dense_kernel, grid, kBlockThreads, s, t, (long long)rows, d_ids, d_mask, d_types, st);
    if (int rc = finish_status(*ws.ws, s)) return rc;
    if (ws->host_status->flags & kFlagRange) return set_error(OVTK_E_RANGE, "encode_tail: a row reads past its data tensor");
    int err = 0;
    err = err ? err : copy_back(out_ids, d_ids, size_t(need) * 4, mem, s);
    if (out_mask) err = err ? err : copy_back(out_mask, d_mask, size_t(need), mem, s);
    if (out_type_ids) err = err ? err : copy_back(out_type_ids, d_types, size_t(need) * 4, mem, s);
    if (err) return err;
    if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
    return OVTK_OK;
}

}  // extern "C"

#ifdef OVTK_PROBE
// tools/probe_deferred.py: the wave clocks of wordpiece_deferred_kernel (this translation unit's copy of g_ts; a debug build only).
extern "C" __attribute__((visibility("default"))) int ovtk_debug_probe_ops(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ts), sizeof(unsigned long long) * 8192 * 12);
    if (reset) {
        void* p = nullptr;
        (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_ts));
        (void)hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 12);
    }
    return 0;
}
#endif
