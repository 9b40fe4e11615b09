/*
 * ovtk_amd.h -- C ABI of the MI355X-native tokenizer hot path (libovtk_amd.so).
 *
 * Each entry point replaces the evaluate() body of one custom op of openvino_tokenizers
 * (reference paths are relative to the upstream repository root).  The reference's ops take
 * ov::TensorVector; an OpenVINO adapter maps tensors onto these plain
 * pointer + size structs, so no C++ or torch type crosses this boundary.
 *
 * Data model = the reference's decomposed tensors (src/utils.cpp:84-102):
 *   string tensor         begins i32[n], ends i32[n], chars u8[n_chars]
 *   ragged string tensor  ragged_begins i32[rows], ragged_ends i32[rows] indexing begins/ends
 *   ragged i32 tensor     begins i32[rows], ends i32[rows], data i32[total]
 * Offsets need not be ordered or contiguous on input; outputs are ascending and gap-free,
 * exactly as the reference emits them.
 *
 * Conventions
 *   - every function returns OVTK_OK (0) or a negative OVTK_E_* code; ovtk_last_error() gives
 *     the message for the calling thread (the adapter turns it into OPENVINO_THROW);
 *   - "create" copies/compiles the constant inputs + attributes of an op into device tables
 *     (what the reference builds lazily under call_once / a mutex on first evaluate());
 *     a handle is immutable afterwards, so "run" may be called concurrently from several
 *     threads / streams, like evaluate() on a shared node;
 *   - "run" takes I/O buffers owned by the caller, all in host memory (OVTK_MEM_HOST: staged
 *     over PCIe by the library) or all in device memory (OVTK_MEM_DEVICE: zero copies), enqueues
 *     its kernels on `stream` (a hipStream_t, NULL = default stream) and returns after the
 *     element counts of the variable-length outputs are known (one stream synchronisation);
 *   - variable-length outputs are written into caller buffers of stated capacity (the reference
 *     pre-sizes them the same way and shrinks afterwards); a too small buffer is OVTK_E_CAPACITY;
 *   - there is NO CPU execution path: without a usable HIP device create fails with OVTK_E_HIP.
 */
#ifndef OVTK_AMD_H
#define OVTK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVTK_OK 0
#define OVTK_E_ARG (-1)          /* bad attribute / input (reference: OPENVINO_ASSERT in validate_and_infer_types) */
#define OVTK_E_CAPACITY (-2)     /* output buffer too small (reference: OPENVINO_ASSERT(ragged_offset < size)) */
#define OVTK_E_VOCAB (-3)        /* merge token missing from vocab (reference: std::out_of_range, bpe_tokenizer.cpp:363-366) */
#define OVTK_E_UNSUPPORTED (-4)  /* valid for the reference but outside what the device tables can hold / the split patterns compiled for the GPU */
#define OVTK_E_HIP (-5)          /* HIP runtime failure, or no device */
#define OVTK_E_RANGE (-6)        /* an index would leave its buffer (undefined behaviour in the reference) */

#define OVTK_MEM_HOST 0
#define OVTK_MEM_DEVICE 1

const char* ovtk_last_error(void);
/* ABI version of this header: major*1000 + minor. */
int ovtk_abi_version(void);
/* Name of the device the library runs on ("gfx950 ..."), or NULL when no HIP device is usable. */
const char* ovtk_device_name(void);

/* ---------------------------------------------------------------- tensors */
typedef struct ovtk_strings {
    const int32_t* begins;
    const int32_t* ends;
    const uint8_t* chars;
    int64_t n;       /* number of strings */
    int64_t n_chars; /* size of chars */
} ovtk_strings;

typedef struct ovtk_ragged_strings {
    const int32_t* ragged_begins;
    const int32_t* ragged_ends;
    int64_t n_rows;
    ovtk_strings strings;
} ovtk_ragged_strings;

/* Ragged i32 output: begins/ends sized n_rows by the caller, data sized data_capacity. */
typedef struct ovtk_ragged_i32_out {
    int32_t* begins;
    int32_t* ends;
    int32_t* data;
    int64_t data_capacity;
    int64_t n_data; /* out: elements written */
    int64_t n_rows; /* out: rows written = input rows, except 1 for the all-empty-batch quirk of RegexSplit (regex_split.cpp:129-143) */
} ovtk_ragged_i32_out;

/* ---------------------------------------------------------------- RegexSplit
 * Replaces RegexSplit::evaluate, src/regex_split.cpp:124-324 (+ PCRE2Wrapper::match, src/utils.cpp:396-420).
 * The pattern is not interpreted by PCRE2 on the device: create() recognises the pattern families the
 * reference's converter emits (python/openvino_tokenizers/tokenizer_pipeline.py:392-457) and selects a
 * hand-written gfx950 scanner with identical results; any other pattern is compiled into a leftmost-first DFA
 * (csrc/regex_compile.cpp: the PCRE2 subset listed in regex_compile.hpp -- classes, \p{..} by General_Category,
 * and script, groups, alternation, greedy / lazy / possessive repeats, atomic groups, anchors, look-ahead over
 * anything that is decided within 7 characters of a match's end, look-behind over up to 8 characters) and run one
 * lane per row; only constructs outside that subset (back-references, recursion, conditions, \X, \K ...) are
 * OVTK_E_UNSUPPORTED -- never an approximation, never a CPU fallback; a pattern pcre2_compile itself rejects
 * splits nothing, as the reference's null pattern (src/utils.cpp:264-271).  The GPT-2 family, the Llama-3 family
 * (Llama-3's own pattern, Qwen2's, tiktoken's cl100k_base) are scanned inside the fused encode's lookup kernels; a
 * compiled DFA runs as one pass of its own in front of them on the same stream (no host wait: ovtk_encode_enqueue
 * returns before the split has finished); only the class patterns (\s+, the BERT delimiters) in front of a BPETokenizer,
 * and scanner patterns with max_splits, still take the op's count + write passes with one host wait for the piece count.
 * Inputs 0-4 (+5 skips) of the op = `in` (+ `skips`); input "pattern" and attributes = params. */
typedef struct ovtk_regex_split_params {
    const char* pattern;
    int64_t pattern_len;
    const char* behaviour; /* remove|isolate|contiguous|mergedwithprevious|mergedwithnext (regex_split.cpp:16-22) */
    int invert;
    int max_splits; /* -1 or > 0 (regex_split.cpp:114-117) */
    int device;     /* HIP device ordinal */
} ovtk_regex_split_params;

typedef struct ovtk_regex_split ovtk_regex_split;

typedef struct ovtk_ragged_strings_out {
    int32_t* ragged_begins; /* [max(n_rows,1)] */
    int32_t* ragged_ends;
    int64_t n_rows;         /* out: n_rows, or 1 for the all-empty batch (regex_split.cpp:129-143) */
    int32_t* begins;        /* [capacity]; reference bound: n_chars + n strings (regex_split.cpp:182) */
    int32_t* ends;
    uint8_t* skips;         /* [capacity] or NULL (6-input form) */
    int64_t capacity;
    int64_t n;              /* out: pieces written; -1 = "string outputs alias the inputs" (empty batch) */
} ovtk_ragged_strings_out;

int ovtk_regex_split_create(const ovtk_regex_split_params* params, ovtk_regex_split** out);
/* skips: bool[in->strings.n] or NULL.  chars are never copied: output 4 of the op is its input 4 (regex_split.cpp:203). */
int ovtk_regex_split_run(ovtk_regex_split* h, const ovtk_ragged_strings* in, const uint8_t* skips,
                         ovtk_ragged_strings_out* out, int mem, void* stream);
void ovtk_regex_split_destroy(ovtk_regex_split* h);

/* ---------------------------------------------------------------- SpecialTokensSplit
 * Replaces SpecialTokensSplit::evaluate, src/special_tokens_split.cpp:61-162 (+ PCRE2Wrapper::match_and_find_group,
 * src/utils.cpp:423-461): the op in front of RegexSplit that cuts the added / special tokens out of the text and
 * produces the `skips` flags.  The pattern (input 5 / 6) is the alternation of quoted token lists that
 * SpecialTokensSplitStep generates (python/openvino_tokenizers/tokenizer_pipeline.py:138-159); create() parses it back
 * into literal tokens and strip flags, anything else is OVTK_E_UNSUPPORTED.  `skips` (7-input form) may be NULL;
 * out->skips is mandatory (output 5).  Capacity: the reference sizes the outputs to n_chars (:88-92). */
typedef struct ovtk_special_tokens_split ovtk_special_tokens_split;
int ovtk_special_tokens_split_create(const char* pattern, int64_t pattern_len, int device, ovtk_special_tokens_split** out);
int ovtk_special_tokens_split_run(ovtk_special_tokens_split* h, const ovtk_ragged_strings* in, const uint8_t* skips,
                                  ovtk_ragged_strings_out* out, int mem, void* stream);
void ovtk_special_tokens_split_destroy(ovtk_special_tokens_split* h);

/* ---------------------------------------------------------------- BPETokenizer
 * Replaces BPETokenizer::evaluate + BPETokenizerImpl, src/bpe_tokenizer.cpp:47-388, src/bpe_tokenizer.hpp:40-131.
 * Constant inputs 5-7 (vocab), 8-10 (merges: "left right" lines, or left halves), 11-13 (right halves; NULL
 * for the 11/15-input text form), last four (added tokens + ids; n_added = 0 if absent) and the attributes
 * unk_token, fuse_unk, suffix_indicator, end_suffix, byte_fallback, cache_capacity (bpe_tokenizer.hpp:220-228).
 * cache_capacity: the reference's piece cache is pure memoisation (bpe_tokenizer.cpp:197-205,331-338) and so is its
 * counterpart here, the piece memo: BPE of every vocabulary token as a whole piece, built at create, plus pieces that
 * take several tokens (at most 15 bytes; at most 6 ids when every id of the vocabulary fits 16 bits, 3 otherwise), kept by
 * the device the first time it merges them and while there is room -- the reference's rule (:335 `size() < capacity`,
 * nothing is evicted).  0 disables the memo altogether, exactly as it disables the reference's cache.  Results are
 * identical for every value and every history of calls; a handle may be used from several streams at once.
 * memo_learn: how many such pieces.  The attribute bounds the HOST memory of the reference's std::string cache; the tables
 * here are allocated at create at a size that does not depend on what they come to hold, and the memo's second level (the
 * piece store, below) is sized by this library already.  So by default (0) the first level learns up to
 * max(cache_capacity, the store's capacity) pieces -- what the store would hold for every later call to fetch from
 * merge_kernel is found by the lookup kernel instead --; < 0: exactly cache_capacity pieces, the reference's count;
 * > 0: that many.  (Round 5: ABI 1002.  Until then the count was cache_capacity and an entry held 3 ids.) */
typedef struct ovtk_bpe_params {
    ovtk_strings vocab;
    ovtk_strings merges;       /* text lines or left halves */
    ovtk_strings merges_right; /* .begins == NULL -> text form */
    ovtk_strings added_tokens;
    const int32_t* added_ids;  /* [added_tokens.n] */
    const char* unk_token;
    int64_t unk_token_len;
    int fuse_unk;
    const char* suffix_indicator;
    int64_t suffix_indicator_len;
    const char* end_suffix;
    int64_t end_suffix_len;
    int byte_fallback;
    int64_t cache_capacity;
    int device;
    int64_t memo_store;   /* entries of the handle's piece store (below): 0 = the library's default (ovtk_set_memo_store), < 0 = none */
    int64_t memo_learn;   /* pieces the first level may learn (above): 0 = max(cache_capacity, the store's capacity), < 0 = cache_capacity, > 0 = that many */
} ovtk_bpe_params;

typedef struct ovtk_bpe ovtk_bpe;

int ovtk_bpe_create(const ovtk_bpe_params* params, ovtk_bpe** out);
/* The op itself: pre-split pieces in, ragged ids out (out->data_capacity: reference uses n_chars). */
int ovtk_bpe_run(ovtk_bpe* h, const ovtk_ragged_strings* in, ovtk_ragged_i32_out* out, int mem, void* stream);
void ovtk_bpe_destroy(ovtk_bpe* h);
/* Entries of the piece memo: built at create from the vocabulary (*fixed) and kept since by the device (*learned, at most
 * what memo_learn says; with memo_learn < 0 at most cache_capacity -- the size() of the reference's m_cache,
 * bpe_tokenizer.hpp:150, which the reference does not expose).
 * Waits for the device. */
int ovtk_bpe_memo_entries(ovtk_bpe* h, int64_t* fixed, int64_t* learned);
/* The memo's second level, the piece store: what a handle had to MERGE once (a piece of up to 31 bytes that came to at most 15
 * ids -- 7 when an id needs more than 16 bits) is filed in a table of the handle's own and found there by every later call
 * before any merging starts: one table probe instead of the chain of dependent merge-table lookups.  Like the reference's
 * cache it is pure memoisation -- results are identical with any capacity, any history, and without it -- but it is sized by
 * this library, not by cache_capacity: that attribute bounds the host memory of the reference's std::string cache, an entry
 * here is 64 bytes of HBM.  cache_capacity == 0 still means "no memo at all".  `entries`: capacity of the store of handles
 * created afterwards (default 1048576, and never more than four entries per vocabulary token; 0 = no store: the memo is then exactly the reference's cache_capacity entries).
 * The PROCESS-WIDE default for handles created afterwards on any thread (round 5; until then a setting of the calling thread, which a
 * host that configures on one thread and creates on another never saw); a handle's own value: ovtk_bpe_params::memo_store /
 * ovtk_wordpiece_params::memo_store.  Memory: 3 x 64 bytes per entry of capacity, rounded up to a power of two (GPT-2: 64 MB,
 * Llama-3: 128 MB), allocated and cleared at create; ovtk_bpe_store_entries reports a handle's count and capacity (waits for the
 * device). */
int ovtk_set_memo_store(int64_t entries);
int ovtk_bpe_store_entries(ovtk_bpe* h, int64_t* stored, int64_t* capacity);

/* Fused RegexSplit -> BPETokenizer (the sub-graph tokenizer_pipeline.py:1613-1631 builds for byte-level BPE
 * models): same result as ovtk_regex_split_run followed by ovtk_bpe_run on its outputs, without the piece
 * begins/ends round trip through HBM.  `skips` as for RegexSplit (skipped strings reach BPE unsplit). */
int ovtk_encode_run(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                    ovtk_ragged_i32_out* out, int mem, void* stream);

/* The same call in two halves, for device-resident batches (OVTK_MEM_DEVICE): enqueue launches the kernels on `stream`
 * and returns; finish waits for them -- for their own event, not for work the caller put on the stream afterwards --
 * repeats the launch if a workspace proved too small, fills *out (n_data, n_rows) and frees the pending call, also on
 * error.  The buffers named by `in`, `skips` and `out` must stay valid until finish; the structs themselves are
 * copied.  Lets a host keep the next batch's launches (or an exchange, see the row-shard section) behind the GPU
 * instead of idling in evaluate() as the reference's synchronous ops do. */
/* How the BPE kernels share rows among their (persistent) waves, process-wide: 0 (default) -- every wave owns a fixed
 * share, fastest when the GPU runs nothing else; n > 0 -- rows are handed out n at a time, so that blocks which become
 * resident late because another stream's kernel (RCCL's all-gather during the row-shard exchange) holds their CU take
 * less work instead of finishing last.  Results are identical either way. */
int ovtk_set_row_tickets(int rows_per_ticket);
/* The short path (round 6), process-wide.  ovtk_encode_run / _enqueue (and the wire / dense / special forms) with a pattern the span kernel
 * scans (the GPT-2, Llama-3, o200k and DeepSeek-V3 families) and ovtk_wordpiece_encode_* were four launches per call: the span kernel,
 * lookup_kernel for the rows it leaves (rows of several strings, skipped strings: rare), merge_kernel / wordpiece_deferred_kernel for the
 * pieces the memo does not hold, compact_kernel.  Now the span kernel looks those pieces up in the handle's piece store itself and
 * counts what is in neither table, and the two kernels in the middle are launched only when the handle's last calls had work for them
 * (every call that had sets a count of 4 calls, every call that had not takes one off; a new handle starts with merging expected).
 * When a kernel was left out and had work after all, compact_kernel writes nothing and the kernel follows, with compact_kernel again,
 * from ovtk_encode_finish / inside ovtk_encode_run.  Results are identical either way.
 * 0: never (every call launches all four, round 5's form); 1 (default): as described; 2: every call leaves out both kernels of the
 * middle first, small batches included (tests: the way to the second set of launches). */
int ovtk_set_short_path(int mode);
/* Process-wide counts since the library was loaded: calls whose first set of launches left a kernel of the middle out, and those of
 * them that needed no second set. */
int ovtk_short_path_stats(int64_t* tried, int64_t* exact);

typedef struct ovtk_pending ovtk_pending;
/* SpecialTokensSplit -> RegexSplit -> BPETokenizer in one call: the sub-graph every converted HF byte-level BPE tokenizer runs
 * (python/openvino_tokenizers/tokenizer_pipeline.py:1613-1636; SpecialTokensSplit::evaluate src/special_tokens_split.cpp:61-162 in
 * front of RegexSplit::evaluate src/regex_split.cpp:124-324 and BPETokenizer::evaluate src/bpe_tokenizer.cpp:47-164).  The split
 * strings and their skip flags stay in device buffers of the reference's capacity between the stages; nothing waits in between.
 * `in` / `skips`: SpecialTokensSplit's inputs 0-4 (+ 5 of the 7-input form, or NULL); `out`: BPETokenizer's outputs.  Same result
 * as ovtk_special_tokens_split_run followed by ovtk_encode_run on its outputs.  run: blocking, buffers in `mem`; enqueue: device
 * buffers, completed by ovtk_encode_finish like ovtk_encode_enqueue. */
int ovtk_encode_special_run(ovtk_special_tokens_split* special, ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in,
                            const uint8_t* skips, ovtk_ragged_i32_out* out, int mem, void* stream);
int ovtk_encode_special_enqueue(ovtk_special_tokens_split* special, ovtk_regex_split* split, ovtk_bpe* bpe,
                                const ovtk_ragged_strings* in, const uint8_t* skips, const ovtk_ragged_i32_out* out, void* stream,
                                ovtk_pending** pending);
/* The whole graph of a converted byte-level BPE tokenizer behind StringTensorUnpack, in ONE call and without the ragged ids tensor
 * (tokenizer_pipeline.py:1613-1636 + TruncationStep / CombineSegmentsStep / PaddingStep): [SpecialTokensSplit ->] RegexSplit ->
 * BPETokenizer -> Truncate (src/truncate.cpp:37-150, one input) -> CombineSegments with constant ids in front / behind
 * (src/combine_segments.cpp:36-134: the post-processor's BOS / EOS) -> RaggedToDense x 2 (src/ragged_to_dense.cpp:70-174) =
 * input_ids i32[n_rows, T] and attention_mask u8[n_rows, T].  The encode's last pass writes the dense tensors itself.  Same values as
 * ovtk_encode_special_run (or ovtk_encode_run) followed by ovtk_encode_tail_run.  special may be NULL.  Device buffers; completed by
 * ovtk_encode_dense_finish, which reports T (target_dim < 0: the longest row) and the number of ids before truncation; OVTK_E_CAPACITY
 * when n_rows * T exceeds `capacity` cells (*width = the T it needs). */
typedef struct {
    int32_t max_length;        /* Truncate: ids of a row that stay */
    int trunc_left;            /* 0 "right": the first max_length stay; 1 "left": the last */
    int pad_right;             /* RaggedToDense pad_right */
    int32_t pad_value;         /* input_ids padding (attention_mask pads with 0) */
    int32_t target_dim;        /* row width; < 0: the longest row (the PaddingStep's ReduceMax) */
    const int32_t* prefix;     /* HOST memory: constant ids in front of every row's ids (at most 4) */
    int n_prefix;
    const int32_t* suffix;     /* ... and behind them (at most 4) */
    int n_suffix;
} ovtk_dense_params;
int ovtk_encode_dense_enqueue(ovtk_special_tokens_split* special, ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in,
                              const uint8_t* skips, const ovtk_dense_params* params, int32_t* out_ids, uint8_t* out_mask, int64_t capacity,
                              void* stream, ovtk_pending** pending);
int ovtk_encode_dense_finish(ovtk_pending* pending, int32_t* width, int64_t* n_ids);
int ovtk_encode_enqueue(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                        const ovtk_ragged_i32_out* out, void* stream, ovtk_pending** pending);
int ovtk_encode_finish(ovtk_pending* pending, ovtk_ragged_i32_out* out);
/* The two halves for HOST buffers -- what a CPU-plugin evaluate() holds (every input of BPETokenizer::evaluate is a host
 * tensor, src/bpe_tokenizer.cpp:122-140).  enqueue puts the copies of the inputs to the device and the kernels on
 * `stream` and returns; finish (ovtk_encode_finish) returns when begins / ends and exactly n_data ids are in the caller's
 * buffers.  With PINNED buffers (hipHostMalloc / hipHostRegister) and a split pattern the fused scanners cover (the
 * GPT-2 / Llama-3 families; see RegexSplit above for the others) nothing blocks in between: the input copies are
 * asynchronous, and the output buffers are written by the last kernel itself through their device-side addresses (no
 * device-to-host copy; finish only waits for the call's event) -- a host that keeps a few batches in flight on different
 * streams runs the host-to-device copies of the next batches under the kernels and PCIe stores of this one.  Pageable
 * buffers work too (staged on the device, copied synchronously).  Buffers must stay valid and untouched until finish;
 * pinned output buffers may hold partial data while the call is in flight and after a failed call. */
int ovtk_encode_enqueue_host(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                             const ovtk_ragged_i32_out* out, void* stream, ovtk_pending** pending);

/* The same, from the PACKED form of the string tensor -- [i32 n][i32 begin_0][i32 end_i x n][bytes], parse_packed_strings
 * src/utils.cpp:18-29, what a host holds before StringTensorUnpack (src/string_tensor_unpack.cpp:53-71): `packed` is HOST
 * memory (pinned: the copy is asynchronous), ONE buffer crosses PCIe, begins / ends / chars are views of its device copy and
 * never exist as separate tensors; every string is a row (the ragged dimension converted pipelines build with Range,
 * tokenizer_pipeline.py:1668-1676).  = StringTensorUnpack -> RegexSplit -> BPETokenizer.  `out` lives in out_mem; pinned
 * host outputs are written by the kernels as above.  The packed buffer must stay valid until finish. */
int ovtk_encode_enqueue_packed(ovtk_regex_split* split, ovtk_bpe* bpe, const uint8_t* packed, int64_t n_bytes,
                               const ovtk_ragged_i32_out* out, int out_mem, void* stream, ovtk_pending** pending);

/* ---------------------------------------------------------------- WordpieceTokenizer
 * Replaces WordpieceTokenizer::evaluate, src/wordpiece_tokenizer.cpp:49-133.  Inputs 5-7 + attributes at
 * create; input 8 (unk_token_id) is read every call, as in the reference (:74). */
typedef struct ovtk_wordpiece_params {
    ovtk_strings vocab;
    const char* suffix_indicator;
    int64_t suffix_indicator_len;
    int max_bytes_per_word;
    int device;
    int64_t memo_store;   /* entries of the handle's word store (ovtk_wordpiece_encode_run): 0 = the library's default, < 0 = none */
} ovtk_wordpiece_params;
typedef struct ovtk_wordpiece ovtk_wordpiece;
int ovtk_wordpiece_create(const ovtk_wordpiece_params* params, ovtk_wordpiece** out);
int ovtk_wordpiece_run(ovtk_wordpiece* h, const ovtk_ragged_strings* in, int32_t unk_token_id,
                       ovtk_ragged_i32_out* out, int mem, void* stream);
/* Fused RegexSplit(\s+, remove) -> RegexSplit(BERT delimiters, isolate) -> WordpieceTokenizer: the sub-graph
 * tokenizer_pipeline.py:392-435 (bert_splitter) + :641-659 builds for BERT models; same result as chaining the three ops,
 * without the word begins/ends round trips through HBM.  Any other pair of split handles is OVTK_E_UNSUPPORTED.
 * The handle memoises words -> ids as it meets them (a first-level table of the vocabulary's own words plus up to 4 V learned
 * ones of at most 15 bytes and 6 ids, a second-level store sized like the BPE handle's: ovtk_set_memo_store): pure memoisation of
 * wordpiece_tokenizer.cpp:94-130, no result depends on it or on the calls before; a word that came out as unk_token_id is never
 * kept, so input 8 may differ from call to call.  Calls on one handle may run concurrently (tables are insert-only). */
int ovtk_wordpiece_encode_run(ovtk_wordpiece* h, ovtk_regex_split* whitespace, ovtk_regex_split* delimiters,
                              const ovtk_ragged_strings* in, int32_t unk_token_id, ovtk_ragged_i32_out* out, int mem,
                              void* stream);
/* ovtk_wordpiece_encode_run in two halves (device memory; see ovtk_encode_enqueue): finish with ovtk_encode_finish. */
int ovtk_wordpiece_encode_enqueue(ovtk_wordpiece* h, ovtk_regex_split* whitespace, ovtk_regex_split* delimiters,
                                  const ovtk_ragged_strings* in, int32_t unk_token_id, const ovtk_ragged_i32_out* out,
                                  void* stream, ovtk_pending** pending);
void ovtk_wordpiece_destroy(ovtk_wordpiece* h);

/* ---------------------------------------------------------------- VocabEncoder
 * Replaces VocabEncoder::evaluate_impl<T>, src/vocab_encoder.cpp:55-94.  value_size 4 (i32) or 8 (i64). */
typedef struct ovtk_vocab_encoder_params {
    ovtk_strings keys;
    const void* values;
    int value_size;
    int device;
} ovtk_vocab_encoder_params;
typedef struct ovtk_vocab_encoder ovtk_vocab_encoder;
int ovtk_vocab_encoder_create(const ovtk_vocab_encoder_params* params, ovtk_vocab_encoder** out);
/* out: T[in->n]; default_value: pointer to one T in HOST memory (input 7). */
int ovtk_vocab_encoder_run(ovtk_vocab_encoder* h, const ovtk_strings* in, const void* default_value, void* out,
                           int mem, void* stream);
void ovtk_vocab_encoder_destroy(ovtk_vocab_encoder* h);

/* ---------------------------------------------------------------- RaggedToDense
 * Replaces RaggedToDense::evaluate, src/ragged_to_dense.cpp:70-174.  Stateless.
 * data: n_data ragged elements of elem_size*inner_elems bytes; out_dense/out_mask: [n_rows, target_dim, inner].
 * default_value points to one element (elem_size bytes) in HOST memory.  out_mask may be NULL. */
int ovtk_ragged_to_dense(const int32_t* begins, const int32_t* ends, int64_t n_rows, const void* data,
                         int64_t n_data, int elem_size, int64_t inner_elems, int32_t target_dim,
                         const void* default_value, int pad_right, int pad_max_length, void* out_dense,
                         uint8_t* out_mask, int mem, int device, void* stream);

/* ---------------------------------------------------------------- VocabDecoder / ByteFallback / FuzeRagged
 * Replace VocabDecoder::evaluate (src/vocab_decoder.cpp:23-87), ByteFallback::evaluate
 * (src/byte_fallback.cpp:16-50) and FuzeRagged::evaluate (src/fuze.cpp:20-40). */
typedef struct ovtk_vocab_decoder_params {
    ovtk_strings vocab;
    const int32_t* skip_tokens; /* attribute skip_tokens; input 4 overrides it per call */
    int64_t n_skip_tokens;
    int device;
} ovtk_vocab_decoder_params;
typedef struct ovtk_vocab_decoder ovtk_vocab_decoder;

typedef struct ovtk_strings_out {
    int32_t* begins; /* [n] */
    int32_t* ends;
    uint8_t* chars;  /* [chars_capacity] */
    int64_t chars_capacity;
    int64_t n_chars; /* out */
} ovtk_strings_out;

int ovtk_vocab_decoder_create(const ovtk_vocab_decoder_params* params, ovtk_vocab_decoder** out);
/* ids: i32[batch, seq_len].  skip_tokens_input: NULL -> use the attribute; else i32[n] in HOST memory (input 4).
 * out_ragged_begins/ends: [batch]; out->begins/ends: [batch * max(seq_len,1)]. */
int ovtk_vocab_decoder_run(ovtk_vocab_decoder* h, const int32_t* ids, int64_t batch, int64_t seq_len,
                           const int32_t* skip_tokens_input, int64_t n_skip_tokens_input,
                           int32_t* out_ragged_begins, int32_t* out_ragged_ends, ovtk_strings_out* out,
                           int mem, void* stream);
void ovtk_vocab_decoder_destroy(ovtk_vocab_decoder* h);

/* out->begins/ends: [in->n]; out->chars capacity: reference uses in->n_chars (byte_fallback.cpp:24). */
int ovtk_byte_fallback(const ovtk_strings* in, ovtk_strings_out* out, int mem, int device, void* stream);

int ovtk_fuze_ragged(const int32_t* ragged_begins, const int32_t* ragged_ends, int64_t n_rows,
                     const int32_t* begins, const int32_t* ends, int64_t n, int32_t* out_begins,
                     int32_t* out_ends, int mem, int device, void* stream);

/* ---------------------------------------------------------------- string tensor wire format
 * The packed u8 form of a string tensor, [i32 n][i32 begin_0][i32 end_i x n][bytes] (parse_packed_strings,
 * src/utils.cpp:18-29): the staging step either side of the path -- ONE buffer crosses PCIe, the decomposed tensors
 * the ops consume (begins / ends / chars) exist only in HBM.
 * ovtk_string_tensor_unpack replaces the u8 branch of StringTensorUnpack::evaluate, src/string_tensor_unpack.cpp:53-71:
 * `packed` lives in packed_mem (host or device), the outputs are DEVICE buffers (out->begins/ends: capacity
 * rows_capacity; out->chars: chars_capacity); *n = strings, out->n_chars = bytes.
 * ovtk_string_tensor_pack is the inverse (what StringTensorPack + the Python pack_strings helper produce,
 * python/openvino_tokenizers/utils.py): device strings, gaps and order as they are -> gap-free packed buffer in
 * packed_mem of ovtk_string_tensor_packed_bytes(n, sum of lengths) bytes. */
int64_t ovtk_string_tensor_packed_bytes(int64_t n, int64_t n_chars);
int ovtk_string_tensor_unpack(const uint8_t* packed, int64_t n_bytes, int packed_mem, ovtk_strings_out* out,
                              int64_t rows_capacity, int64_t* n, int device, void* stream);
int ovtk_string_tensor_pack(const ovtk_strings* in, uint8_t* packed, int64_t capacity, int packed_mem,
                            int64_t* n_bytes, int device, void* stream);

/* ---------------------------------------------------------------- row-shard exchange
 * No reference counterpart (the reference is single-process).  Rows shard contiguously over the ranks of one node
 * (any contiguous partition in rank order: balanced by row count, or by bytes -- every wire says how many rows it holds;
 * max_shard_rows = the largest shard's row count, <= 0: ceil(n_rows / world)); the one exchange step is an all-gather
 * of fixed-size "wires": a 16-byte header (n_ids, n_rows), i32 ends[max_rows] (the shard's own end offsets), then
 * pad_ids ids of id_bytes (2 when every id < 65536, else 4) bytes.
 *   ovtk_shard_pack    builds this rank's wire from the ragged ids an encode call returned (ascending, gap-free
 *                      from 0).  Device memory: only enqueues work on `stream`.
 *   (the caller all-gathers the wires: RCCL)
 *   ovtk_shard_unpack  rebuilds the global ragged tensor from the `world` wires in one kernel (rank r's ids start at the
 *                      sum of the n_ids before it, its rows' offsets are that base plus the local ones): begins/ends
 *                      [n_rows], ids widened to i32.  Device memory: only enqueues work on `stream`; `result` lives
 *                      in device memory.  Host memory: synchronous, `result` in host memory.
 * result->status: OVTK_OK; OVTK_E_ARG when the wires' row counts are not a partition of n_rows (or exceed max_shard_rows);
 * OVTK_E_CAPACITY with max_shard_ids > pad_ids when some shard held more ids than the wire
 * has room for (identical on every rank: repeat the exchange with a larger pad); OVTK_E_RANGE when out_capacity is
 * too small; OVTK_E_UNSUPPORTED beyond 2^31 ids. */
typedef struct {
    int64_t n_ids;          /* ids in the global tensor */
    int64_t max_shard_ids;  /* largest shard */
    int64_t status;
    int64_t reserved;
} ovtk_shard_result;
typedef struct ovtk_shard_exchange ovtk_shard_exchange;
int ovtk_shard_exchange_create(int world, int64_t n_rows, int id_bytes, int64_t max_shard_rows, int device,
                               ovtk_shard_exchange** out);
int64_t ovtk_shard_max_rows(const ovtk_shard_exchange* h);                    /* lens slots per wire */
int64_t ovtk_shard_wire_bytes(const ovtk_shard_exchange* h, int64_t pad_ids); /* pad_ids: multiple of 8 */
int ovtk_shard_pack(ovtk_shard_exchange* h, const int32_t* begins, const int32_t* ends, const int32_t* ids,
                    int64_t rows, int64_t n_ids, int64_t pad_ids, void* wire, int mem, void* stream);
int ovtk_shard_unpack(ovtk_shard_exchange* h, const void* wires, int64_t pad_ids, int32_t* out_begins,
                      int32_t* out_ends, int32_t* out_ids, int64_t out_capacity, ovtk_shard_result* result,
                      int mem, void* stream);
void ovtk_shard_exchange_destroy(ovtk_shard_exchange* h);
/* The fused encode of ovtk_encode_enqueue with its result written straight in wire form (device memory): the ids leave the
 * last kernel narrowed to id_bytes and in place for the all-gather -- no i32 ids buffer, no ovtk_shard_pack.  `wire`:
 * ovtk_shard_wire_bytes(h, pad_ids) bytes; max_rows = ovtk_shard_max_rows(h); id_bytes as given to the exchange.  Finish
 * with ovtk_encode_finish: out->n_data = the shard's id count -- above pad_ids the wire is cut exactly as ovtk_shard_pack
 * cuts it (every receiver sees OVTK_E_CAPACITY in its unpack verdict; encode again into a larger wire). */
int ovtk_encode_enqueue_wire(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                             void* wire, int64_t max_rows, int64_t pad_ids, int id_bytes, void* stream, ovtk_pending** pending);

/* ---------------------------------------------------------------- TrieTokenizer (RWKV)
 * Replaces TrieTokenizer::evaluate, src/trie_tokenizer.cpp:23-81: greedy longest match over a trie of vocab[i] ->
 * indices[i] (inputs 5-8, consumed at create like the reference's lazy init :27-45).  Where no entry matches at some
 * byte the reference's loop never terminates (:72-75); this library returns OVTK_E_VOCAB instead.
 * out->data capacity: the reference allocates in->strings.n_chars ids (:60). */
typedef struct ovtk_trie_tokenizer ovtk_trie_tokenizer;
int ovtk_trie_tokenizer_create(const ovtk_strings* vocab, const int32_t* indices, int device, ovtk_trie_tokenizer** out);
int ovtk_trie_tokenizer_run(ovtk_trie_tokenizer* h, const ovtk_ragged_strings* in, ovtk_ragged_i32_out* out, int mem,
                            void* stream);
void ovtk_trie_tokenizer_destroy(ovtk_trie_tokenizer* h);

/* ---------------------------------------------------------------- UTF8Validate
 * Replaces UTF8Validate::evaluate, src/utf8_validate.cpp:18-143.  replace_mode 0: drop invalid bytes, 1: U+FFFD.
 * out->begins/ends: [in->n]; out->chars capacity: the reference allocates 3 * in->n_chars (:31-33).  Offsets start
 * at in->begins[0] like the reference's (:46); out->n_chars = last offset (bytes before begins[0] are not written). */
int ovtk_utf8_validate(const ovtk_strings* in, int replace_mode, ovtk_strings_out* out, int mem, int device,
                       void* stream);

/* ---------------------------------------------------------------- Truncate / CombineSegments
 * ovtk_truncate replaces Truncate::evaluate, src/truncate.cpp:37-150 (the reference edits begins/ends in place; here
 * out_* may alias the inputs).  n_inputs 1: begins1/ends1/out_*1 and mode are ignored.  side: "left"|"right";
 * mode: "only_first"|"only_second"|"longest_first". */
int ovtk_truncate(int n_inputs, const int32_t* begins0, const int32_t* ends0, const int32_t* begins1,
                  const int32_t* ends1, int64_t n, int32_t max_length, const char* side, const char* mode,
                  int32_t* out_begins0, int32_t* out_ends0, int32_t* out_begins1, int32_t* out_ends1, int mem,
                  int device, void* stream);

/* ovtk_combine_segments replaces CombineSegments::evaluate, src/combine_segments.cpp:36-134, for i32 elements (token
 * ids, the only element type tokenizer_pipeline.py builds it with).  segs[j]: begins/ends [n], data [n_data]; n == 1
 * is broadcast over the rows.  segment_ids: i32[n_segs] in HOST memory (the op's last input).
 * out_begins/out_ends: [max n]; out_data/out_ids: capacity out_capacity, *n_out elements written. */
typedef struct {
    const int32_t* begins;
    const int32_t* ends;
    const int32_t* data;
    int64_t n;
    int64_t n_data;
} ovtk_ragged_i32;
int ovtk_combine_segments(const ovtk_ragged_i32* segs, int n_segs, const int32_t* segment_ids, int32_t* out_begins,
                          int32_t* out_ends, int32_t* out_data, int32_t* out_ids, int64_t out_capacity,
                          int64_t* n_out, int mem, int device, void* stream);

/* Fused Truncate -> CombineSegments -> RaggedToDense x 2 (tokenizer_pipeline.py TruncationStep, CombineSegmentsStep,
 * PaddingStep): input_ids / attention_mask / token_type_ids straight from the ragged segments, same values as chaining
 * ovtk_truncate, ovtk_combine_segments and ovtk_ragged_to_dense.  trunc_a / trunc_b: indices into segs of the truncated
 * segment(s) (-1: none; both >= 0: pair truncation with `mode`).  target_dim < 0: the longest combined row, as the
 * PaddingStep's ReduceMax computes it (one extra kernel and host wait); with pad_max_length the larger of max_length...
 * is the caller's business: pass the target you want.  *out_target_dim = the row width used; outputs are [n_rows, width]
 * with capacity out_capacity elements each (out_mask / out_type_ids may be NULL). */
typedef struct {
    const ovtk_ragged_i32* segs;
    int n_segs;
    const int32_t* segment_ids;   /* i32[n_segs], HOST memory */
    int trunc_a, trunc_b;
    int32_t max_length;
    const char* trunc_side;       /* "left" | "right" */
    const char* trunc_mode;       /* "only_first" | "only_second" | "longest_first" (pair truncation) */
    int32_t target_dim;
    int32_t pad_value;            /* input_ids padding */
    int32_t type_pad_value;       /* token_type_ids padding */
    int pad_right;
} ovtk_encode_tail_params;
int ovtk_encode_tail_run(const ovtk_encode_tail_params* p, int32_t* out_ids, uint8_t* out_mask, int32_t* out_type_ids,
                         int64_t out_capacity, int32_t* out_target_dim, int mem, int device, void* stream);

/* Fused VocabDecoder -> [ByteFallback] -> FuzeRagged (tokenizer_pipeline.py:1321-1371): one string per row.
 * Offsets are int32 as in the reference (src/vocab_decoder.cpp:62-63,69,80 count chars in int32): a call whose output would
 * not fit out->chars_capacity (at most 2^31 - 2 bytes) returns OVTK_E_CAPACITY with out->n_chars = the bytes it needs
 * (INT32_MAX: beyond int32).  Batches of that size -- BASELINE config 5, 1 M x 2 048 ids = 8-9 GB of text -- are cut into
 * row chunks by the caller; the Python mirror's FusedDetokenizer.evaluate_chunked() is that loop (chunks pipelined over
 * HIP streams with the two-half form below, an overflowing chunk is cut again). */
int ovtk_detokenize_run(ovtk_vocab_decoder* h, const int32_t* ids, int64_t batch, int64_t seq_len,
                        const int32_t* skip_tokens_input, int64_t n_skip_tokens_input, int byte_fallback,
                        ovtk_strings_out* out, int mem, void* stream);
/* ovtk_detokenize_run in two halves for device buffers (OVTK_MEM_DEVICE), like ovtk_encode_enqueue / ovtk_encode_finish:
 * enqueue launches the passes on `stream` and returns; finish waits for that call (its own event), fills out->n_chars (on
 * OVTK_E_CAPACITY: the bytes the call needs), and releases `pending` whatever happens.  Calls in flight are independent. */
int ovtk_detokenize_enqueue(ovtk_vocab_decoder* h, const int32_t* ids, int64_t batch, int64_t seq_len, const int32_t* skip_in,
                            int64_t n_skip_in, int byte_fallback, ovtk_strings_out* out, void* stream, ovtk_pending** pending);
int ovtk_detokenize_finish(ovtk_pending* pending, ovtk_strings_out* out);

/* ---------------------------------------------------------------- measurement hooks (bench.py)
 * With profiling on, every kernel launch of the library is bracketed by hipEvents on the stream it is
 * launched on; times are accumulated per kernel name after the call's own synchronisation. */
void ovtk_profile_enable(int on);
void ovtk_profile_reset(void);
/* Returns 0 and fills total_ms / launches for `kernel`, or -1 if it has not been launched. */
int ovtk_profile_get(const char* kernel, double* total_ms, int64_t* launches);
/* Writes up to `cap` bytes of a '\n'-separated "name total_ms launches" table. Returns bytes needed. */
int64_t ovtk_profile_dump(char* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
