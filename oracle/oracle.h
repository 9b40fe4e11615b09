/*
 * oracle.h -- C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the reference's
 * algorithms for the tokenizer hot path (openvino_tokenizers src/*.cpp).  It is the
 * checker for the HIP path and the "port" CPU baseline of bench.py.  Nothing under
 * openvino_tokenizers_amd/ (the product) may include, link, import or call it.
 *
 * Parity status: the reference itself cannot be built in this image (it needs the
 * OpenVINO developer package, PCRE2 10.46 and sentencepiece fetched at configure time),
 * so the oracle is pinned against
 *   - the reference's own known-answer vectors for RegexSplit (tests/layer_tests.py:331-389)
 *     and RaggedToDense (tests/layer_tests.py:497-598), and
 *   - HuggingFace `tokenizers` 0.22.2 -- the same differential oracle the reference's
 *     tests/tokenizers_test.py uses -- on tokenizers trained in-process,
 * see tests/test_oracle_*.py.  Third-party arithmetic: PCRE2 (system libpcre2-8 10.39,
 * reference pins 10.46: identical semantics, Unicode tables 14.0 vs 16.0) and libstdc++'s
 * std::priority_queue (used directly, same library family as the reference build).
 *
 * All strings are the reference's decomposed layout (src/utils.cpp:84-102):
 *   string i = chars[begins[i] .. ends[i]),  ragged row r = elements [rb[r] .. re[r]).
 * Every function returns 0 on success, a negative ORC_E_* code otherwise.
 */
#ifndef OVTK_ORACLE_H
#define OVTK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_E_ARG -1       /* bad argument (unknown behaviour string, bad max_splits ...) */
#define ORC_E_CAPACITY -2  /* an output buffer is too small (reference: OPENVINO_ASSERT) */
#define ORC_E_VOCAB -3     /* merge token missing from vocab (reference: std::out_of_range) */
#define ORC_E_PCRE2 -4     /* libpcre2-8 could not be loaded */
#define ORC_E_RANGE -5     /* an index would leave its buffer (undefined behaviour in the reference) */

const char* orc_last_error(void);

/* ---- RegexSplit : src/regex_split.cpp:124-324, PCRE2Wrapper src/utils.cpp:256-272,396-420 ---- */
typedef struct orc_regex orc_regex;
/* behaviour: remove|isolate|contiguous|mergedwithprevious|mergedwithnext */
int orc_regex_split_create(const char* pattern, int64_t pattern_len, const char* behaviour,
                           int invert, int max_splits, orc_regex** out);
void orc_regex_split_destroy(orc_regex*);
/* 1 when PCRE2 compiled the pattern, 0 when it rejected it (the handle then never matches: src/utils.cpp:264-271, 397-399). */
int orc_regex_compiled(const orc_regex*);
/* skips may be NULL (6-input form).  Outputs: out_rb/out_re sized B (or 1 when nchars == 0:
 * *n_rows_out tells), out_begins/out_ends/out_skips sized `cap` (reference bound: nchars + N). */
int orc_regex_split_run(const orc_regex*, const int32_t* rb, const int32_t* re, int64_t B,
                        const int32_t* begins, const int32_t* ends, int64_t N,
                        const uint8_t* chars, int64_t nchars, const uint8_t* skips,
                        int32_t* out_rb, int32_t* out_re, int64_t* n_rows_out,
                        int32_t* out_begins, int32_t* out_ends, uint8_t* out_skips,
                        int64_t cap, int64_t* n_out);
/* Raw matcher access for table generation / differential tests: one match from `start`.
 * Returns 1 and fills m[2] on a match, 0 on no match. */
int orc_regex_match(const orc_regex*, const uint8_t* s, int64_t len, int64_t start, int64_t* m);

/* ---- SpecialTokensSplit : src/special_tokens_split.cpp:61-162 (the pattern is compiled with orc_regex_split_create;
 * behaviour / invert / max_splits are not used).  skips may be NULL (6-input form); out_skips is always written.
 * Output buffers sized `cap` (reference bound: nchars). */
int orc_special_tokens_split_run(const orc_regex*, const int32_t* rb, const int32_t* re, int64_t B,
                                 const int32_t* begins, const int32_t* ends, const uint8_t* chars,
                                 const uint8_t* skips, int32_t* out_rb, int32_t* out_re,
                                 int32_t* out_begins, int32_t* out_ends, uint8_t* out_skips, int64_t cap, int64_t* n_out);

/* ---- BPETokenizer : src/bpe_tokenizer.cpp:47-388, src/bpe_tokenizer.hpp:40-131 ---- */
typedef struct orc_bpe orc_bpe;
/* merges: if mr_begins == NULL the merges are "left right" text lines in (ml_*), split at the
 * first space (11/15-input form); otherwise left halves in ml_*, right halves in mr_* (14/18). */
int orc_bpe_create(const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                   const int32_t* ml_begins, const int32_t* ml_ends, const uint8_t* ml_chars,
                   const int32_t* mr_begins, const int32_t* mr_ends, const uint8_t* mr_chars, int64_t M,
                   const int32_t* a_begins, const int32_t* a_ends, const uint8_t* a_chars,
                   const int32_t* a_ids, int64_t A,
                   const char* unk_token, int64_t unk_len, int fuse_unk,
                   const char* suffix_indicator, int64_t si_len,
                   const char* end_suffix, int64_t es_len,
                   int byte_fallback, int64_t cache_capacity, orc_bpe** out);
void orc_bpe_destroy(orc_bpe*);
/* out_ids capacity `cap` (reference: nchars). */
int orc_bpe_run(orc_bpe*, const int32_t* rb, const int32_t* re, int64_t B,
                const int32_t* begins, const int32_t* ends, const uint8_t* chars,
                int32_t* out_begins, int32_t* out_ends, int32_t* out_ids, int64_t cap, int64_t* n_ids);
/* Diagnostics: number of merges whose two new neighbour pairs were both pushed with the same
 * rank (the only way two queue entries can tie), since create / last clear. */
int64_t orc_bpe_tie_events(const orc_bpe*);
void orc_bpe_clear_cache(orc_bpe*);

/* ---- WordpieceTokenizer : src/wordpiece_tokenizer.cpp:49-133 ---- */
typedef struct orc_wordpiece orc_wordpiece;
int orc_wordpiece_create(const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                         const char* suffix_indicator, int64_t si_len, int max_bytes_per_word,
                         orc_wordpiece** out);
void orc_wordpiece_destroy(orc_wordpiece*);
int orc_wordpiece_run(const orc_wordpiece*, const int32_t* rb, const int32_t* re, int64_t B,
                      const int32_t* begins, const int32_t* ends, const uint8_t* chars, int32_t unk_id,
                      int32_t* out_begins, int32_t* out_ends, int32_t* out_ids, int64_t cap, int64_t* n_ids);

/* ---- VocabEncoder : src/vocab_encoder.cpp:55-94 (values i32 or i64: elem_size 4 or 8) ---- */
typedef struct orc_vocab_encoder orc_vocab_encoder;
int orc_vocab_encoder_create(const int32_t* k_begins, const int32_t* k_ends, const uint8_t* k_chars,
                             const void* values, int64_t V, int elem_size, orc_vocab_encoder** out);
void orc_vocab_encoder_destroy(orc_vocab_encoder*);
int orc_vocab_encoder_run(const orc_vocab_encoder*, const int32_t* begins, const int32_t* ends,
                          const uint8_t* chars, int64_t N, const void* default_value, void* out);

/* ---- RaggedToDense : src/ragged_to_dense.cpp:70-174 ---- */
int orc_ragged_to_dense(const int32_t* begins, const int32_t* ends, int64_t B,
                        const void* data, int64_t n_data, int elem_size, int64_t inner_elems,
                        int32_t target_dim, const void* default_value,
                        int pad_right, int pad_max_length, void* out_dense, uint8_t* out_mask);

/* ---- TrieTokenizer : src/trie_tokenizer.cpp:23-81 ---- */
typedef struct orc_trie_tokenizer orc_trie_tokenizer;
int orc_trie_tokenizer_create(const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                              const int32_t* indices, orc_trie_tokenizer** out);
int orc_trie_tokenizer_run(const orc_trie_tokenizer* t, const int32_t* rb, const int32_t* re, int64_t B,
                           const int32_t* begins, const int32_t* ends, const uint8_t* chars, int32_t* out_begins,
                           int32_t* out_ends, int32_t* out_ids, int64_t cap, int64_t* n_ids);
void orc_trie_tokenizer_destroy(orc_trie_tokenizer* t);

/* ---- UTF8Validate : src/utf8_validate.cpp:18-143.  out_chars capacity: the reference allocates 3 * n_chars (:31-33);
 * offsets start at begins[0] (:46), *n_chars_out = last offset - begins[0] (:140). ---- */
int orc_utf8_validate(const int32_t* begins, const int32_t* ends, const uint8_t* chars, int64_t n, int replace_mode,
                      int32_t* out_begins, int32_t* out_ends, uint8_t* out_chars, int64_t cap, int64_t* n_chars_out);

/* ---- Truncate : src/truncate.cpp:37-150 (in place, like the reference); n_inputs 1 or 2, mode NULL for 1 ---- */
int orc_truncate(int n_inputs, int32_t* b0, int32_t* e0, int32_t* b1, int32_t* e1, int64_t n, int32_t max_length,
                 const char* side, const char* mode);

/* ---- CombineSegments : src/combine_segments.cpp:36-134 (i32 elements / ids; a 1-row segment is broadcast) ---- */
int orc_combine_segments(int n_segs, const int32_t* const* begins, const int32_t* const* ends,
                         const int32_t* const* data, const int64_t* n_rows, const int32_t* seg_ids, int64_t max_rows,
                         int32_t* out_begins, int32_t* out_ends, int32_t* out_data, int32_t* out_ids, int64_t cap,
                         int64_t* n_out);

/* ---- VocabDecoder : src/vocab_decoder.cpp:23-87 ---- */
int orc_vocab_decoder(const int32_t* ids, int64_t B, int64_t S,
                      const int32_t* v_begins, const int32_t* v_ends, const uint8_t* v_chars, int64_t V,
                      const int32_t* skip, int64_t n_skip,
                      int32_t* out_rb, int32_t* out_re, int32_t* out_begins, int32_t* out_ends,
                      uint8_t* out_chars, int64_t chars_cap, int64_t* n_chars);

/* ---- ByteFallback : src/byte_fallback.cpp:16-50, PieceToByte src/sentence_piece.cpp:27-46 ---- */
int orc_byte_fallback(const int32_t* begins, const int32_t* ends, const uint8_t* chars, int64_t N,
                      int32_t* out_begins, int32_t* out_ends, uint8_t* out_chars, int64_t* n_chars);

/* ---- FuzeRagged : src/fuze.cpp:20-40 ---- */
int orc_fuze(const int32_t* rb, const int32_t* re, int64_t B,
             const int32_t* begins, const int32_t* ends, int64_t N,
             int32_t* out_begins, int32_t* out_ends);

#ifdef __cplusplus
}
#endif
#endif
