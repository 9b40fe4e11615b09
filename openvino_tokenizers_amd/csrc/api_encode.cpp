// api_encode.cpp -- C-ABI entry points of RegexSplit, BPETokenizer and the fused encode path.
// Compiled as HIP (hipcc -x hip).  Reference behaviour replaced: src/regex_split.cpp:124-324,
// src/bpe_tokenizer.cpp:47-164 (evaluate) and :341-388 (table construction).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <cctype>
#include <string>
#include <vector>

#include <cstdlib>

#include "api_common.hpp"
#include "encode_kernels.hpp"
#include "span_kernel.hpp"
#include "runtime.hpp"
#include "tables.hpp"
#include "unicode_tables.inc"

using namespace ovtk;

namespace {

const char kGpt2Pattern[] = "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";
const char kGpt2DigitsPattern[] = "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";
// the split pattern of Llama-3's tokenizer.json (tiktoken cl100k family); reaches RegexSplit through hf_parser's Split step
const char kLlama3Pattern[] =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
// two more of that family (what separates them from Llama-3's: SplitDev::l3_digits1 / l3_tail_ws): Qwen2's tokenizer.json,
// and cl100k_base as tiktoken writes it (possessive quantifiers -- which change nothing where the repeated class and what
// follows it are disjoint, as in every place here --, `\s++$`, and `\s*[\r\n]` for `\s*[\r\n]+`: the greedy `\s*` in front leaves
// both at the run's last line break)
const char kQwen2Pattern[] =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
const char kCl100kPattern[] =
    "'(?i:[sdmt]|ll|ve|re)|[^\\r\\n\\p{L}\\p{N}]?+\\p{L}++|\\p{N}{1,3}+| ?[^\\s\\p{L}\\p{N}]++[\\r\\n]*+|\\s++$|\\s*[\\r\\n]|\\s+(?!\\S)|\\s+";
// Two patterns the compiled DFA runs for every caller but the fused encode, which has a scan for them (span_fam.hpp): o200k_base as
// tiktoken and the tokenizer.json files of its family write it, and the last Split of DeepSeek-V3's tokenizer.json
const char kO200kPattern[] =
    "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?|"
    "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]+[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?|\\p{N}{1,3}|"
    " ?[^\\s\\p{L}\\p{N}]+[\\r\\n/]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
const char kDeepSeekV3Pattern[] =
    "[!\"#$%&'()*+,\\-./:;<=>?@\\[\\\\\\]^_`{|}~][A-Za-z]+|[^\\r\\n\\p{L}\\p{P}\\p{S}]?[\\p{L}\\p{M}]+| ?[\\p{P}\\p{S}]+[\\r\\n]*|"
    "\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
// tokenizer_pipeline.py:392-426 (bert_whitespace_splitter / bert_keep_delimeters_splitter)
const char kBertWhitespacePattern[] = "\\s+";
const char kBertDelimitersPattern[] =
    "[!-/]|[:-@]|[\\[-`]|[{-~]|[\\p{P}]|[\\x{4E00}-\\x{9FFF}]|[\\x{3400}-\\x{4DBF}]|[\\x{20000}-\\x{2A6DF}]|"
    "[\\x{2A700}-\\x{2B73F}]|[\\x{2B740}-\\x{2B81F}]|[\\x{2B820}-\\x{2CEAF}]|[\\x{F900}-\\x{FAFF}]|[\\x{2F800}-\\x{2FA1F}]";

// Unicode property tables, one copy per device.
struct UnicodeTables {
    DevBuf index, blocks, flat, cls4;
};
int unicode_tables(int device, const uint16_t** index, const uint8_t** blocks, const uint8_t** flat = nullptr, const uint8_t** cls4 = nullptr) {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<UnicodeTables>> per_device;
    std::lock_guard<std::mutex> lk(mu);
    auto& t = per_device[device];
    if (!t) {
        auto fresh = std::make_unique<UnicodeTables>();
        if (int rc = fresh->index.upload(kUcIndex, sizeof kUcIndex)) return rc;
        if (int rc = fresh->blocks.upload(kUcBlocks, sizeof kUcBlocks)) return rc;
        // the class bits of planes 0 and 1 flat (SplitDev::uc_flat, span_l3.hpp kUcFlatLimit), from the same two-level table
        std::vector<uint8_t> f(0x20000 / 4, 0);
        for (uint32_t cp = 0; cp < 0x20000u; ++cp) {
            const uint32_t b = kUcBlocks[size_t(kUcIndex[cp >> 7]) * 64 + ((cp & 127) >> 1)];
            const uint32_t nib = (cp & 1) ? (b >> 4) : (b & 15u);
            f[cp >> 2] |= uint8_t((nib & 3u) << (2 * (cp & 3u)));
        }
        if (int rc = fresh->flat.upload(f.data(), f.size())) return rc;
        // four class bits per code point for the scans of span_fam.hpp (SplitDev::uc_cls4): General_Category folded to seven classes, white
        // space from the \s bit of the table above
        std::vector<uint8_t> gc;
        unicode_general_categories(gc);
        std::vector<uint8_t> c4(0x110000 / 2, 0);
        for (uint32_t cp = 0; cp < 0x110000u; ++cp) {
            const uint32_t b = kUcBlocks[size_t(kUcIndex[cp >> 7]) * 64 + ((cp & 127) >> 1)];
            const uint32_t nib = (cp & 1) ? (b >> 4) : (b & 15u);
            const uint32_t c = (nib & 3u) == kClsS ? kC4Space : uint32_t(kGcToC4[gc[cp]]);
            c4[cp >> 1] |= uint8_t(c << (4 * (cp & 1u)));
        }
        if (int rc = fresh->cls4.upload(c4.data(), c4.size())) return rc;
        OVTK_HIP(hipStreamSynchronize(nullptr));
        t = std::move(fresh);
    }
    *index = t->index.as<uint16_t>();
    *blocks = t->blocks.as<uint8_t>();
    if (flat) *flat = t->flat.as<uint8_t>();
    if (cls4) *cls4 = t->cls4.as<uint8_t>();
    return OVTK_OK;
}

}  // namespace


struct ovtk_bpe {
    int device = 0;
    BpeDev dev{};
    DevBuf root, edges, merges, new_id, bf, pieces, memo_room, store, store_room;
    int32_t store_capacity = 0;  // entries the piece store may take
    size_t memo_entries = 0;
    int32_t memo_capacity = 0;  // entries the device may add (memo_learn: cache_capacity, or as many as the store holds)
    bool narrow_ids = false;  // every token id < 65536: merge_kernel keeps ids as u16 in LDS
    bool stage16 = false;     // every token id < 65535: the staging entries of a call are u16 (0xFFFF = unused entry)
    // calls that still leave the piece store out: set to 32 after four calls in a row in which fewer than one probe in eight
    // hit (then the store is asked again, and so on -- text changes)
    mutable std::atomic<int> store_pause{0}, store_low{0};  // store_low: calls in a row with that little use of it
    // The short path (span_kernel.hpp): calls during which the kernels of the middle are still launched whatever the last call said --
    // lookup_kernel<kFused> for left-over rows, merge_kernel for pieces in neither table.  Set to kShortPathKeep by every call that had such
    // rows / pieces, counted down by every call that had none; a new handle's tables are empty: its first calls merge.
    mutable std::atomic<int> expect_pending{0}, expect_merge{kShortPathKeep};
    mutable std::atomic<int> last_unresolved{-1};   // pieces the last such call left for merge_kernel (sizes its launch)
};

namespace {
int run_encode(const ovtk_regex_split* split, const ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
               ovtk_ragged_i32_out* out, int mem, void* stream);

// The piece memo (tables.hpp PieceEntry): BPE(t) for every vocabulary token t used as a whole piece, computed by the
// device BPE itself -- the handle (still without memo) encodes its own vocabulary, one token per row -- plus, filled
// while encoding, pieces that take several tokens (how many: memo_learn, below).  It is the parallel-machine form of the
// reference's piece cache (bpe_tokenizer.cpp:197-205,331-338): the same pure function piece -> ids, so results never
// depend on what was encoded before; only the time does.
// memo_learn (ovtk_bpe_params): what the first level may learn.  < 0: exactly cache_capacity entries, the reference's count;
// 0: the larger of cache_capacity and the capacity of the handle's piece store -- the store holds what had to be merged once whatever
// cache_capacity says, and a piece of at most 15 bytes and kPieceMaxIds (6 with u16 ids) ids that it would hold is kept where
// the lookup kernels find it instead; > 0: that many.
int build_memo(ovtk_bpe* h, const ovtk_strings& vocab, int64_t cache_capacity, int64_t memo_store, int64_t memo_learn) {
    const int64_t V = vocab.n;
    if (V == 0) return OVTK_OK;
    const size_t nv = size_t(V);
    std::vector<int32_t> rb(nv), re(nv), ob(nv), oe(nv);
    for (int64_t i = 0; i < V; ++i) {
        rb[size_t(i)] = int32_t(i);
        re[size_t(i)] = int32_t(i + 1);
    }
    const int64_t cap = (vocab.n_chars + V) * (1 + h->dev.suffix_len);
    if (cap >= INT32_MAX) return OVTK_OK;  // absurdly large vocabulary text: run without memo
    std::vector<int32_t> ids(static_cast<size_t>(cap) + 1);
    ovtk_ragged_strings in{rb.data(), re.data(), V, vocab};
    ovtk_ragged_i32_out out{ob.data(), oe.data(), ids.data(), cap, 0, 0};
    if (int rc = run_encode(nullptr, h, &in, nullptr, &out, OVTK_MEM_HOST, nullptr)) return rc;
    PieceTableHost host;
    // (sized for the learned entries too, up to a couple per vocabulary token: a larger cache_capacity still counts, its entries
    // just compete for the buckets there are)
    // An entry's ids are six u16 when every id fits (PieceTableDev::packed6; lookup_span_kernel reads all six, the other lookup
    // kernels the entries of up to three), three i32 otherwise.
    const bool packed6 = h->stage16;
    build_piece_table(view_of(vocab), ob.data(), oe.data(), ids.data(), host,
                      size_t(std::min<int64_t>(std::max<int64_t>(cache_capacity, 0), 2 * V + 65536)), packed6);
    if (int rc = h->pieces.upload(host.slots.data(), host.slots.size() * sizeof(PieceEntry))) return rc;
    // The dynamic part (memo_insert in encode_kernels.hpp): up to `learn` further pieces, the ones the vocabulary
    // needs more than one token for, kept the first time merge_kernel computes them -- the reference's rule, its numbers.
    const int64_t store_cap = piece_store_capacity(V, resolve_memo_store(memo_store));
    const int64_t learn = memo_learn < 0 ? cache_capacity : (memo_learn > 0 ? memo_learn : std::max<int64_t>(cache_capacity, store_cap));
    // (never more than half the table's slots: direct-mapped, a piece whose slot is taken stays with the store)
    const int32_t room = int32_t(std::min<int64_t>({std::max<int64_t>(learn, 0), int64_t(host.slots.size() / 2), int64_t(INT32_MAX / 2)}));
    const uint32_t room_mask = room >= 4096 ? uint32_t(kRoomShards - 1) : 0u;   // (a handful of entries: one counter)
    std::vector<int32_t> rooms(size_t(kRoomShards) * kRoomStride, 0);
    for (uint32_t k = 0; k <= room_mask; ++k)
        rooms[size_t(k) * kRoomStride] = room / int32_t(room_mask + 1) + (int32_t(k) < room % int32_t(room_mask + 1) ? 1 : 0);
    if (int rc = h->memo_room.upload(rooms.data(), rooms.size() * sizeof(int32_t))) return rc;
    h->memo_capacity = room;
    OVTK_HIP(hipStreamSynchronize(nullptr));
    h->dev.pieces = PieceTableDev{h->pieces.as<PieceEntry>(), host.shift, h->memo_room.as<int32_t>(), room_mask, packed6 ? 1u : 0u};
    h->memo_entries = host.stored;
    // The second level (tables.hpp "piece store"): empty at create, filled by merge_kernel.
    if (int rc = alloc_piece_store(h->store, h->store_room, V, h->narrow_ids, h->dev.store, h->store_capacity, resolve_memo_store(memo_store))) return rc;
    return OVTK_OK;
}
}  // namespace

struct ovtk_special_tokens_split {
    int device = 0;
    SpecialDev dev{};
    DevBuf tb, te, tc, gf, gfl;
};

namespace {
// Parses the pattern SpecialTokensSplitStep generates (tokenizer_pipeline.py:138-159):
//   alt ( "|" alt )*,  alt = ["(?:\\s*)"] "(" token ( "|" token )* ")" ["(?:\\s*)"],  token = quote_meta(text).
// Returns false for anything else.
bool parse_special_pattern(const std::string& pat, std::vector<std::string>& tokens, std::vector<int32_t>& group_first,
                           std::vector<uint8_t>& group_flags) {
    static const std::string kStrip = "(?:\\s*)";
    size_t i = 0;
    group_first.clear();
    group_flags.clear();
    tokens.clear();
    if (pat.empty()) return false;
    while (true) {
        uint8_t flags = 0;
        if (pat.compare(i, kStrip.size(), kStrip) == 0) { flags |= 1; i += kStrip.size(); }
        if (i >= pat.size() || pat[i] != '(' || pat.compare(i, 2, "(?") == 0) return false;
        ++i;
        group_first.push_back(int32_t(tokens.size()));
        std::string tok;
        bool closed = false;
        while (i < pat.size()) {
            const char c = pat[i];
            if (c == '\\') {
                if (i + 1 >= pat.size()) return false;
                const unsigned char n = static_cast<unsigned char>(pat[i + 1]);
                if (n < 0x80 && std::isalnum(n)) return false;  // an escape sequence, not a quoted literal
                tok.push_back(char(n));
                i += 2;
            } else if (c == '|' || c == ')') {
                if (tok.empty()) return false;
                tokens.push_back(tok);
                tok.clear();
                ++i;
                if (c == ')') { closed = true; break; }
            } else if (c == '(' || c == '[' || c == '*' || c == '+' || c == '?' || c == '.' || c == '^' || c == '$' || c == '{') {
                return false;  // quote_meta would have escaped it: not a literal token list
            } else {
                tok.push_back(c);
                ++i;
            }
        }
        if (!closed) return false;
        if (pat.compare(i, kStrip.size(), kStrip) == 0) { flags |= 2; i += kStrip.size(); }
        group_flags.push_back(flags);
        if (i == pat.size()) break;
        if (pat[i] != '|') return false;
        ++i;
    }
    group_first.push_back(int32_t(tokens.size()));
    return true;
}
}  // namespace

extern "C" {

const char* ovtk_last_error(void) { return last_error(); }
int ovtk_abi_version(void) { return 1002; }   // (1001: ovtk_bpe_params / ovtk_wordpiece_params::memo_store, the dense / special encode calls; 1002: ovtk_bpe_params::memo_learn)

const char* ovtk_device_name(void) {
    static std::string name;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return nullptr;
    name = std::string(prop.gcnArchName) + " " + prop.name;
    return name.c_str();
}

void ovtk_profile_enable(int on) { Profiler::get().enable(on != 0); }
void ovtk_profile_reset(void) { Profiler::get().reset(); }
int ovtk_profile_get(const char* kernel, double* total_ms, int64_t* launches) {
    return Profiler::get().lookup(kernel, total_ms, launches) ? 0 : -1;
}
int64_t ovtk_profile_dump(char* buf, int64_t cap) {
    const std::string s = Profiler::get().dump();
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(size_t(cap) - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return int64_t(s.size()) + 1;
}

// ------------------------------------------------------------------------------- RegexSplit
int ovtk_regex_split_create(const ovtk_regex_split_params* p, ovtk_regex_split** out) {
    if (!p || !out || !p->pattern || !p->behaviour) return set_error(OVTK_E_ARG, "regex_split: null argument");
    auto h = std::make_unique<ovtk_regex_split>();
    const std::string beh(p->behaviour), pat(p->pattern, p->pattern + p->pattern_len);
    // regex_split.cpp:16-22,113-117
    if (beh == "remove") h->mode = 0;
    else if (beh == "isolate" || beh == "contiguous") h->mode = 1;
    else if (beh == "mergedwithprevious") h->mode = 2;
    else if (beh == "mergedwithnext") h->mode = 3;
    else return set_error(OVTK_E_ARG, "RegexSplit doesn't support unknown split mode: " + beh);
    if (!(p->max_splits == -1 || p->max_splits > 0))
        return set_error(OVTK_E_ARG, "RegexSplit max_splits attribute must be greater then `0` or equal to `-1`");
    h->invert = p->invert != 0;
    h->max_splits = p->max_splits;
    // regex_split.cpp:33-37: "contiguous" is "isolate" on (pattern)+, unless the pattern already ends in '+'
    std::string eff = pat;
    if (beh == "contiguous" && (pat.empty() || pat.back() != '+')) eff = "(" + pat + ")+";
    // Hand-written scanners for the patterns the converter emits, where the behaviour is one they implement; every
    // other pattern / behaviour runs the compiled DFA (regex_compile.cpp).
    h->dev.kind = kSplitGeneral;
    if (eff == kGpt2Pattern && h->mode == 1) h->dev.kind = kSplitGpt2;
    else if (eff == kGpt2DigitsPattern && h->mode == 1) h->dev.kind = kSplitGpt2Digits;
    else if (eff == kLlama3Pattern && h->mode == 1) h->dev.kind = kSplitLlama3;
    else if (eff == kQwen2Pattern && h->mode == 1) { h->dev.kind = kSplitLlama3; h->dev.l3_digits1 = 1; }
    else if (eff == kCl100kPattern && h->mode == 1) { h->dev.kind = kSplitLlama3; h->dev.l3_tail_ws = 1; }
    else if (eff == kBertWhitespacePattern && h->mode <= 1) h->dev.kind = kSplitWhitespace;
    else if (eff == kBertDelimitersPattern && h->mode <= 1) h->dev.kind = kSplitBertPunct;
    if (h->dev.kind == kSplitGeneral && h->mode == 1 && eff == kO200kPattern) h->dev.family = kFamO200k;
    if (h->dev.kind == kSplitGeneral && h->mode == 1 && eff == kDeepSeekV3Pattern) h->dev.family = kFamDs3;
    if (h->dev.kind >= kSplitWhitespace && h->dev.kind != kSplitGeneral)
        // regex_split.cpp:244-284: "remove" drops the pieces flagged `invert`: the matches, or the gaps when invert is set
        h->dev.drop = h->mode == 0 ? (h->invert ? 2 : 1) : 0;
    RegexProgram prog;
    if (h->dev.kind == kSplitGeneral) {
        std::string err;
        if (int rc = compile_regex(eff, prog, err)) return set_error(rc, err);
    }
    if (int rc = use_device(p->device)) return rc;
    h->device = p->device;
    if (int rc = unicode_tables(p->device, &h->dev.uc_index, &h->dev.uc_blocks, &h->dev.uc_flat, &h->dev.uc_cls4)) return rc;
    if (h->dev.kind == kSplitGeneral) {
        int e = 0;
        e = e ? e : h->r_trans.upload(prog.trans.data(), prog.trans.size() * sizeof(uint16_t));
        e = e ? e : h->r_ascii.upload(prog.ascii_class, sizeof prog.ascii_class);
        e = e ? e : h->r_index.upload(prog.cp_index.data(), prog.cp_index.size() * sizeof(uint16_t));
        e = e ? e : h->r_blocks.upload(prog.cp_blocks.data(), prog.cp_blocks.size());
        e = e ? e : h->r_ctx.upload(prog.ctx_next.data(), std::max<size_t>(prog.ctx_next.size(), 1));
        if (e) return e;
        OVTK_HIP(hipStreamSynchronize(nullptr));
        RegexDev& r = h->regex;
        r.trans = h->r_trans.as<uint16_t>();
        r.ascii_class = h->r_ascii.as<uint8_t>();
        r.cp_index = h->r_index.as<uint16_t>();
        r.cp_blocks = h->r_blocks.as<uint8_t>();
        r.ctx_next = h->r_ctx.as<uint8_t>();
        r.n_syms = prog.n_syms;
        r.n_states = prog.n_states;
        r.sym_eot = prog.sym_eot;
        r.sym_final_nl = prog.sym_final_nl;
        r.n_ctx = prog.n_ctx;
        r.behind_chars = prog.behind_chars;
        r.cp_blocks_bytes = int32_t(prog.cp_blocks.size());
        std::memcpy(r.start, prog.start, sizeof r.start);
        r.mode = h->mode;
        r.invert = h->invert ? 1 : 0;
        r.max_splits = h->max_splits;
    }
    *out = h.release();
    return OVTK_OK;
}

void ovtk_regex_split_destroy(ovtk_regex_split* h) { delete h; }

// ------------------------------------------------------------------------------- SpecialTokensSplit
int ovtk_special_tokens_split_create(const char* pattern, int64_t pattern_len, int device, ovtk_special_tokens_split** out) {
    if (!pattern || !out || pattern_len < 0) return set_error(OVTK_E_ARG, "special_tokens_split: null argument");
    std::vector<std::string> tokens;
    std::vector<int32_t> group_first;
    std::vector<uint8_t> group_flags;
    if (!parse_special_pattern(std::string(pattern, pattern + pattern_len), tokens, group_first, group_flags))
        return set_error(OVTK_E_UNSUPPORTED,
                         "SpecialTokensSplit: the pattern is not a list of quoted special tokens as "
                         "tokenizer_pipeline.py:138-159 generates it; PCRE2 is not executed on the device");
    if (int rc = use_device(device)) return rc;
    auto h = std::make_unique<ovtk_special_tokens_split>();
    h->device = device;
    std::vector<int32_t> tb, te;
    std::string chars;
    bool any_strip_left = false;
    for (uint8_t f : group_flags) any_strip_left = any_strip_left || (f & 1);
    for (const auto& t : tokens) {
        tb.push_back(int32_t(chars.size()));
        chars += t;
        te.push_back(int32_t(chars.size()));
        chars.append((16 - chars.size() % 16) % 16 + 16, '\0');   // (token_at compares 16 bytes at a time: SpecialDev::tok_padded)
        const uint8_t b = uint8_t(t[0]);
        h->dev.first_bytes[b >> 5] |= 1u << (b & 31);
    }
    if (any_strip_left)  // a match may start with \\s (PCRE2_UCP): tab..CR, space, and the lead bytes of U+0085, U+00A0, U+1680,
                         // U+2000-200A / 2028 / 2029 / 202F / 205F, U+3000
        for (uint8_t b : {9, 10, 11, 12, 13, 32, 0xC2, 0xE1, 0xE2, 0xE3}) h->dev.first_bytes[b >> 5] |= 1u << (b & 31);
    h->dev.tok_padded = 1;
    h->dev.any_strip_left = any_strip_left ? 1 : 0;
    {   // the two byte sets as lists (split_seq_device.hpp: the 16-bytes-at-a-time filters)
        std::vector<uint8_t> tf, all;
        for (const auto& t : tokens)
            if (std::find(tf.begin(), tf.end(), uint8_t(t[0])) == tf.end()) tf.push_back(uint8_t(t[0]));
        for (int b = 0; b < 256; ++b)
            if ((h->dev.first_bytes[b >> 5] >> (b & 31)) & 1u) all.push_back(uint8_t(b));
        h->dev.n_tok_first = tf.size() <= sizeof h->dev.tok_first ? int32_t(tf.size()) : -1;
        for (size_t i = 0; h->dev.n_tok_first > 0 && i < tf.size(); ++i) h->dev.tok_first[i] = tf[i];
        h->dev.n_first = all.size() <= sizeof h->dev.first ? int32_t(all.size()) : -1;
        for (size_t i = 0; h->dev.n_first > 0 && i < all.size(); ++i) h->dev.first[i] = all[i];
    }
    int e = 0;
    e = e ? e : h->tb.upload(tb.data(), tb.size() * 4);
    e = e ? e : h->te.upload(te.data(), te.size() * 4);
    e = e ? e : h->tc.upload(chars.data(), chars.size());
    e = e ? e : h->gf.upload(group_first.data(), group_first.size() * 4);
    e = e ? e : h->gfl.upload(group_flags.data(), group_flags.size());
    if (e) return e;
    OVTK_HIP(hipStreamSynchronize(nullptr));
    h->dev.tok_begins = h->tb.as<int32_t>();
    h->dev.tok_ends = h->te.as<int32_t>();
    h->dev.tok_chars = h->tc.as<uint8_t>();
    h->dev.group_first = h->gf.as<int32_t>();
    h->dev.group_flags = h->gfl.as<uint8_t>();
    h->dev.n_groups = int32_t(group_flags.size());
    h->dev.n_tokens = int32_t(tb.size());
    h->dev.n_tok_chars = int32_t(chars.size());
    h->dev.uc.kind = kSplitWhitespace;
    if (int rc = unicode_tables(device, &h->dev.uc.uc_index, &h->dev.uc.uc_blocks)) return rc;
    *out = h.release();
    return OVTK_OK;
}

void ovtk_special_tokens_split_destroy(ovtk_special_tokens_split* h) { delete h; }

// ------------------------------------------------------------------------------- BPETokenizer
int ovtk_bpe_create(const ovtk_bpe_params* p, ovtk_bpe** out) {
    if (!p || !out) return set_error(OVTK_E_ARG, "bpe: null argument");
    if (int rc = use_device(p->device)) return rc;
    BpeHost host;
    std::string err;
    const StringsView right = view_of(p->merges_right);
    const int rc = build_bpe(view_of(p->vocab), view_of(p->merges), p->merges_right.begins ? &right : nullptr,
                             view_of(p->added_tokens), p->added_ids,
                             std::string(p->unk_token ? p->unk_token : "", size_t(p->unk_token ? p->unk_token_len : 0)),
                             std::string(p->end_suffix ? p->end_suffix : "", size_t(p->end_suffix ? p->end_suffix_len : 0)),
                             p->byte_fallback != 0, host, err);
    if (rc) return set_error(rc, err);
    auto h = std::make_unique<ovtk_bpe>();
    h->device = p->device;
    h->narrow_ids = p->vocab.n <= 65536;
    for (int64_t i = 0; i < p->added_tokens.n && p->added_ids; ++i)
        if (p->added_ids[i] < 0 || p->added_ids[i] > 65535) h->narrow_ids = false;
    h->stage16 = h->narrow_ids && p->vocab.n <= 65535;
    for (int64_t i = 0; i < p->added_tokens.n && p->added_ids; ++i)
        if (p->added_ids[i] > 65534) h->stage16 = false;
    int e = 0;
    e = e ? e : h->root.upload(host.trie.root.data(), host.trie.root.size() * sizeof(I2));
    e = e ? e : h->edges.upload(host.trie.edges.data(), host.trie.edges.size() * sizeof(TrieEdge));
    e = e ? e : h->merges.upload(host.merges.data(), host.merges.size() * sizeof(MergeBucket));
    e = e ? e : h->new_id.upload(host.new_id.data(), host.new_id.size() * sizeof(int32_t));
    e = e ? e : h->bf.upload(host.byte_fallback_id.data(), host.byte_fallback_id.size() * sizeof(int32_t));
    if (e) return e;
    OVTK_HIP(hipStreamSynchronize(nullptr));
    BpeDev& d = h->dev;
    d.trie.root = h->root.as<I2>();
    d.trie.edges = h->edges.as<TrieEdge>();
    d.trie.edge_mask = host.trie.edge_mask;
    d.trie.edge_shift = host.trie.edge_shift;
    d.merges = h->merges.as<MergeBucket>();
    d.bucket_shift = host.bucket_shift;
    d.pieces = PieceTableDev{nullptr, 30, nullptr, 0, 0};
    d.store = PieceStoreDev{nullptr, 30, nullptr, 0};
    d.new_id = h->new_id.as<int32_t>();
    d.byte_fallback_id = h->bf.as<int32_t>();
    d.unk_id = host.unk_id;
    d.suffix_len = int32_t(host.suffix.size());
    std::memset(d.suffix, 0, sizeof d.suffix);
    std::memcpy(d.suffix, host.suffix.data(), host.suffix.size());
    // cache_capacity == 0 disables the reference's piece cache (bpe_tokenizer.cpp:331: size() < capacity); here it
    // disables the memo the same way.  Results are identical either way.
    if (p->cache_capacity != 0)
        if (int rc = build_memo(h.get(), p->vocab, p->cache_capacity, p->memo_store, p->memo_learn)) return rc;
    *out = h.release();
    return OVTK_OK;
}

void ovtk_bpe_destroy(ovtk_bpe* h) { delete h; }

int ovtk_bpe_memo_entries(ovtk_bpe* h, int64_t* fixed, int64_t* learned) {
    if (!h || !fixed || !learned) return set_error(OVTK_E_ARG, "null argument");
    *fixed = int64_t(h->memo_entries);
    *learned = 0;
    if (!h->dev.pieces.room) return OVTK_OK;
    if (int rc = use_device(h->device)) return rc;
    std::vector<int32_t> rooms(size_t(kRoomShards) * kRoomStride, 0);
    OVTK_HIP(hipDeviceSynchronize());
    OVTK_HIP(hipMemcpy(rooms.data(), h->dev.pieces.room, rooms.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    int64_t left = 0;
    for (uint32_t k = 0; k <= h->dev.pieces.room_mask; ++k) left += std::max<int32_t>(rooms[size_t(k) * kRoomStride], 0);
    *learned = h->memo_capacity - left;
    return OVTK_OK;
}

int ovtk_set_memo_store(int64_t entries) {
    if (entries < 0 || entries > (int64_t(1) << 22)) return set_error(OVTK_E_ARG, "memo store: 0 (off) .. 4194304 entries");
    memo_store_entries().store(entries, std::memory_order_relaxed);
    return OVTK_OK;
}

int ovtk_bpe_store_entries(ovtk_bpe* h, int64_t* stored, int64_t* capacity) {
    if (!h || !stored || !capacity) return set_error(OVTK_E_ARG, "null argument");
    *stored = 0;
    *capacity = h->store_capacity;
    if (!h->dev.store.room) return OVTK_OK;
    if (int rc = use_device(h->device)) return rc;
    int32_t room = 0;
    OVTK_HIP(hipDeviceSynchronize());
    OVTK_HIP(hipMemcpy(&room, h->dev.store.room, sizeof room, hipMemcpyDeviceToHost));
    *stored = int64_t(h->store_capacity) - room;  // (room may end a few below zero: every wave reads it once per launch)
    return OVTK_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------- run pipelines
namespace {

// RegexSplit on device-resident rows: validation, count pass, offsets, write pass; waits for the piece count.
// d_* are device buffers (d_b / d_e / d_sk: `capacity` entries; d_sk may be null).
int split_on_device(const ovtk_regex_split* h, Workspace& ws, const RowsIn& d_in, hipStream_t s, int32_t* d_rb, int32_t* d_re,
                    int32_t* d_b, int32_t* d_e, uint8_t* d_sk, int64_t capacity, int64_t* n_out) {
    const int n_rows = d_in.n_rows;
    const int grid = grid_lookup(h->device, n_rows);
    const int n_tiles = (n_rows + kRowTile - 1) / kRowTile;
    int e = 0;
    e = e ? e : ws.row_cnt.ensure(size_t(n_rows) * 4);
    e = e ? e : ws.wave_off.ensure(size_t(grid * kWavesPerBlock + 1) * sizeof(long long));
    e = e ? e : ws.tiles.ensure(size_t(n_tiles + 1) * sizeof(long long));
    e = e ? e : ws.status.ensure(sizeof(RunStatus));
    if (e) return e;
    EncodeWork w{};
    w.n_waves = grid * kWavesPerBlock;
    w.wave_off = ws.wave_off.as<long long>();
    w.row_cnt = ws.row_cnt.as<int32_t>();
    w.tile_off = ws.tiles.as<long long>();
    w.stage_cap = INT32_MAX;
    w.status = ws.status.as<RunStatus>();
    OVTK_HIP(hipMemsetAsync(w.status, 0, sizeof(RunStatus), s));
    // range validation of the inputs (the staging arenas it also computes are not needed here)
    OVTK_LAUNCH(ws.marks, "prep_rows", prep_rows_kernel, std::min(grid, kTicketBlocks), kBlockThreads, s, d_in, 1, w);
    const int lane_grid = (n_rows + kBlockThreads - 1) / kBlockThreads;  // the DFA runs one lane per row
    int32_t* const nil = nullptr;
    if (h->dev.kind == kSplitGeneral)
        OVTK_LAUNCH(ws.marks, "regex_count", regex_split_kernel<0>, lane_grid, kBlockThreads, s, d_in, h->regex, w, nil, nil, nil, nil,
                    (uint8_t*)nullptr);
    else if (h->dev.kind == kSplitLlama3)
        OVTK_LAUNCH(ws.marks, "split_count", (split_kernel<0, true>), grid, kBlockThreads, s, d_in, h->dev, h->max_splits, w, nil, nil,
                    nil, nil, (uint8_t*)nullptr);
    else
        OVTK_LAUNCH(ws.marks, "split_count", split_kernel<0>, grid, kBlockThreads, s, d_in, h->dev, h->max_splits, w, nil, nil, nil, nil,
                    (uint8_t*)nullptr);
    OVTK_LAUNCH(ws.marks, "count_scan", count_scan_kernel, std::min((n_tiles + 3) / 4, kTicketBlocks), kBlockThreads, s, n_rows, w,
                (long long)capacity);
    if (h->dev.kind == kSplitGeneral)
        OVTK_LAUNCH(ws.marks, "regex_write", regex_split_kernel<1>, lane_grid, kBlockThreads, s, d_in, h->regex, w, d_rb, d_re, d_b, d_e,
                    d_sk);
    else if (h->dev.kind == kSplitLlama3)
        OVTK_LAUNCH(ws.marks, "split_write", (split_kernel<1, true>), grid, kBlockThreads, s, d_in, h->dev, h->max_splits, w, d_rb, d_re,
                    d_b, d_e, d_sk);
    else
        OVTK_LAUNCH(ws.marks, "split_write", split_kernel<1>, grid, kBlockThreads, s, d_in, h->dev, h->max_splits, w, d_rb, d_re, d_b,
                    d_e, d_sk);
    if (int rc = finish_status(ws, s)) return rc;
    const RunStatus& st = *ws.host_status;
    if (st.flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside their tensors");
    if (st.flags & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "RegexSplit: output begins/ends too small");
    *n_out = st.n_out;
    return OVTK_OK;
}

// The split runs inside lookup_kernel (scanner -> pieces -> memo probe in one pass over the text).
bool fusable(const ovtk_regex_split* split, const ovtk_bpe* bpe = nullptr) {
    if (split->max_splits != -1) return false;
    if (split->dev.kind <= kSplitGpt2Digits || split->dev.kind == kSplitLlama3) return true;
    // the families of span_fam.hpp: lookup_span_kernel scans them -- it probes the memo for every piece, a handle without one takes the
    // compiled DFA like any other pattern
    return split->dev.family != kFamNone && bpe && bpe->dev.pieces.slots;
}

// The op's dense outputs from special_sparse_kernel's layout (round 6): a row's strings stand at [rb, re) of the sparse buffers -- entry `row`
// for the rows of at most one string, behind the n_rows entries for the others --; the scan of the rows' counts is the reference's running
// offset (special_tokens_split.cpp:100-150), a lane per row moves its strings there.
struct SparseRowLen {
    const int32_t* rb;
    const int32_t* re;
    RunStatus* status;   // regex_sparse_kernel marks a row of bad offsets with a begin of -1: the range error is raised here (nullptr: no such rows)
    __device__ long long operator()(long long i) const {
        if (rb[i] < 0) {
            if (status) atomicOr(&status->flags, kFlagRange);
            return 0;
        }
        return re[i] - rb[i];
    }
};
struct SparseRowGather {
    const int32_t* rb;
    const int32_t* b;
    const int32_t* e;
    const uint8_t* sk;
    int32_t *o_rb, *o_re, *o_b, *o_e;
    uint8_t* o_sk;
    __device__ void operator()(long long i, long long off, long long len) const {
        o_rb[i] = int32_t(off);
        o_re[i] = int32_t(off + len);
        const int src = rb[i];
        for (long long k = 0; k < len; ++k) {
            o_b[off + k] = b[src + k];
            o_e[off + k] = e[src + k];
            if (o_sk) o_sk[off + k] = sk[src + k];
        }
    }
};
// ... the same for rows of many strings (RegexSplit: a row's pieces): the scan files the offsets, a WAVE per row moves the strings
struct SparseRowOffsets {
    int32_t *o_rb, *o_re;
    __device__ void operator()(long long i, long long off, long long len) const {
        o_rb[i] = int32_t(off);
        o_re[i] = int32_t(off + len);
    }
};
static __global__ __launch_bounds__(kBlockThreads) void sparse_gather_kernel(int n_rows, const int32_t* rb, const int32_t* b, const int32_t* e, const uint8_t* sk,
                                                                             const int32_t* o_rb, const int32_t* o_re, int32_t* o_b, int32_t* o_e, uint8_t* o_sk,
                                                                             const RunStatus* status, uint32_t skip_flags) {
    if (status->flags & skip_flags) return;
    const int l = lane_id();
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int row = int(blockIdx.x) * kWavesPerBlock + wave_in_block(); row < n_rows; row += n_waves) {
        const int src = rb[row], off = o_rb[row], len = o_re[row] - off;
        for (int k = l; k < len; k += kWave) {
            o_b[off + k] = b[src + k];
            o_e[off + k] = e[src + k];
            if (o_sk) o_sk[off + k] = sk[src + k];
        }
    }
}
struct SparseFin {   // total -> status->n_out (special_sparse_kernel used the word as its region counter) + capacity flag
    RunStatus* status;
    long long cap;
    __device__ void operator()(long long total) const {
        status->n_out = total > INT32_MAX ? INT32_MAX : int32_t(total);
        if (total > cap) atomicOr(&status->flags, kFlagOutCapacity);
    }
};
// RegexSplit, the op alone, for a COMPILED pattern (the DFA): regex_sparse_kernel -- the fused encode's one-pass form, a lane per row, a
// character per turn -- into buffers of the reference's capacity, the scan of the rows' counts, the gather.  Until round 6 the automaton ran
// twice (regex_split_kernel<0 / 1>, split_on_device).  The hand-written scanners keep their count and write passes.
int regex_one_pass(const ovtk_regex_split* h, Workspace& sw, const RowsIn& d_in, hipStream_t s, int32_t* d_rb, int32_t* d_re, int32_t* d_b,
                   int32_t* d_e, uint8_t* d_sk, long long capacity) {
    const int n_rows = d_in.n_rows;
    const long long cap = (long long)d_in.n_chars + d_in.n_strings;   // src/regex_split.cpp:182
    int e = 0;
    e = e ? e : sw.gen[0].ensure(size_t(n_rows) * 4);
    e = e ? e : sw.gen[1].ensure(size_t(n_rows) * 4);
    e = e ? e : sw.gen[2].ensure(size_t(cap) * 4);
    e = e ? e : sw.gen[3].ensure(size_t(cap) * 4);
    e = e ? e : sw.gen[4].ensure(size_t(cap));
    e = e ? e : sw.tiles.ensure(scan_tiles_bytes(n_rows));
    e = e ? e : sw.status.ensure(sizeof(RunStatus));
    if (e) return e;
    RunStatus* st = sw.status.as<RunStatus>();
    OVTK_HIP(hipMemsetAsync(st, 0, sizeof(RunStatus), s));
    int32_t *t_rb = sw.gen[0].as<int32_t>(), *t_re = sw.gen[1].as<int32_t>(), *t_b = sw.gen[2].as<int32_t>(), *t_e = sw.gen[3].as<int32_t>();
    uint8_t* t_sk = sw.gen[4].as<uint8_t>();
    const int lane_grid = (n_rows + kBlockThreads - 1) / kBlockThreads;
    const RegexSparseLds lay = regex_sparse_layout(h->regex.n_states * h->regex.n_syms, h->regex.cp_blocks_bytes);
    note_launch(sw.marks, s);
    Profiler& pf = Profiler::get();
    if (pf.enabled()) pf.begin("regex_split", s, sw.marks);
    if (h->mode == 1 && h->max_splits == -1)
        hipLaunchKernelGGL(regex_sparse_kernel<true>, dim3(lane_grid), dim3(kBlockThreads), size_t(lay.total), s, d_in, h->regex, st, cap, t_rb, t_re, t_b, t_e,
                           t_sk);
    else
        hipLaunchKernelGGL(regex_sparse_kernel<false>, dim3(lane_grid), dim3(kBlockThreads), size_t(lay.total), s, d_in, h->regex, st, cap, t_rb, t_re, t_b, t_e,
                           t_sk);
    if (pf.enabled()) pf.end(s, sw.marks);
    // (a row of bad offsets has begin -1 / end 0 there: SparseBadRows turns that into the range error the count pass used to raise)
    launch_scan(sw.marks, "regex_split", s, (long long)n_rows, SparseRowLen{t_rb, t_re, st}, SparseRowOffsets{d_rb, d_re}, SparseFin{st, capacity},
                sw.tiles.as<long long>(), st, kFlagOutCapacity | kFlagRange);
    OVTK_LAUNCH(sw.marks, "regex_split", sparse_gather_kernel, grid_lookup(h->device, n_rows), kBlockThreads, s, n_rows, (const int32_t*)t_rb, (const int32_t*)t_b,
                (const int32_t*)t_e, (const uint8_t*)t_sk, (const int32_t*)d_rb, (const int32_t*)d_re, d_b, d_e, d_sk, (const RunStatus*)st,
                uint32_t(kFlagOutCapacity | kFlagRange));
    return OVTK_OK;
}

// RegexSplit, the op alone, with a hand-written scanner, in one pass: the rows' bounds (a slot per byte and one per string) -> scan: their regions;
// split_kernel<2> writes there and files the counts; scan of the counts; a wave per row gathers.  (Count pass 79 us + write pass 98 us
// became write pass + gather: 0.25 -> 0.19 ms per config-2 batch.)
struct RowBound {
    RowsIn in;
    RunStatus* status;
    __device__ long long operator()(long long i) const {
        const long long cb = in.ragged_begins[i], ce = in.ragged_ends[i];
        bool bad = cb < ce && (cb < 0 || ce > in.n_strings);
        long long sum = 0;
        for (long long col = cb; col < ce && !bad; ++col) {
            const long long b = in.begins[col], e = in.ends[col];
            if (b < 0 || e < b || e > in.n_chars) bad = true;
            sum += e - b + 1;
        }
        if (bad) {
            atomicOr(&status->flags, kFlagRange);
            return 0;
        }
        return sum;
    }
};
struct RowRegion {
    int32_t* region;
    __device__ void operator()(long long i, long long off, long long) const { region[i] = int32_t(off); }
};
struct RegionFin {   // (strings that overlap ask for more than the reference's capacity: the count and write passes take over)
    RunStatus* status;
    long long cap;
    __device__ void operator()(long long total) const {
        if (total > cap) atomicOr(&status->flags, kFlagOutCapacity);
    }
};
struct RowCount {
    const int32_t* cnt;
    __device__ long long operator()(long long i) const { return cnt[i]; }
};
int split_one_pass(const ovtk_regex_split* h, Workspace& sw, const RowsIn& d_in, hipStream_t s, int32_t* d_rb, int32_t* d_re, int32_t* d_b,
                   int32_t* d_e, uint8_t* d_sk, long long capacity) {
    const int n_rows = d_in.n_rows;
    const int grid = grid_lookup(h->device, n_rows);
    const long long cap = (long long)d_in.n_chars + d_in.n_strings;   // src/regex_split.cpp:182
    int e = 0;
    e = e ? e : sw.row_cnt.ensure(size_t(n_rows) * 4);
    e = e ? e : sw.row_stage.ensure(size_t(n_rows) * 4);
    e = e ? e : sw.gen[2].ensure(size_t(cap) * 4);
    e = e ? e : sw.gen[3].ensure(size_t(cap) * 4);
    e = e ? e : sw.gen[4].ensure(size_t(cap));
    e = e ? e : sw.tiles.ensure(scan_tiles_bytes(n_rows));
    e = e ? e : sw.status.ensure(sizeof(RunStatus));
    if (e) return e;
    RunStatus* st = sw.status.as<RunStatus>();
    OVTK_HIP(hipMemsetAsync(st, 0, sizeof(RunStatus), s));
    EncodeWork w{};
    w.n_waves = grid * kWavesPerBlock;
    w.row_cnt = sw.row_cnt.as<int32_t>();
    w.row_stage = sw.row_stage.as<int32_t>();
    w.stage_cap = INT32_MAX;
    w.status = st;
    int32_t *t_b = sw.gen[2].as<int32_t>(), *t_e = sw.gen[3].as<int32_t>();
    uint8_t* t_sk = sw.gen[4].as<uint8_t>();
    launch_scan(sw.marks, "split_regions", s, (long long)n_rows, RowBound{d_in, st}, RowRegion{w.row_stage}, RegionFin{st, cap}, sw.tiles.as<long long>(), st,
                kFlagOutCapacity | kFlagRange);
    int32_t* const nil = nullptr;
    if (h->dev.kind == kSplitLlama3)
        OVTK_LAUNCH(sw.marks, "split_write", (split_kernel<2, true>), grid, kBlockThreads, s, d_in, h->dev, h->max_splits, w, nil, nil, t_b, t_e, t_sk);
    else
        OVTK_LAUNCH(sw.marks, "split_write", split_kernel<2>, grid, kBlockThreads, s, d_in, h->dev, h->max_splits, w, nil, nil, t_b, t_e, t_sk);
    launch_scan(sw.marks, "split_offsets", s, (long long)n_rows, RowCount{w.row_cnt}, SparseRowOffsets{d_rb, d_re}, SparseFin{st, capacity},
                sw.tiles.as<long long>(), st, kFlagOutCapacity | kFlagRange);
    OVTK_LAUNCH(sw.marks, "split_gather", sparse_gather_kernel, grid, kBlockThreads, s, n_rows, (const int32_t*)w.row_stage, (const int32_t*)t_b,
                (const int32_t*)t_e, (const uint8_t*)t_sk, (const int32_t*)d_rb, (const int32_t*)d_re, d_b, d_e, d_sk, (const RunStatus*)st,
                uint32_t(kFlagOutCapacity | kFlagRange));
    return OVTK_OK;
}

// SpecialTokensSplit, the op alone, in ONE pass over the text: special_sparse_kernel (the fused encode's: a sweep of the text for the tokens'
// first bytes, only the rows in which one turns up are walked) into buffers of the reference's capacity, the scan of the rows' counts, the
// gather.  Until round 6: count pass, scan, write pass, a lane walking each row twice (special_on_device below: 0.24 ms for a config-2 batch).
int special_one_pass(const ovtk_special_tokens_split* h, Workspace& sw, const RowsIn& d_in, hipStream_t s, int32_t* d_rb, int32_t* d_re,
                     int32_t* d_b, int32_t* d_e, uint8_t* d_sk, long long capacity) {
    const int n_rows = d_in.n_rows;
    const long long cap_extra = (long long)d_in.n_chars + d_in.n_strings;   // special_tokens_split.cpp:88-92, + skipped empty strings
    const long long cap = n_rows + cap_extra;
    int e = 0;
    e = e ? e : sw.gen[0].ensure(size_t(n_rows) * 4);
    e = e ? e : sw.gen[1].ensure(size_t(n_rows) * 4);
    e = e ? e : sw.gen[2].ensure(size_t(cap) * 4);
    e = e ? e : sw.gen[3].ensure(size_t(cap) * 4);
    e = e ? e : sw.gen[4].ensure(size_t(cap));
    e = e ? e : sw.tiles.ensure(scan_tiles_bytes(n_rows));
    e = e ? e : sw.status.ensure(sizeof(RunStatus));
    if (e) return e;
    RunStatus* st = sw.status.as<RunStatus>();
    OVTK_HIP(hipMemsetAsync(st, 0, sizeof(RunStatus), s));
    int32_t *t_rb = sw.gen[0].as<int32_t>(), *t_re = sw.gen[1].as<int32_t>(), *t_b = sw.gen[2].as<int32_t>(), *t_e = sw.gen[3].as<int32_t>();
    uint8_t* t_sk = sw.gen[4].as<uint8_t>();
    OVTK_LAUNCH(sw.marks, "special_split", special_sparse_kernel, (n_rows + kWave - 1) / kWave, kBlockThreads, s, d_in, h->dev, st, cap_extra, t_rb, t_re, t_b,
                t_e, t_sk);
    launch_scan(sw.marks, "special_split", s, (long long)n_rows, SparseRowLen{t_rb, t_re, nullptr}, SparseRowGather{t_rb, t_b, t_e, t_sk, d_rb, d_re, d_b, d_e, d_sk},
                SparseFin{st, capacity}, sw.tiles.as<long long>(), st, kFlagOutCapacity | kFlagRange);
    return OVTK_OK;
}

// SpecialTokensSplit's passes into device buffers (count, offsets, write): launched on `s`, nobody waits.  The workspace's row_cnt /
// wave_off / tiles / status serve them; the number of output strings stays on the device (status->n_out).
int special_on_device(const ovtk_special_tokens_split* h, Workspace& sw, const RowsIn& d_in, hipStream_t s, int32_t* d_rb, int32_t* d_re,
                      int32_t* d_b, int32_t* d_e, uint8_t* d_sk, long long capacity) {
    const int n_rows = d_in.n_rows;
    const int grid = grid_lookup(h->device, n_rows);
    const int n_tiles = (n_rows + kRowTile - 1) / kRowTile;
    int e = 0;
    e = e ? e : sw.row_cnt.ensure(size_t(n_rows) * 4);
    e = e ? e : sw.wave_off.ensure(size_t(grid * kWavesPerBlock + 1) * sizeof(long long));
    e = e ? e : sw.tiles.ensure(size_t(n_tiles + 1) * sizeof(long long));
    e = e ? e : sw.status.ensure(sizeof(RunStatus));
    if (e) return e;
    EncodeWork w{};
    w.n_waves = grid * kWavesPerBlock;
    w.wave_off = sw.wave_off.as<long long>();
    w.row_cnt = sw.row_cnt.as<int32_t>();
    w.tile_off = sw.tiles.as<long long>();
    w.stage_cap = INT32_MAX;
    w.status = sw.status.as<RunStatus>();
    OVTK_HIP(hipMemsetAsync(w.status, 0, sizeof(RunStatus), s));
    OVTK_LAUNCH(sw.marks, "prep_rows", prep_rows_kernel, std::min(grid, kTicketBlocks), kBlockThreads, s, d_in, 1, w);  // offset validation
    const int seq_grid = (n_rows + kBlockThreads - 1) / kBlockThreads;
    OVTK_LAUNCH(sw.marks, "special_count", special_split_kernel<0>, seq_grid, kBlockThreads, s, d_in, h->dev, w, (int32_t*)nullptr,
                (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (uint8_t*)nullptr);
    OVTK_LAUNCH(sw.marks, "count_scan", count_scan_kernel, std::min((n_tiles + 3) / 4, kTicketBlocks), kBlockThreads, s, n_rows, w, capacity);
    OVTK_LAUNCH(sw.marks, "special_write", special_split_kernel<1>, seq_grid, kBlockThreads, s, d_in, h->dev, w, d_rb, d_re, d_b, d_e, d_sk);
    return OVTK_OK;
}

// RegexSplit [+] BPETokenizer.  split == nullptr: `in` already holds pieces (the BPETokenizer op).
// Launches the kernels; `run` stays empty when the result was complete without any (empty batches).
int start_encode(const ovtk_regex_split* split_in, const ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                 ovtk_ragged_i32_out* out, int mem, void* stream, std::unique_ptr<PendingRun>& run, const WireSink* wire = nullptr,
                 std::shared_ptr<void> device_inputs = nullptr, const ovtk_special_tokens_split* special = nullptr,
                 const DenseSink* dense = nullptr, std::shared_ptr<int32_t> dense_width = nullptr) {
    // device_inputs: `in` names device memory whatever `mem` says about the outputs (ovtk_encode_enqueue_packed: the packed
    // batch was copied to the device by the caller of this function); the object owns that memory until the run is over.
    const int in_mem = device_inputs ? OVTK_MEM_DEVICE : mem;
    const ovtk_regex_split* split = split_in;
    if (int rc = check_rows(in)) return rc;
    if (!bpe || !out) return set_error(OVTK_E_ARG, "null argument");
    if (split && split->device != bpe->device) return set_error(OVTK_E_ARG, "split and bpe handles live on different devices");
    if (out->data_capacity < 0 || out->data_capacity >= INT32_MAX) return set_error(OVTK_E_ARG, "bad output capacity");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(bpe->device));
    out->n_data = 0;
    out->n_rows = in->n_rows;
    if (split && in->strings.n_chars == 0) {
        // regex_split.cpp:129-143 quirk: an all-empty batch leaves RegexSplit with ragged shape {1} = [0],[0];
        // BPETokenizer then emits one empty row (bpe_tokenizer.cpp:129-131,146-161).
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

    const int dev = bpe->device;
    // SpecialTokensSplit in front (the graph of tokenizer_pipeline.py:1613-1636: SpecialTokensSplit -> RegexSplit -> BPETokenizer):
    // its passes write the split strings and their skip flags into device buffers of the reference's capacity, the kernels below
    // read them from there on the same stream -- the string count stays on the device (the lookup kernels want a bound, not the
    // number), and the host waits for nothing in between.
    std::shared_ptr<WorkspaceLease> special_ws;
    ovtk_ragged_strings special_out{};
    if (special) {
        if (special->device != dev) return set_error(OVTK_E_ARG, "special-tokens and bpe handles live on different devices");
        special_ws = std::make_shared<WorkspaceLease>(dev);
        Workspace& sw = *special_ws->ws;
        if (!sw.host_status) return set_error(OVTK_E_HIP, "pinned host allocation failed");
        RowsIn d_in{};
        if (int rc = stage_input(sw, in, skips, in_mem, s, d_in)) return rc;
        // special_tokens_split.cpp:88-92, + skipped empty strings -- behind one entry per row (special_sparse_kernel's layout)
        const int64_t cap_extra = in->strings.n_chars + in->strings.n;
        const int64_t cap = in->n_rows + cap_extra;
        if (cap >= INT32_MAX) return set_error(OVTK_E_ARG, "tensor sizes must fit int32 offsets");
        int e = 0;
        e = e ? e : sw.gen[0].ensure(size_t(d_in.n_rows) * 4);
        e = e ? e : sw.gen[1].ensure(size_t(d_in.n_rows) * 4);
        e = e ? e : sw.gen[2].ensure(size_t(cap) * 4);
        e = e ? e : sw.gen[3].ensure(size_t(cap) * 4);
        e = e ? e : sw.gen[4].ensure(size_t(cap));
        if (e) return e;
        // one pass, a wave per 64 rows, every wave's strings in a region of their own (special_sparse_kernel)
        if (int rc = sw.status.ensure(sizeof(RunStatus))) return rc;
        OVTK_HIP(hipMemsetAsync(sw.status.as<void>(), 0, sizeof(RunStatus), s));
        OVTK_LAUNCH(sw.marks, "special_split", special_sparse_kernel, (d_in.n_rows + kWave - 1) / kWave, kBlockThreads, s, d_in,
                    special->dev, sw.status.as<RunStatus>(), (long long)cap_extra, sw.gen[0].as<int32_t>(), sw.gen[1].as<int32_t>(),
                    sw.gen[2].as<int32_t>(), sw.gen[3].as<int32_t>(), sw.gen[4].as<uint8_t>());
        OVTK_HIP(hipMemcpyAsync(sw.host_status, sw.status.as<RunStatus>(), sizeof(RunStatus), hipMemcpyDeviceToHost, s));   // (read at finish)
        special_out = ovtk_ragged_strings{sw.gen[0].as<int32_t>(), sw.gen[1].as<int32_t>(), in->n_rows,
                                          ovtk_strings{sw.gen[2].as<int32_t>(), sw.gen[3].as<int32_t>(), d_in.chars, cap, in->strings.n_chars}};
        in = &special_out;
        skips = sw.gen[4].as<uint8_t>();
    }
    const int in_mem2 = special ? OVTK_MEM_DEVICE : in_mem;
    // A split the lookup kernel has no scanner for (the compiled DFA, the class patterns, max_splits): the pieces are
    // produced in device memory first -- the chain RegexSplit -> BPETokenizer inside one call; the piece offsets make
    // one round trip through HBM and the host waits once for their count.
    std::shared_ptr<WorkspaceLease> pieces_ws;
    bool sparse_status = false;   // the split stage left a status block to look at when the run has finished
    ovtk_ragged_strings pieces{};
    // (a family of span_fam.hpp behind other splits -- DeepSeek-V3's three in a row -- gets rows of several strings: the span kernel
    // leaves those to the literal matcher, a lane per row, and the compiled DFA is the faster of the two for them)
    const bool one_string_rows = special != nullptr || in->strings.n == in->n_rows;
    if (split && (!fusable(split, bpe) || (split->dev.family != kFamNone && !one_string_rows))) {
        pieces_ws = std::make_shared<WorkspaceLease>(dev);
        Workspace& sw = *pieces_ws->ws;
        if (!sw.host_status) return set_error(OVTK_E_HIP, "pinned host allocation failed");
        RowsIn d_in{};
        if (int rc = stage_input(sw, in, skips, in_mem2, s, d_in)) return rc;
        const int64_t cap = in->strings.n_chars + in->strings.n;  // regex_split.cpp:182
        if (cap >= INT32_MAX) return set_error(OVTK_E_ARG, "tensor sizes must fit int32 offsets");
        int e = 0;
        e = e ? e : sw.gen[0].ensure(size_t(d_in.n_rows) * 4);
        e = e ? e : sw.gen[1].ensure(size_t(d_in.n_rows) * 4);
        e = e ? e : sw.gen[2].ensure(size_t(cap) * 4);
        e = e ? e : sw.gen[3].ensure(size_t(cap) * 4);
        if (e) return e;
        int64_t n_pieces = 0;
        if (split->dev.kind == kSplitGeneral) {
            // the compiled DFA in one pass: every row's pieces in a region of its own inside the buffers of the reference's
            // capacity, the ragged begins / ends pointing there -- nothing is counted, the host does not wait (regex_device.hpp)
            if (int rc = sw.status.ensure(sizeof(RunStatus))) return rc;
            OVTK_HIP(hipMemsetAsync(sw.status.as<void>(), 0, sizeof(RunStatus), s));
            const int lane_grid = (d_in.n_rows + kBlockThreads - 1) / kBlockThreads;
            const RegexSparseLds lay = regex_sparse_layout(split->regex.n_states * split->regex.n_syms, split->regex.cp_blocks_bytes);
            note_launch(sw.marks, s);
            Profiler& pf = Profiler::get();
            if (pf.enabled()) pf.begin("regex_split", s, sw.marks);
            if (split->mode == 1 && split->max_splits == -1)
                hipLaunchKernelGGL(regex_sparse_kernel<true>, dim3(lane_grid), dim3(kBlockThreads), size_t(lay.total), s, d_in, split->regex,
                                   sw.status.as<RunStatus>(), (long long)cap, sw.gen[0].as<int32_t>(), sw.gen[1].as<int32_t>(),
                                   sw.gen[2].as<int32_t>(), sw.gen[3].as<int32_t>(), (uint8_t*)nullptr);
            else
                hipLaunchKernelGGL(regex_sparse_kernel<false>, dim3(lane_grid), dim3(kBlockThreads), size_t(lay.total), s, d_in, split->regex,
                                   sw.status.as<RunStatus>(), (long long)cap, sw.gen[0].as<int32_t>(), sw.gen[1].as<int32_t>(),
                                   sw.gen[2].as<int32_t>(), sw.gen[3].as<int32_t>(), (uint8_t*)nullptr);
            if (pf.enabled()) pf.end(s, sw.marks);
            OVTK_HIP(hipMemcpyAsync(sw.host_status, sw.status.as<RunStatus>(), sizeof(RunStatus), hipMemcpyDeviceToHost, s));   // (read at finish)
            sparse_status = true;
            n_pieces = cap;   // (the offsets index the whole buffers)
        } else if (int rc = split_on_device(split, sw, d_in, s, sw.gen[0].as<int32_t>(), sw.gen[1].as<int32_t>(), sw.gen[2].as<int32_t>(),
                                            sw.gen[3].as<int32_t>(), nullptr, cap, &n_pieces))
            return rc;
        pieces = ovtk_ragged_strings{sw.gen[0].as<int32_t>(), sw.gen[1].as<int32_t>(), in->n_rows,
                                     ovtk_strings{sw.gen[2].as<int32_t>(), sw.gen[3].as<int32_t>(), d_in.chars, n_pieces,
                                                  in->strings.n_chars}};
        in = &pieces;
        skips = nullptr;  // BPETokenizer has no skips input: skipped strings arrive as whole pieces
        split = nullptr;
    }
    // This call's view of the tables: the piece store is left out while it is paused (a text whose pieces never repeat -- uniform
    // random bytes -- only pays for it: probes that miss, inserts nobody reads; 0.56 -> 0.62 ms per step before the pause existed)
    BpeDev T = bpe->dev;
    if (T.store.slots && bpe->store_pause.load(std::memory_order_relaxed) > 0) {
        bpe->store_pause.fetch_sub(1, std::memory_order_relaxed);
        T.store = PieceStoreDev{nullptr, 30, nullptr, 0};
    }
    auto r = make_rows_run(dev, "BPETokenizer", in, skips, 1 + T.suffix_len, out, mem, s,
                           [=](Workspace& ws, const RowsIn& d_in, const EncodeWork& w, int grid) {
                               const bool tickets = w.rows_per_ticket != 0;
                               const bool llama3 = split && split->dev.kind == kSplitLlama3;
                               const int family = split ? split->dev.family : int(kFamNone);
                               if (w.small) {  // the whole call in one launch of one block
                                   const SplitDev sd = split ? split->dev : SplitDev{};
                                   const bool nar = bpe->narrow_ids;
#define OVTK_SMALL(MODE)                                                                                                          \
    do {                                                                                                                          \
        if (nar) OVTK_LAUNCH(ws.marks, "encode_small", (encode_small_kernel<MODE, true>), grid, kBlockThreads, s, d_in, sd, T, w); \
        else OVTK_LAUNCH(ws.marks, "encode_small", (encode_small_kernel<MODE, false>), grid, kBlockThreads, s, d_in, sd, T, w);    \
    } while (0)
                                   if (llama3) OVTK_SMALL(kFusedLlama3);
                                   else if (family) OVTK_SMALL(kFusedSeq);
                                   else if (split) OVTK_SMALL(kFused);
                                   else OVTK_SMALL(kPieces);
#undef OVTK_SMALL
                                   return;
                               }
                               if (llama3 && tickets)
                                   OVTK_LAUNCH(ws.marks, "lookup_fused", (lookup_kernel<kFusedLlama3, true>), grid, kBlockThreads, s, d_in,
                                               split->dev, T, w);
                               else if (llama3) {
                                   // several rows per scan block, the rule algebra on bit masks (lookup_span_kernel<kSpanLlama3>, span_l3.hpp;
                                   // round 4's row-per-scan kernel stays for handles without a memo); what it leaves goes through the
                                   // generic kernel
                                   EncodeWork w1 = w;
                                   const int grid1 = rows_grid(d_in.n_rows, grid, w1.rows_per_wave);
                                   if (!(w.launch_mask & kLaunchSpan)) {}   // (the short path's second set of launches: the span kernel has run)
                                   else if (T.pieces.slots && w1.stage16)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanLlama3, true>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else if (T.pieces.slots)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanLlama3, false>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else
                                       OVTK_LAUNCH(ws.marks, "lookup_rows", lookup_rows_kernel<kRowsLlama3>, grid1, kBlockThreads, s, d_in, split->dev,
                                                   T, w1);
                                   EncodeWork w2 = w;
                                   w2.only_pending = 1;
                                   if (w.launch_mask & kLaunchPending)
                                       OVTK_LAUNCH(ws.marks, "lookup_fused", lookup_kernel<kFusedLlama3>, grid, kBlockThreads, s, d_in, split->dev,
                                                   T, w2);
                               }
                               else if (family && tickets)
                                   OVTK_LAUNCH(ws.marks, "lookup_fused", (lookup_kernel<kFusedSeq, true>), grid, kBlockThreads, s, d_in,
                                               split->dev, T, w);
                               else if (family) {
                                   // DeepSeek-V3's pattern, o200k_base: the scan of span_fam.hpp on the span kernel's blocks; the rows it leaves
                                   // (several strings, skipped ones) are matched literally, a lane per row's window
                                   EncodeWork w1 = w;
                                   const int grid1 = rows_grid(d_in.n_rows, grid, w1.rows_per_wave);
                                   if (!(w.launch_mask & kLaunchSpan)) {}
                                   else if (family == kFamDs3 && w1.stage16)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanDs3, true>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else if (family == kFamDs3)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanDs3, false>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else if (w1.stage16)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanO200k, true>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanO200k, false>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   EncodeWork w2 = w;
                                   w2.only_pending = 1;
                                   if (w.launch_mask & kLaunchPending)
                                       OVTK_LAUNCH(ws.marks, "lookup_fused", lookup_kernel<kFusedSeq>, grid, kBlockThreads, s, d_in, split->dev, T, w2);
                               }
                               else if (split && tickets)
                                   OVTK_LAUNCH(ws.marks, "lookup_fused", (lookup_kernel<kFused, true>), grid, kBlockThreads, s, d_in,
                                               split->dev, T, w);
                               else if (split && split->dev.kind <= kSplitGpt2Digits) {
                                   // several rows per scan block: lookup_span_kernel (it probes the memo for every piece: a handle
                                   // without one -- cache_capacity = 0 -- takes the row-per-scan kernel); whatever they leave
                                   // (marked in row_used, listed in pending_rows) goes through the generic kernel
                                   EncodeWork w1 = w;
                                   const int grid1 = rows_grid(d_in.n_rows, grid, w1.rows_per_wave);
                                   if (!(w.launch_mask & kLaunchSpan)) {}
                                   else if (T.pieces.slots && split->dev.kind == kSplitGpt2Digits && w1.stage16)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanGpt2Digits, true>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else if (T.pieces.slots && split->dev.kind == kSplitGpt2Digits)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanGpt2Digits, false>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else if (T.pieces.slots && w1.stage16)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanGpt2, true>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else if (T.pieces.slots)
                                       OVTK_LAUNCH(ws.marks, "lookup_span", (lookup_span_kernel<kSpanGpt2, false>), grid1, kBlockThreads, s, d_in, split->dev, T, w1);
                                   else if (split->dev.kind == kSplitGpt2Digits)
                                       OVTK_LAUNCH(ws.marks, "lookup_rows", lookup_rows_kernel<kRowsGpt2Digits>, grid1, kBlockThreads, s, d_in,
                                                   split->dev, T, w1);
                                   else
                                       OVTK_LAUNCH(ws.marks, "lookup_rows", lookup_rows_kernel<kRowsGpt2>, grid1, kBlockThreads, s, d_in, split->dev,
                                                   T, w1);
                                   // what it left: the generic kernel (it returns at once when nothing was left).  A smaller stand-by
                                   // grid for handles whose last call left no row was measured: nothing gained on all-ASCII text
                                   // (6.0 vs 6.2 us), and the rows that do turn up then wait for 64 blocks to walk every row's flag
                                   EncodeWork w2 = w;
                                   w2.only_pending = 1;
                                   if (w.launch_mask & kLaunchPending)
                                       OVTK_LAUNCH(ws.marks, "lookup_fused", lookup_kernel<kFused>, grid, kBlockThreads, s, d_in, split->dev,
                                                   T, w2);
                               } else if (split)
                                   OVTK_LAUNCH(ws.marks, "lookup_fused", lookup_kernel<kFused>, grid, kBlockThreads, s, d_in,
                                               split->dev, T, w);
                               else if (tickets)
                                   OVTK_LAUNCH(ws.marks, "lookup_pieces", (lookup_kernel<kPieces, true>), grid, kBlockThreads, s, d_in,
                                               SplitDev{}, T, w);
                               else
                                   OVTK_LAUNCH(ws.marks, "lookup_pieces", lookup_kernel<kPieces>, grid, kBlockThreads, s, d_in,
                                               SplitDev{}, T, w);
                               if (!(w.launch_mask & kLaunchMerge)) return;   // (the short path: nothing is expected to be left to merge)
                               const int tail_rows = w.fold_tail ? d_in.n_rows : 0;
                               if (bpe->narrow_ids) {
                                   static const int per_cu = resident_blocks_per_cu(merge_kernel<true>);
                                   OVTK_LAUNCH(ws.marks, "bpe_merge", merge_kernel<true>,
                                               dim3(kShards, grid_deferred_hinted(grid_deferred_per_shard(d_in.n_chars, d_in.n_strings, device_cu_count(dev) * per_cu / kShards), w.span_sums, w.merge_hint)),
                                               kBlockThreads, s, d_in, T, w, tail_rows, w.out_cap);
                               } else {
                                   static const int per_cu = resident_blocks_per_cu(merge_kernel<false>);
                                   OVTK_LAUNCH(ws.marks, "bpe_merge", merge_kernel<false>,
                                               dim3(kShards, grid_deferred_hinted(grid_deferred_per_shard(d_in.n_chars, d_in.n_strings, device_cu_count(dev) * per_cu / kShards), w.span_sums, w.merge_hint)),
                                               kBlockThreads, s, d_in, T, w, tail_rows, w.out_cap);
                               }
                               if (!w.fold_tail) OVTK_LAUNCH(ws.marks, "bpe_exact", exact_kernel, 64, kBlockThreads, s, d_in, T, w);
                           },
                           /*self_alloc=*/true,
                           !split ? resident_blocks_per_cu(lookup_kernel<kPieces>)
                                  : split->dev.family != kFamNone
                                      ? (!row_tickets().load(std::memory_order_relaxed) ? resident_blocks_per_cu(lookup_span_kernel<kSpanO200k, false>)
                                                                                        : resident_blocks_per_cu(lookup_kernel<kFusedSeq>))
                                  : split->dev.kind == kSplitLlama3
                                      ? (T.pieces.slots && !row_tickets().load(std::memory_order_relaxed)
                                             ? resident_blocks_per_cu(lookup_span_kernel<kSpanLlama3, false>)
                                             : resident_blocks_per_cu(lookup_kernel<kFusedLlama3>))
                                  : split->dev.kind <= kSplitGpt2Digits && !row_tickets().load(std::memory_order_relaxed)
                                      ? resident_blocks_per_cu(lookup_span_kernel<kSpanGpt2, true>, 6)
                                      : resident_blocks_per_cu(lookup_kernel<kFused>),
                           /*tail_in_middle=*/true);
    if (pieces_ws) r->also_settles(pieces_ws);
    if (special_ws) r->also_settles(special_ws);
    if (special_ws || sparse_status) {
        std::shared_ptr<WorkspaceLease> sparse_ws = sparse_status ? pieces_ws : nullptr;
        r->front_check([special_ws, sparse_ws]() -> int {
            if (special_ws) {
                const RunStatus& st = *special_ws->ws->host_status;
                if (st.flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside their tensors");
                if (st.flags & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "SpecialTokensSplit: more strings than the reference's capacity");
            }
            if (sparse_ws && (sparse_ws->ws->host_status->flags & kFlagOutCapacity))
                return set_error(OVTK_E_CAPACITY, "RegexSplit: the strings overlap -- their pieces need more room than the reference's n_chars + n_strings "
                                                  "(regex_split.cpp:182)");
            return OVTK_OK;
        });
    }
    if (pieces_ws || special_ws)
        r->input_on_device(std::make_shared<std::tuple<std::shared_ptr<void>, std::shared_ptr<void>, std::shared_ptr<void>>>(pieces_ws, special_ws, device_inputs));
    else if (device_inputs)
        r->input_on_device(device_inputs);
    const bool has_store = T.store.slots != nullptr;
    // The short path (span_kernel.hpp): where the call's first kernel is lookup_span_kernel, the kernels behind it are launched only when
    // the handle's last calls had work for them.
    const bool short_path = split && (split->dev.kind <= kSplitGpt2Digits || split->dev.kind == kSplitLlama3 || split->dev.family != kFamNone) && T.pieces.slots &&
                            !row_tickets().load(std::memory_order_relaxed) && short_path_mode().load(std::memory_order_relaxed) != 0;
    if (short_path)
        r->enable_short_path(bpe->expect_pending.load(std::memory_order_relaxed) > 0, bpe->expect_merge.load(std::memory_order_relaxed) > 0 || !has_store,
                             has_store ? bpe->last_unresolved.load(std::memory_order_relaxed) : -1);
    if (has_store || dense_width || short_path)   // what the store did for this call decides whether the next ones ask it at all
        r->on_status([bpe, has_store, dense_width](const RunStatus& st) {
            if (dense_width) *dense_width = st.width;
            if (st.short_path) {   // the span kernel counted: what this call's text left to the kernels of the middle
                auto note = [](std::atomic<int>& expect, bool had_work) {
                    if (had_work) expect.store(kShortPathKeep, std::memory_order_relaxed);
                    else if (expect.load(std::memory_order_relaxed) > 0) expect.fetch_sub(1, std::memory_order_relaxed);
                };
                note(bpe->expect_pending, st.n_pending > 0);
                if (st.short_path != 3) {   // (3: the long way, nothing was counted)
                    note(bpe->expect_merge, st.n_unresolved > 0);
                    bpe->last_unresolved.store(st.n_unresolved, std::memory_order_relaxed);
                }
            }
            if (!has_store) return;
            if (st.n_store_probe < 256) return;   // (counted by one wave in 64)
            // (a cold store misses everything too: only a run of such calls says that the text is the reason)
            if (st.n_store_hit * 8 >= st.n_store_probe) bpe->store_low.store(0, std::memory_order_relaxed);
            else if (bpe->store_low.fetch_add(1, std::memory_order_relaxed) + 1 >= 4) {
                bpe->store_low.store(0, std::memory_order_relaxed);
                bpe->store_pause.store(32, std::memory_order_relaxed);
            }
        });
    if (!row_tickets().load(std::memory_order_relaxed)) r->enable_small();
    if (bpe->stage16) r->enable_stage16();
    if (split && (split->dev.kind <= kSplitGpt2Digits || split->dev.kind == kSplitLlama3 || split->dev.family != kFamNone) && T.pieces.slots) r->stage_twice();   // (lookup_span_kernel in front of the generic kernel)
    if (wire) r->output_to_wire(*wire);
    if (dense) r->output_dense(*dense);
    if (int rc = r->start()) return rc;
    run = std::move(r);
    return OVTK_OK;
}

int run_encode(const ovtk_regex_split* split, const ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
               ovtk_ragged_i32_out* out, int mem, void* stream) {
    std::unique_ptr<PendingRun> run;
    if (int rc = start_encode(split, bpe, in, skips, out, mem, stream, run)) return rc;
    return run ? run->finish(out) : OVTK_OK;
}

int check_fused(const ovtk_regex_split* split) {
    if (!split) return set_error(OVTK_E_ARG, "null split handle");
    return OVTK_OK;
}

}  // namespace

extern "C" {

int ovtk_bpe_run(ovtk_bpe* h, const ovtk_ragged_strings* in, ovtk_ragged_i32_out* out, int mem, void* stream) {
    return run_encode(nullptr, h, in, nullptr, out, mem, stream);
}

int ovtk_encode_run(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                    ovtk_ragged_i32_out* out, int mem, void* stream) {
    if (int rc = check_fused(split)) return rc;
    return run_encode(split, bpe, in, skips, out, mem, stream);
}

int ovtk_encode_special_run(ovtk_special_tokens_split* special, ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in,
                            const uint8_t* skips, ovtk_ragged_i32_out* out, int mem, void* stream) {
    if (!special) return set_error(OVTK_E_ARG, "null special-tokens handle");
    if (int rc = check_fused(split)) return rc;
    std::unique_ptr<PendingRun> run;
    if (int rc = start_encode(split, bpe, in, skips, out, mem, stream, run, nullptr, nullptr, special)) return rc;
    return run ? run->finish(out) : OVTK_OK;
}

int ovtk_encode_special_enqueue(ovtk_special_tokens_split* special, ovtk_regex_split* split, ovtk_bpe* bpe,
                                const ovtk_ragged_strings* in, const uint8_t* skips, const ovtk_ragged_i32_out* out, void* stream,
                                ovtk_pending** pending) {
    if (!pending || !out || !special) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_fused(split)) return rc;
    auto p = std::make_unique<ovtk_pending>();
    p->out = *out;
    if (int rc = start_encode(split, bpe, in, skips, &p->out, OVTK_MEM_DEVICE, stream, p->run, nullptr, nullptr, special)) return rc;
    *pending = p.release();
    return OVTK_OK;
}

int ovtk_encode_dense_enqueue(ovtk_special_tokens_split* special, ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in,
                              const uint8_t* skips, const ovtk_dense_params* params, int32_t* out_ids, uint8_t* out_mask, int64_t capacity,
                              void* stream, ovtk_pending** pending) {
    if (!pending || !params || !out_ids || !in || capacity < 0) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_fused(split)) return rc;
    if (params->n_prefix < 0 || params->n_prefix > kDenseAffix || params->n_suffix < 0 || params->n_suffix > kDenseAffix ||
        (params->n_prefix && !params->prefix) || (params->n_suffix && !params->suffix) || params->max_length < 0)
        return set_error(OVTK_E_ARG, "encode_dense: at most 4 constant ids in front / behind, max_length >= 0");
    if (int rc = check_rows(in)) return rc;
    DenseSink d{};
    d.ids = out_ids;
    d.mask = out_mask;
    d.capacity = capacity;
    d.max_length = params->max_length;
    d.trunc_left = params->trunc_left ? 1 : 0;
    d.pad_right = params->pad_right ? 1 : 0;
    d.pad_value = params->pad_value;
    d.target_dim = params->target_dim;
    d.n_pre = params->n_prefix;
    d.n_suf = params->n_suffix;
    for (int i = 0; i < d.n_pre; ++i) d.pre[i] = params->prefix[i];
    for (int i = 0; i < d.n_suf; ++i) d.suf[i] = params->suffix[i];
    auto p = std::make_unique<ovtk_pending>();
    p->dense_width = std::make_shared<int32_t>(params->target_dim < 0 ? 0 : params->target_dim);
    p->out = ovtk_ragged_i32_out{nullptr, nullptr, nullptr, INT32_MAX - 2, 0, 0};   // (no ragged ids: nothing of that kind overflows)
    if (in->strings.n_chars == 0 || in->n_rows == 0) {
        // no text (the all-empty quirk of regex_split.cpp:129-143 is the ragged tensor's; the dense tensors keep their rows): every row is
        // the constant segments and padding -- T cells of which the host knows everything
        const int32_t len = d.n_pre + d.n_suf, T = d.target_dim < 0 ? len : d.target_dim;
        *p->dense_width = T;
        if (in->n_rows * int64_t(T) > capacity) return set_error(OVTK_E_CAPACITY, "encode_dense: output too small");
        if (T > 0 && in->n_rows > 0) {
            std::vector<int32_t> row(size_t(T), d.pad_value);
            std::vector<uint8_t> mrow(size_t(T), 0);
            const int col0 = d.pad_right ? 0 : std::max(0, T - len);
            for (int k = 0; k < len && col0 + k < T; ++k) {
                row[size_t(col0 + k)] = k < d.n_pre ? d.pre[k] : d.suf[k - d.n_pre];
                mrow[size_t(col0 + k)] = 1;
            }
            hipStream_t s = static_cast<hipStream_t>(stream);
            OVTK_HIP(hipSetDevice(bpe ? bpe->device : 0));
            for (int64_t r = 0; r < in->n_rows; ++r) {
                OVTK_HIP(hipMemcpyAsync(out_ids + r * T, row.data(), size_t(T) * 4, hipMemcpyHostToDevice, s));
                if (out_mask) OVTK_HIP(hipMemcpyAsync(out_mask + r * T, mrow.data(), size_t(T), hipMemcpyHostToDevice, s));
            }
            OVTK_HIP(hipStreamSynchronize(s));
        }
        p->out.n_rows = in->n_rows;
        *pending = p.release();
        return OVTK_OK;
    }
    if (int rc = start_encode(split, bpe, in, skips, &p->out, OVTK_MEM_DEVICE, stream, p->run, nullptr, nullptr, special, &d, p->dense_width))
        return rc;
    *pending = p.release();
    return OVTK_OK;
}

int ovtk_encode_dense_finish(ovtk_pending* pending, int32_t* width, int64_t* n_ids) {
    if (!pending) return set_error(OVTK_E_ARG, "null pending call");
    std::unique_ptr<ovtk_pending> p(pending);
    int rc = OVTK_OK;
    if (p->run) rc = p->run->finish(&p->out);
    if (width) *width = p->dense_width ? *p->dense_width : 0;
    if (n_ids) *n_ids = p->out.n_data;
    return rc;
}

int ovtk_set_short_path(int mode) {
    if (mode < 0 || mode > 2) return set_error(OVTK_E_ARG, "short path: 0 (never), 1 (where a handle's last calls say it works), 2 (every eligible call tries)");
    short_path_mode().store(mode, std::memory_order_relaxed);
    return OVTK_OK;
}

int ovtk_short_path_stats(int64_t* tried, int64_t* exact) {
    if (tried) *tried = short_path_counts().tried.load(std::memory_order_relaxed);
    if (exact) *exact = short_path_counts().exact.load(std::memory_order_relaxed);
    return OVTK_OK;
}

int ovtk_set_row_tickets(int rows_per_ticket) {
    if (rows_per_ticket < 0 || rows_per_ticket > 64) return set_error(OVTK_E_ARG, "row tickets: 0 (static) .. 64 rows per ticket");
    row_tickets().store(rows_per_ticket, std::memory_order_relaxed);
    return OVTK_OK;
}

int ovtk_encode_enqueue(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                        const ovtk_ragged_i32_out* out, void* stream, ovtk_pending** pending) {
    if (!pending || !out) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_fused(split)) return rc;
    auto p = std::make_unique<ovtk_pending>();
    p->out = *out;
    if (int rc = start_encode(split, bpe, in, skips, &p->out, OVTK_MEM_DEVICE, stream, p->run)) return rc;
    *pending = p.release();
    return OVTK_OK;
}

int ovtk_encode_enqueue_host(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips,
                             const ovtk_ragged_i32_out* out, void* stream, ovtk_pending** pending) {
    if (!pending || !out) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_fused(split)) return rc;
    auto p = std::make_unique<ovtk_pending>();
    p->out = *out;
    if (int rc = start_encode(split, bpe, in, skips, &p->out, OVTK_MEM_HOST, stream, p->run)) return rc;
    *pending = p.release();
    return OVTK_OK;
}

int ovtk_encode_enqueue_wire(ovtk_regex_split* split, ovtk_bpe* bpe, const ovtk_ragged_strings* in, const uint8_t* skips, void* wire,
                             int64_t max_rows, int64_t pad_ids, int id_bytes, void* stream, ovtk_pending** pending) {
    if (!pending || !wire || !in) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_fused(split)) return rc;
    if ((id_bytes != 2 && id_bytes != 4) || pad_ids < 0 || pad_ids % 8 || pad_ids >= INT32_MAX || max_rows % 4 || in->n_rows > max_rows)
        return set_error(OVTK_E_ARG, "encode to wire: bad wire geometry (ovtk_shard_max_rows / pad_ids / id_bytes of the exchange)");
    if (int rc = check_rows(in)) return rc;
    auto p = std::make_unique<ovtk_pending>();
    p->out = ovtk_ragged_i32_out{nullptr, nullptr, nullptr, INT32_MAX - 2, 0, 0};  // the wire cuts at pad_ids, nothing overflows
    uint8_t* base = static_cast<uint8_t*>(wire);
    if (in->strings.n_chars == 0 || in->n_rows == 0) {
        // A shard without text (shard_rows_by_bytes next to one very long row, or rows of empty strings): its wire is the
        // header {0 ids, n_rows} and row ends of zero -- written here, no kernel has anything to do.  (The one-row quirk
        // of an all-empty RegexSplit batch, regex_split.cpp:129-143, is a property of a whole batch; a shard keeps its rows.)
        if (!bpe) return set_error(OVTK_E_ARG, "null argument");
        hipStream_t s = static_cast<hipStream_t>(stream);
        OVTK_HIP(hipSetDevice(bpe->device));
        OVTK_HIP(hipMemsetAsync(base, 0, size_t(kShardHeaderBytes) + size_t(max_rows) * 4, s));
        const int32_t rows32 = int32_t(in->n_rows);
        OVTK_HIP(hipMemcpyAsync(base + 4, &rows32, 4, hipMemcpyHostToDevice, s));
        OVTK_HIP(hipStreamSynchronize(s));   // (rows32 is read by then) ovtk_encode_finish() of this call has nothing to wait for
        p->out.n_rows = in->n_rows;
        *pending = p.release();
        return OVTK_OK;
    }
    const WireSink sink{reinterpret_cast<int32_t*>(base), reinterpret_cast<int32_t*>(base + kShardHeaderBytes),
                        base + kShardHeaderBytes + max_rows * 4, int32_t(pad_ids), int32_t(max_rows), id_bytes};
    if (int rc = start_encode(split, bpe, in, skips, &p->out, OVTK_MEM_DEVICE, stream, p->run, &sink)) return rc;
    *pending = p.release();
    return OVTK_OK;
}

int ovtk_encode_enqueue_packed(ovtk_regex_split* split, ovtk_bpe* bpe, const uint8_t* packed, int64_t n_bytes,
                               const ovtk_ragged_i32_out* out, int out_mem, void* stream, ovtk_pending** pending) {
    if (!pending || !out || !packed || !bpe) return set_error(OVTK_E_ARG, "null argument");
    if (int rc = check_fused(split)) return rc;
    if (out_mem != OVTK_MEM_HOST && out_mem != OVTK_MEM_DEVICE) return set_error(OVTK_E_ARG, "out_mem must be OVTK_MEM_HOST or OVTK_MEM_DEVICE");
    // the reference's format checks (src/utils.cpp:21-25), on the host copy
    if (n_bytes < 4) return set_error(OVTK_E_ARG, "Incorrect packed string tensor format: no batch size in the packed string tensor");
    int32_t batch = 0, total = 0;
    std::memcpy(&batch, packed, 4);
    if (batch < 0 || n_bytes < 8 + 4 * int64_t(batch))
        return set_error(OVTK_E_ARG, "Incorrect packed string tensor format: the packed string tensor must contain first string offset and end indices");
    if (batch > 0) std::memcpy(&total, packed + 4 + 4 * size_t(batch), 4);  // end_ids[batch - 1] (string_tensor_unpack.cpp:59)
    if (total < 0 || 8 + 4 * int64_t(batch) + total > n_bytes) return set_error(OVTK_E_RANGE, "packed string tensor: end offsets exceed the buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(bpe->device));
    auto p = std::make_unique<ovtk_pending>();
    p->out = *out;
    if (batch == 0 || total == 0) {
        // No text at all: the result follows from the header (start_encode's empty-batch paths read no tensor and launch
        // nothing), so nothing is uploaded and no workspace is leased -- a lease handed back with a copy or a kernel still in
        // flight would be another thread's input buffer a moment later.
        const ovtk_ragged_strings none{nullptr, nullptr, batch, ovtk_strings{nullptr, nullptr, nullptr, batch, 0}};
        if (int rc = start_encode(split, bpe, &none, nullptr, &p->out, out_mem, stream, p->run)) return rc;
        *pending = p.release();
        return OVTK_OK;
    }
    // ONE copy over PCIe; the decomposed tensors are views of the device copy (begin_ids = words 1.., end_ids = words 2..,
    // utils.cpp:26-27), the rows are the strings (ragged_begins = 0, 1, ..; ragged_ends = 1, 2, ..: one table, read at +0 and +1)
    auto ws = std::make_shared<WorkspaceLease>(bpe->device);
    Workspace& w = *ws->ws;
    // From here on work is queued on `s` into the leased buffers: every return that does not hand the lease to a run first waits
    // for the stream (the lease's buffers, and the caller's `packed`, may be reused as soon as this function has returned).
    auto fail = [&](int rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    };
    const size_t used = 8 + 4 * size_t(batch) + size_t(total);
    if (int rc = w.in_chars.upload(packed, used, s)) return fail(rc);
    if (int rc = w.in_rb.ensure((size_t(batch) + 1) * 4)) return fail(rc);
    hipLaunchKernelGGL(iota_kernel, dim3(std::min<int>((batch + kBlockThreads) / kBlockThreads, 1024)), dim3(kBlockThreads), 0, s,
                       batch + 1, w.in_rb.as<int32_t>());  // (this lease is held until the run is over: finish() has waited by then)
    if (hipGetLastError() != hipSuccess) return fail(set_error(OVTK_E_HIP, "iota_kernel launch failed"));
    const uint8_t* d = w.in_chars.as<uint8_t>();
    const int32_t* iota = w.in_rb.as<int32_t>();
    const ovtk_ragged_strings in{iota, iota + 1, batch,
                                 ovtk_strings{reinterpret_cast<const int32_t*>(d + 4), reinterpret_cast<const int32_t*>(d + 8),
                                              d + 8 + 4 * size_t(batch), batch, total}};
    if (int rc = start_encode(split, bpe, &in, nullptr, &p->out, out_mem, stream, p->run, nullptr, ws)) return fail(rc);
    if (!p->run) (void)hipStreamSynchronize(s);   // (not reachable with total > 0; kept so that an empty run never strands queued work)
    *pending = p.release();
    return OVTK_OK;
}

int ovtk_encode_finish(ovtk_pending* pending, ovtk_ragged_i32_out* out) {
    if (!pending) return set_error(OVTK_E_ARG, "null argument");
    std::unique_ptr<ovtk_pending> p(pending);  // released whatever happens
    const int rc = p->run ? p->run->finish(&p->out) : OVTK_OK;
    if (out) *out = p->out;
    return rc;
}

int ovtk_regex_split_run(ovtk_regex_split* h, const ovtk_ragged_strings* in, const uint8_t* skips,
                         ovtk_ragged_strings_out* out, int mem, void* stream) {
    if (int rc = check_rows(in)) return rc;
    if (!h || !out) return set_error(OVTK_E_ARG, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    out->n = 0;
    out->n_rows = in->n_rows;
    if (in->strings.n_chars == 0) {  // regex_split.cpp:129-143
        const int32_t zero = 0;
        if (mem == OVTK_MEM_HOST) {
            out->ragged_begins[0] = 0;
            out->ragged_ends[0] = 0;
        } else {
            OVTK_HIP(hipMemcpyAsync(out->ragged_begins, &zero, 4, hipMemcpyHostToDevice, s));
            OVTK_HIP(hipMemcpyAsync(out->ragged_ends, &zero, 4, hipMemcpyHostToDevice, s));
            OVTK_HIP(hipStreamSynchronize(s));
        }
        out->n_rows = 1;
        out->n = -1;
        return OVTK_OK;
    }
    if (in->n_rows == 0) return OVTK_OK;
    WorkspaceLease ws(h->device);
    if (!ws->host_status) return set_error(OVTK_E_HIP, "pinned host allocation failed");
    RowsIn d_in{};
    if (int rc = stage_input(*ws.ws, in, skips, mem, s, d_in)) return rc;
    const int n_rows = d_in.n_rows;
    int32_t *d_rb = nullptr, *d_re = nullptr, *d_b = nullptr, *d_e = nullptr;
    uint8_t* d_sk = nullptr;
    int e = 0;
    e = e ? e : out_target(ws->out_a, out->ragged_begins, size_t(n_rows) * 4, mem, &d_rb);
    e = e ? e : out_target(ws->out_b, out->ragged_ends, size_t(n_rows) * 4, mem, &d_re);
    e = e ? e : out_target(ws->out_c, out->begins, size_t(out->capacity) * 4, mem, &d_b);
    e = e ? e : out_target(ws->out_d, out->ends, size_t(out->capacity) * 4, mem, &d_e);
    if (out->skips) e = e ? e : out_target(ws->out_e, out->skips, size_t(out->capacity), mem, &d_sk);
    if (e) return e;
    int64_t n_out = 0;
    bool done = false;
    if ((long long)d_in.n_chars + d_in.n_strings < INT32_MAX) {   // one pass over the text (a compiled pattern: the automaton runs once)
        if (h->dev.kind == kSplitGeneral) {
            if (int rc = regex_one_pass(h, *ws.ws, d_in, s, d_rb, d_re, d_b, d_e, d_sk, (long long)out->capacity)) return rc;
        } else if (int rc = split_one_pass(h, *ws.ws, d_in, s, d_rb, d_re, d_b, d_e, d_sk, (long long)out->capacity)) {
            return rc;
        }
        if (int rc = finish_status(*ws.ws, s)) return rc;
        const RunStatus& st = *ws->host_status;
        // (the capacity flag may also mean the one-pass form's own buffers -- strings that overlap ask for more than n_chars + n_strings, and
        // the rows that found no room there look like rows of bad offsets --: the count and write passes below have the last word)
        if (!(st.flags & kFlagOutCapacity)) {
            if (st.flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside their tensors");
            n_out = st.n_out;
            done = true;
        }
    }
    if (!done)
        if (int rc = split_on_device(h, *ws.ws, d_in, s, d_rb, d_re, d_b, d_e, d_sk, out->capacity, &n_out)) return rc;
    out->n = n_out;
    if (mem == OVTK_MEM_HOST) {
        OVTK_HIP(hipMemcpyAsync(out->ragged_begins, d_rb, size_t(n_rows) * 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipMemcpyAsync(out->ragged_ends, d_re, size_t(n_rows) * 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipMemcpyAsync(out->begins, d_b, size_t(n_out) * 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipMemcpyAsync(out->ends, d_e, size_t(n_out) * 4, hipMemcpyDeviceToHost, s));
        if (out->skips) OVTK_HIP(hipMemcpyAsync(out->skips, d_sk, size_t(n_out), hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipStreamSynchronize(s));
    }
    return OVTK_OK;
}

int ovtk_special_tokens_split_run(ovtk_special_tokens_split* h, const ovtk_ragged_strings* in, const uint8_t* skips,
                                  ovtk_ragged_strings_out* out, int mem, void* stream) {
    if (int rc = check_rows(in)) return rc;
    if (!h || !out || !out->skips) return set_error(OVTK_E_ARG, "special_tokens_split: null argument (the skips output is mandatory)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OVTK_HIP(hipSetDevice(h->device));
    out->n = 0;
    out->n_rows = in->n_rows;
    if (in->n_rows == 0) return OVTK_OK;
    WorkspaceLease ws(h->device);
    if (!ws->host_status) return set_error(OVTK_E_HIP, "pinned host allocation failed");
    RowsIn d_in{};
    if (int rc = stage_input(*ws.ws, in, skips, mem, s, d_in)) return rc;
    const int n_rows = d_in.n_rows;
    int e = 0;
    int32_t *d_rb = nullptr, *d_re = nullptr, *d_b = nullptr, *d_e = nullptr;
    uint8_t* d_sk = nullptr;
    e = e ? e : out_target(ws->out_a, out->ragged_begins, size_t(n_rows) * 4, mem, &d_rb);
    e = e ? e : out_target(ws->out_b, out->ragged_ends, size_t(n_rows) * 4, mem, &d_re);
    e = e ? e : out_target(ws->out_c, out->begins, size_t(out->capacity) * 4, mem, &d_b);
    e = e ? e : out_target(ws->out_d, out->ends, size_t(out->capacity) * 4, mem, &d_e);
    e = e ? e : out_target(ws->out_e, out->skips, size_t(out->capacity), mem, &d_sk);
    if (e) return e;
    bool one_pass = (long long)d_in.n_rows + d_in.n_chars + d_in.n_strings < INT32_MAX;   // (the one-pass form's buffers: the three-pass form has none)
    if (one_pass) {
        if (int rc = special_one_pass(h, *ws.ws, d_in, s, d_rb, d_re, d_b, d_e, d_sk, (long long)out->capacity)) return rc;
        if (int rc = finish_status(*ws.ws, s)) return rc;
        // (the capacity flag may also mean the one-pass form's own buffers -- rows that name the same strings again ask for more than
        // n_chars + n_strings --: the count and write passes have the last word)
        if (ws->host_status->flags & kFlagOutCapacity) one_pass = false;
    }
    if (!one_pass) {
        if (int rc = special_on_device(h, *ws.ws, d_in, s, d_rb, d_re, d_b, d_e, d_sk, (long long)out->capacity)) return rc;
        if (int rc = finish_status(*ws.ws, s)) return rc;
    }
    const RunStatus& st = *ws->host_status;
    if (st.flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside their tensors");
    if (st.flags & kFlagOutCapacity) return set_error(OVTK_E_CAPACITY, "SpecialTokensSplit: output begins/ends too small");
    out->n = st.n_out;
    if (mem == OVTK_MEM_HOST) {
        OVTK_HIP(hipMemcpyAsync(out->ragged_begins, d_rb, size_t(n_rows) * 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipMemcpyAsync(out->ragged_ends, d_re, size_t(n_rows) * 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipMemcpyAsync(out->begins, d_b, size_t(st.n_out) * 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipMemcpyAsync(out->ends, d_e, size_t(st.n_out) * 4, hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipMemcpyAsync(out->skips, d_sk, size_t(st.n_out), hipMemcpyDeviceToHost, s));
        OVTK_HIP(hipStreamSynchronize(s));
    }
    return OVTK_OK;
}

}  // extern "C"

#ifdef OVTK_PROBE
extern "C" __attribute__((visibility("default"))) int ovtk_debug_probe(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ts), sizeof(unsigned long long) * 8192 * 12);
    if (reset) {
        void* p = nullptr;
        (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_ts));
        (void)hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 12);
    }
    return 0;
}
#endif
