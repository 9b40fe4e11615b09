// tests/emu/l3_flags_fuzz.cpp -- TEST INFRASTRUCTURE: differential fuzz of span_flags_l3 and span_flags_gpt2m (csrc/span_l3.hpp: the rule
// algebra of the Llama-3 and GPT-2 families on bit masks) against the literal matchers llama3_match_end / gpt2_match_end
// (csrc/split_device.hpp; the first pinned against PCRE2 by tests/test_split_rules.py, the second by the fused-encode tests of
// tests/test_span_kernel.py), on the SIMT emulator.
//
// One wave per case, as lookup_span_kernel calls it: a block of up to 2 048 bytes that holds whole rows, may begin inside a row (at a
// true piece start) and may be cut inside its last row (at_end = false: exactly 2 048 bytes, more text follows).  The truth is the
// literal matcher on the COMPLETE rows -- so the check covers the rule algebra, the row boundaries, and `und` (how far a cut block
// decides) against what really follows the block.
//     g++ -O1 -std=c++17 -I tests/emu -I openvino_tokenizers_amd/csrc tests/emu/l3_flags_fuzz.cpp -o tests/emu/build/l3_flags_fuzz
//     tests/emu/build/l3_flags_fuzz [first_seed] [last_seed] [cases_per_seed]
// Prints a line per seed; exits 1 at the first difference (with the text around it).
#include <hip/hip_runtime.h>

#include <random>
#include <string>

#include "span_fam.hpp"
#include "unicode_gc.inc"
#include "unicode_tables.inc"

using namespace ovtk;

struct Case {
    const uint8_t* chars;   // the block's bytes (and what follows them)
    int b_len;
    int at_end;
    const uint32_t* rs;     // [64]
    uint32_t* flags;        // [64] out
    int* und;               // out
    int* covered;           // out: 0 = the algebra declined (odd)
    int family;             // 0: the Llama-3 family (span_flags_l3), 1 / 2: the GPT-2 family (span_flags_gpt2m<false / true>),
                            // 3: DeepSeek-V3's pattern (span_flags_ds3), 4: o200k_base (span_flags_o200k)
};

static __global__ void l3_case_kernel(Case c, SplitDev sp) {
    __shared__ uint32_t text_w[kWave * 8 + 16];
    __shared__ uint32_t scratch[(kSpanFamScratch > kSpanClassScratch ? kSpanFamScratch : kSpanClassScratch) / 4];
    const int l = lane_id();
    uint32_t x[8];
    for (int j = 0; j < 8; ++j) {
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) v |= uint32_t(c.chars[32 * l + 4 * j + k]) << (8 * k);   // (the buffer is padded: bytes behind the block are real)
        x[j] = v;
        text_w[8 * l + j] = v;
    }
    wave_sync();
    const int nv = c.b_len - 32 * l;
    const uint32_t vm = nv >= 32 ? ~0u : (nv <= 0 ? 0u : ((1u << nv) - 1u));
    uint32_t fl = 0;
    int und = 0;
    bool ok = true;
    if (c.family == 3) {
        ok = span_flags_ds3(x, c.rs[l], vm, reinterpret_cast<const uint8_t*>(text_w), scratch, sp, c.at_end != 0, c.b_len, fl, und);
    } else if (c.family == 4) {
        ok = span_flags_o200k(x, c.rs[l], vm, reinterpret_cast<const uint8_t*>(text_w), scratch, sp, c.at_end != 0, c.b_len, fl, und);
    } else if (c.family == 0) {
        ok = span_flags_l3(x, c.rs[l], vm, reinterpret_cast<const uint8_t*>(text_w), scratch, sp, c.at_end != 0, c.b_len, fl, und);
    } else {
        if (c.family == 1) span_flags_gpt2m<false>(x, c.rs[l], vm, reinterpret_cast<const uint8_t*>(text_w), scratch, sp, c.at_end != 0, c.b_len, fl);
        else span_flags_gpt2m<true>(x, c.rs[l], vm, reinterpret_cast<const uint8_t*>(text_w), scratch, sp, c.at_end != 0, c.b_len, fl);
        und = c.at_end ? c.b_len : c.b_len - 8;   // (lookup_span_kernel's kSpanHalo)
    }
    c.flags[l] = fl;
    if (l == 0) {
        *c.und = und;
        *c.covered = ok ? 1 : 0;
    }
}

static std::vector<uint8_t> g_flat, g_cls4;
static SplitDev make_split(int digits1, int tail_ws) {
    SplitDev sp{};
    sp.kind = kSplitLlama3;
    sp.uc_index = kUcIndex;
    sp.uc_blocks = kUcBlocks;
    if (g_flat.empty()) {
        g_flat.assign(kUcFlatLimit / 4, 0);
        for (uint32_t cp = 0; cp < kUcFlatLimit; ++cp) {
            const uint32_t b = kUcBlocks[size_t(kUcIndex[cp >> 7]) * 64 + ((cp & 127) >> 1)];
            const uint32_t nib = (cp & 1) ? (b >> 4) : (b & 15u);
            g_flat[cp >> 2] |= uint8_t((nib & 3u) << (2 * (cp & 3u)));
        }
    }
    sp.uc_flat = g_flat.data();
    if (g_cls4.empty()) {   // api_encode.cpp's unicode_tables(): General_Category folded to seven classes, white space from the \s bit
        g_cls4.assign(0x110000 / 2, 0);
        for (unsigned i = 0; i < kGcRanges; ++i)
            for (unsigned cp = kGcStart[i]; cp < kGcStart[i + 1] && cp < 0x110000u; ++cp) {
                const uint32_t b = kUcBlocks[size_t(kUcIndex[cp >> 7]) * 64 + ((cp & 127) >> 1)];
                const uint32_t nib = (cp & 1) ? (b >> 4) : (b & 15u);
                const uint32_t c = (nib & 3u) == kClsS ? kC4Space : uint32_t(kGcToC4[kGcValue[i]]);
                g_cls4[cp >> 1] |= uint8_t(c << (4 * (cp & 1u)));
            }
    }
    sp.uc_cls4 = g_cls4.data();
    sp.l3_digits1 = digits1;
    sp.l3_tail_ws = tail_ws;
    return sp;
}

static const char* kFrag[] = {
    "the", "token", "a", "I", "x", "Zq", "hello", "World", " ", " ", " ", "  ", "   ", "\n", "\n", "\r\n", "\r", "\t", "\n\n", " \n", "\n ", " \n ", "\n    ", "\x0b", "\x0c",
    ",", ".", "!", "?", "!?", "...", "--", "(", ")", "\"", "'", "''", "$", "%", "#", "@", "/", "\\", "_", "-", "*",
    "'s", "'t", "'m", "'d", "'re", "'ve", "'ll", "'S", "'T", "'RE", "'Ve", "'lL", "'r", "'v", "'l", "'x", "'", "don't", "we'll", "I'M", "it's",
    "1", "12", "123", "1234", "12345", "123456", "1234567", "0", "7", "2024", "3.14", "1,000", "a1b2", "9x",
    "\xC3\xA9", "na\xC3\xAFve", "stra\xC3\x9F" "e", "\xC3\x97", "\xC2\xA0", "\xC2\xA0\xC2\xA0", "\xC2\xAB", "\xC2\xBB", "\xE2\x80\x94", "\xE2\x80\xA6", "\xE2\x82\xAC",
    "\xE2\x80\xA8", "\xE2\x80\x83", "\xE3\x80\x80", "\xC2\x85", "\xD0\xBF\xD1\x80\xD0\xB8\xD0\xB2\xD0\xB5\xD1\x82", "\xCE\xA9\xCE\xBC\xCE\xAD\xCE\xB3\xCE\xB1",
    "\xE6\x97\xA5\xE6\x9C\xAC\xE8\xAA\x9E", "\xE3\x81\xAE", "\xE3\x80\x82", "\xF0\x9F\x98\x80", "\xF0\x9F\x98\x80\xF0\x9F\x98\x81", "\xF0\x90\x90\x80", "\xE2\x84\xAA",
    "\xEF\xBC\x81", "\xD7\xA9\xD7\x9C\xD7\x95\xD7\x9D", "\xD8\xB3\xD9\x84\xD8\xA7\xD9\x85",
};
// what the families of span_fam.hpp tell apart and the others do not: case, marks (U+0301, U+20DD, U+0903), titlecase (U+01C5), modifier letters
// (U+02B0), slashes behind line breaks, control and format characters, upper-case runs
static const char* kFamFrag[] = {
    "A", "B", "AB", "ABC", "HTTP", "Camel", "camelCase", "XMLHttpRequest", "iPhone", "aB", "Ab", "aBc", "ABc", "abC", "A1", "Z",
    "\xCC\x81", "\xCC\x81\xCC\x81", "e\xCC\x81", "E\xCC\x81", "\xCC\x81" "a", "\xCC\x81" "A", "!\xCC\x81", "!!\xCC\x81", " \xCC\x81", "\xE2\x83\x9D", "\xE0\xA4\x83",
    "\xC7\x85", "\xCA\xB0", "\xE6\x97\xA5", "\xE6\x97\xA5" "A", "A\xE6\x97\xA5", "A\xE6\x97\xA5" "B", "\xE6\x97\xA5" "Ab", "\xD0\x9F\xD1\x80", "\xD0\x9F\xD0\xA0", "\xD0\xBF\xD0\xA0",
    "\xC3\x89", "\xC3\x89t\xC3\xA9", "\xCE\xA9", "'s", "'S", "'ll", "'LL", "A's", "a'T", "B'Re", "\xE6\x97\xA5's", "\xCC\x81's", "'\xCC\x81",
    "/", "//", "\n/", "\n//", "!\n/", "*/\n/*", "/\n", "\n/\n", "\r\n/", "!\n/!\n/a", "*/", "/*",
    "\x01", "\x7F", "\x1B[0m", "\xC2\xAD", "\xE2\x80\x8B", "\xE2\x80\x8D", "\xEF\xBB\xBF", "\xC2\x80", "\xF3\xA0\x80\x81",
    "1a", "1A", "12ab", "a1", "\xC2\xAD" "a", "\xE2\x80\x8B" "B", "\x01" "a", "!a", "!ab", "!A", "!\xC3\xA9", "!a\xC3\xA9", "?b\xCC\x81", "#tag", "@user", "$x", "_id", "-v", "(a", " !a", "!!a", ".com",
};
// fragments the algebra does not cover (a non-ASCII \p{N}, U+017F) and broken UTF-8: rarer
static const char* kOddFrag[] = {"\xC2\xB2", "\xD9\xA3", "\xEF\xBC\x91", "\xC5\xBF", "'\xC5\xBF", "\xC2\xBD", "\xD9\xA3" "a", "a'\xC5\xBF"};
static const char* kBadFrag[] = {"\x80", "\xBF\xBF", "\xC3", "\xE2\x82", "\xF0\x9F\x98", "\xFF", "\xC0\x80", "\xE2", "\xF8\x88\x80\x80"};

struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t s) : g(s) {}
    int below(int n) { return int(g() % uint64_t(n)); }
    bool chance(int pct) { return below(100) < pct; }
};

static bool g_fam_frags = false;
static std::string make_row(Rng& r, int target, int odd_pct, int bad_pct) {
    std::string s;
    const int style = r.below(8);
    while (int(s.size()) < target) {
        if (r.chance(odd_pct)) s += kOddFrag[r.below(sizeof kOddFrag / sizeof *kOddFrag)];
        else if (r.chance(bad_pct)) s += kBadFrag[r.below(sizeof kBadFrag / sizeof *kBadFrag)];
        else if (style == 0) s += std::string(1 + r.below(40), "0123456789"[r.below(10)]);          // long digit runs
        else if (style == 1) s += std::string(1 + r.below(6), "\n\r \t"[r.below(4)]);               // white space of every kind
        else if (style == 2 && r.chance(50)) s += std::string(1 + r.below(70), " \n"[r.below(2)]);  // long runs
        else if (style == 3 && r.chance(40)) s += std::string(1 + r.below(50), "ABCXYZ"[r.below(6)]);  // upper-case runs
        else if (g_fam_frags && r.chance(55)) s += kFamFrag[r.below(sizeof kFamFrag / sizeof *kFamFrag)];
        else s += kFrag[r.below(sizeof kFrag / sizeof *kFrag)];
        if (style >= 5 && r.chance(60)) s += ' ';
    }
    return s;
}

// piece starts of the row s from byte `from` (a true piece start) on
static int g_family = 0;
static void truth_starts(const SplitDev& sp, const std::string& s, int from, std::vector<uint8_t>& is_start) {
    is_start.assign(s.size() + 1, 0);
    const uint8_t* p = reinterpret_cast<const uint8_t*>(s.data());
    for (int q = from; q < int(s.size());) {
        is_start[q] = 1;
        q = g_family == 0 ? llama3_match_end(sp, p, int(s.size()), q)
            : g_family == 3 ? ds3_match_end(sp, p, int(s.size()), q)
            : g_family == 4 ? o200k_match_end(sp, p, int(s.size()), q)
                            : gpt2_match_end(sp, p, int(s.size()), q, g_family == 2);
    }
}

int main(int argc, char** argv) {
    const int lo = argc > 1 ? atoi(argv[1]) : 0, hi = argc > 2 ? atoi(argv[2]) : 4, per_seed = argc > 3 ? atoi(argv[3]) : 300;
    const int only = argc > 4 ? atoi(argv[4]) : -1;   // one family only (0..4)
    long long n_cases = 0, n_odd = 0, n_cut = 0, und_sum = 0;
    for (int seed = lo; seed < hi; ++seed) {
        Rng r(0x9E3779B97F4A7C15ull * uint64_t(seed + 1));
        for (int it = 0; it < per_seed; ++it) {
            const SplitDev sp = make_split(r.chance(25), r.chance(30));
            {   // a quarter of the cases each: the GPT-2 family's mask form, DeepSeek-V3's pattern, o200k_base
                const int pick = r.below(4);
                g_family = pick == 0 ? 1 + r.below(2) : (pick == 1 ? 3 : (pick == 2 ? 4 : 0));
                if (only >= 0) g_family = only;
            }
            g_fam_frags = g_family >= 3;
            const int odd_pct = r.chance(15) ? 3 : 0, bad_pct = r.chance(15) ? 4 : 0;
            // rows until the block is full (or, for a block that ends with its text, until a random length)
            const bool want_cut = r.chance(45);
            const int want = want_cut ? 2048 + 1 + r.below(600) : 1 + r.below(2048);
            std::vector<std::string> rows;
            int total = 0;
            // the block may begin inside a row: a long first row, entered at one of its true piece starts
            int first_from = 0;
            while (total < want) {
                const int kind = r.below(10);
                int target = kind < 3 ? 1 + r.below(12) : (kind < 8 ? 20 + r.below(500) : 600 + r.below(2400));
                if (!want_cut && total + target > want) target = want - total;
                std::string row = make_row(r, target, odd_pct, bad_pct);
                if (!want_cut && total + int(row.size()) > 2048) row.resize(2048 - total);
                if (row.empty()) continue;
                if (rows.empty() && r.chance(30) && row.size() > 8) {
                    std::vector<uint8_t> st;
                    truth_starts(sp, row, 0, st);
                    std::vector<int> cand;
                    for (int q = 1; q < int(row.size()); ++q)
                        if (st[q]) cand.push_back(q);
                    if (!cand.empty()) first_from = cand[r.below(int(cand.size()))];
                }
                total += int(row.size()) - (rows.empty() ? first_from : 0);
                rows.push_back(row);
            }
            // the block's bytes: rows back to back from first_from on; truth per byte
            std::vector<uint8_t> chars, truth, row_start;
            for (size_t i = 0; i < rows.size(); ++i) {
                std::vector<uint8_t> st;
                const int from = i == 0 ? first_from : 0;
                truth_starts(sp, rows[i], from, st);
                for (int q = from; q < int(rows[i].size()); ++q) {
                    chars.push_back(uint8_t(rows[i][q]));
                    truth.push_back(st[q]);
                    row_start.push_back(q == 0 ? 1 : 0);   // (a block that begins inside a row: no row start at its first byte)
                }
            }
            const bool at_end = int(chars.size()) <= 2048;
            const int b_len = at_end ? int(chars.size()) : 2048;
            uint32_t rs[64] = {0};
            for (int q = 0; q < b_len; ++q)
                if (row_start[q]) rs[q >> 5] |= 1u << (q & 31);
            if (at_end && b_len < 2048) rs[b_len >> 5] |= 1u << (b_len & 31);   // (the kernel's sentinel behind the text)
            chars.resize(chars.size() + 4096, uint8_t(r.chance(50) ? 'e' : 0xBF));   // what lies behind the text must not matter
            if (at_end)   // ... nor must what follows a block that ends with its text
                for (int q = b_len; q < b_len + 64; ++q) chars[q] = uint8_t("el 'stx\n1\xA9\xC3"[r.below(11)]);
            uint32_t flags[64];
            int und = -1, covered = -1;
            Case c{chars.data(), b_len, at_end ? 1 : 0, rs, flags, &und, &covered, g_family};
            hipLaunchKernelGGL(l3_case_kernel, 1, 64, 0, nullptr, c, sp);
            ++n_cases;
            if (!covered) {
                if (!odd_pct && !bad_pct) {   // the text holds nothing the algebra does not cover: it must not decline
                    printf("DECLINED without a reason: seed %d case %d b_len %d at_end %d\n", seed, it, b_len, int(at_end));
                    return 1;
                }
                ++n_odd;
                continue;
            }
            if (!at_end) {
                ++n_cut;
                und_sum += und;
            }
            bool bad = und > b_len || (at_end && und != b_len) || und < 1;   // (the first byte is always decided: the kernel counts on it)
            int where = -1;
            for (int q = 0; q < und && q < b_len && !bad; ++q) {
                const int got = (flags[q >> 5] >> (q & 31)) & 1u;
                const int want_bit = q == 0 ? 1 : int(truth[q] | row_start[q]);
                if (got != want_bit) {
                    bad = true;
                    where = q;
                }
            }
            if (bad) {
                printf("DIFFERENCE seed %d case %d family %d: b_len %d at_end %d und %d digits1 %d tail_ws %d at byte %d (got %d)\n", seed, it, g_family, b_len, int(at_end), und,
                       sp.l3_digits1, sp.l3_tail_ws, where, where >= 0 ? int((flags[where >> 5] >> (where & 31)) & 1u) : -1);
                if (where >= 0) {
                    const int a = where > 24 ? where - 24 : 0, b = where + 48 < int(chars.size()) ? where + 48 : int(chars.size());
                    printf("  bytes [%d, %d):", a, b);
                    for (int q = a; q < b; ++q) printf("%s%02x%s", q == where ? " [" : " ", chars[q], q == where ? "]" : "");
                    printf("\n  truth        :");
                    for (int q = a; q < b; ++q) printf("  %c", q < int(truth.size()) ? (row_start[q] ? 'R' : (truth[q] ? '1' : '.')) : '-');
                    printf("\n  flags        :");
                    for (int q = a; q < b; ++q) printf("  %c", q < 2048 ? (((flags[q >> 5] >> (q & 31)) & 1u) ? '1' : '.') : '-');
                    printf("\n");
                }
                return 1;
            }
        }
        printf("seed %d ok\n", seed);
        fflush(stdout);
    }
    printf("%lld cases, %lld left to the literal matcher (odd), %lld cut blocks deciding %.1f bytes on average\n", n_cases, n_odd, n_cut,
           n_cut ? double(und_sum) / double(n_cut) : 0.0);
    return 0;
}
