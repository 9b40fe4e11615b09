// tools/regex_host_check.cpp -- compile_regex's tables run on the host (the loop of regex_device.hpp's regex_next_match, restated) against the
// system's PCRE2 on the same subjects: every successive match, as RegexSplit iterates them.  Test infrastructure (tools/fuzz_regex_host.py drives
// it: thousands of subjects per pattern in milliseconds, where the emulator build takes seconds); the image has libpcre2-8.so.0 but no pcre2.h,
// hence the hand-written prototypes.
//   g++ -std=c++17 -O1 -Iopenvino_tokenizers_amd/csrc tools/regex_host_check.cpp openvino_tokenizers_amd/csrc/regex_compile.cpp \
//       /usr/lib/x86_64-linux-gnu/libpcre2-8.so.0 -o tools/build/regex_host_check
//   tools/build/regex_host_check PATTERN SUBJECT...     exit 0 same, 1 different, 2 outside the subset, 3 PCRE2 rejects it and so does compile_regex
#include <cstddef>
#include <cstdint>
extern "C" {
typedef struct pcre2_real_code_8 pcre2_code; typedef struct pcre2_real_match_data_8 pcre2_match_data; typedef size_t PCRE2_SIZE; typedef const uint8_t* PCRE2_SPTR;
pcre2_code* pcre2_compile_8(PCRE2_SPTR, PCRE2_SIZE, uint32_t, int*, PCRE2_SIZE*, void*);
pcre2_match_data* pcre2_match_data_create_from_pattern_8(const pcre2_code*, void*);
int pcre2_match_8(const pcre2_code*, PCRE2_SPTR, PCRE2_SIZE, PCRE2_SIZE, uint32_t, pcre2_match_data*, void*);
PCRE2_SIZE* pcre2_get_ovector_pointer_8(pcre2_match_data*);
}
#define pcre2_compile pcre2_compile_8
#define pcre2_match_data_create_from_pattern pcre2_match_data_create_from_pattern_8
#define pcre2_match pcre2_match_8
#define pcre2_get_ovector_pointer pcre2_get_ovector_pointer_8
#define PCRE2_ZERO_TERMINATED (~(PCRE2_SIZE)0)
#define PCRE2_UTF 0x00080000u
#define PCRE2_UCP 0x00020000u
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "regex_compile.hpp"
using namespace ovtk;
static int symbol(const RegexProgram& R, const std::string& s, int i, int& len) {
    const int slen = int(s.size());
    const uint32_t b = uint8_t(s[i]);
    if (b < 0x80) { len = 1; if (b == '\n' && i == slen - 1 && R.sym_final_nl >= 0) return R.sym_final_nl; return R.ascii_class[b]; }
    uint32_t cp = b; len = 1;
    if (b >= 0xC0) { int n = b >= 0xF0 ? 4 : (b >= 0xE0 ? 3 : 2); if (i + n > slen) n = slen - i; cp = b & (0xFFu >> (n + 1));
        for (; len < n && (uint8_t(s[i + len]) & 0xC0) == 0x80; ++len) cp = (cp << 6) | (uint8_t(s[i + len]) & 0x3F); }
    if (cp > 0x10FFFF) cp = 0x10FFFF;
    return R.cp_blocks[size_t(R.cp_index[cp >> 7]) * 128 + (cp & 127)];
}
static int step_back(const std::string& s, int lo, int i, int chars) {
    for (; chars > 0 && i > lo; --chars) { const int e = i; --i; while (i > lo && e - i < 4 && (uint8_t(s[i]) & 0xC0) == 0x80) --i; }
    return i;
}
static int context(const RegexProgram& R, const std::string& s, int p) {
    if (R.n_ctx <= 1 || p <= 0) return 0;
    int q = step_back(s, 0, p, R.behind_chars > 0 ? R.behind_chars : 1);
    int ctx = q == 0 ? 0 : 1;
    while (q < p) { int len = 0; int sym = symbol(R, s, q, len); if (sym == R.sym_final_nl) sym = R.ascii_class['\n']; ctx = R.ctx_next[size_t(ctx) * R.n_classes + sym]; q += len; }
    return ctx;
}
static bool next_match(const RegexProgram& R, const std::string& s, int start, int& mb, int& me) {
    const int slen = int(s.size());
    int p = start;
    while (p <= slen) {
        int state = R.start[context(R, s, p)];
        int i = p, last = -1, first_len = 1;
        for (;;) {
            int len = 0;
            const int sym = i < slen ? symbol(R, s, i, len) : R.sym_eot;
            if (i == p) first_len = len;
            const uint32_t t = R.trans[size_t(state) * R.n_syms + sym];
            if (t & kRegexMatchBit) last = step_back(s, p, i, (t >> kRegexDelayShift) & kRegexDelayMask);
            state = int(t & kRegexStateMask);
            if (state == 0 || i >= slen) break;
            i += len;
        }
        if (last >= 0) { mb = p; me = last; return true; }
        if (p >= slen) break;
        p += first_len;
    }
    return false;
}
int main(int argc, char** argv) {
    // argv[1] = pattern, rest = subjects; exit 0 same, 1 differ, 2 unsupported, 3 pcre2 rejects
    std::string pat = argv[1];
    RegexProgram R; std::string err;
    int rc = compile_regex(pat, R, err);
    int ec; PCRE2_SIZE eo;
    pcre2_code* re = pcre2_compile((PCRE2_SPTR)pat.c_str(), PCRE2_ZERO_TERMINATED, PCRE2_UTF | PCRE2_UCP, &ec, &eo, nullptr);
    if (rc) { printf("UNSUPPORTED %s\n", err.c_str()); return 2; }
    if (!re) { printf("PCRE2 rejects; ours invalid=%d\n", int(R.invalid)); return R.invalid ? 3 : 1; }
    if (R.invalid) { printf("ours INVALID (%s), pcre2 accepts\n", R.invalid_why.c_str()); return 1; }
    pcre2_match_data* md = pcre2_match_data_create_from_pattern(re, nullptr);
    int bad = 0;
    for (int a = 2; a < argc; ++a) {
        std::string s = argv[a];
        std::vector<std::pair<int,int>> ours, theirs;
        for (int st = 0;;) { int mb, me; if (!next_match(R, s, st, mb, me)) break; ours.push_back({mb, me}); if (me == mb || me >= int(s.size())) break; st = me; }
        bool gave_up = false;   // PCRE2_ERROR_MATCHLIMIT and the like: the backtracker ran out of steps -- the reference then reports "no match" (src/utils.cpp:411-413), a property of PCRE2's search order and limits, not of the pattern: such subjects are not compared
        for (size_t st = 0;;) { int r = pcre2_match(re, (PCRE2_SPTR)s.data(), s.size(), st, 0, md, nullptr); if (r < -1) gave_up = true; if (r < 0) break; PCRE2_SIZE* ov = pcre2_get_ovector_pointer(md);
            theirs.push_back({int(ov[0]), int(ov[1])}); if (ov[1] == ov[0] || ov[1] >= s.size()) break; st = ov[1]; }
        if (gave_up) continue;
        if (ours != theirs) { ++bad; printf("DIFF on '%s': ours", s.c_str()); for (auto& m : ours) printf(" [%d,%d)", m.first, m.second); printf(" pcre2"); for (auto& m : theirs) printf(" [%d,%d)", m.first, m.second); printf("\n"); }
    }
    if (!bad) printf("ok states=%d ctx=%d\n", R.n_states, R.n_ctx);
    return bad ? 1 : 0;
}
