// regex_compile.hpp -- host-side compiler of RegexSplit patterns into tables a GPU lane can run.
//
// The reference hands the pattern to PCRE2 (PCRE2_UTF | PCRE2_UCP, src/utils.cpp:256-272) and asks for one match at a
// time (PCRE2Wrapper::match, src/utils.cpp:396-420).  PCRE2 is not run on the device; create() compiles the pattern
// once into
//   * a partition of the code points into classes (every character set of the pattern is a union of classes), as a
//     128-entry ASCII table + a two-level table for the rest, and
//   * a DFA over those classes whose states are ORDERED lists of positions of the pattern's backtracking automaton --
//     the order is PCRE2's priority (first alternative, greedy before skip), positions behind an accepting one are cut,
//     so the last accepting point the DFA passes before it dies is the end of exactly the match PCRE2's backtracking
//     finds from that start position (leftmost-first, not leftmost-longest).
// Zero-width assertions are resolved while a transition is computed, from the class of the next character (look-ahead,
// `$`, `\b`) and a small "what was the previous character" context kept in the state (`^`, look-behind, `\b`).
//
// Supported syntax (anything else is OVTK_E_UNSUPPORTED, never a wrong result): literals, `.`, escapes \t \n \r \f \v
// \e \a \0 \xHH \x{H..}, \Q..\E, classes [...] with ranges / negation / POSIX [:alpha:] [:digit:] ..., \d \s \w \h \v \N
// and their negations, \p{..} \P{..} \pL by General_Category and (round 5) by script -- \p{Han} \p{Hira} = Script_Extensions as in
// PCRE2 >= 10.40, \p{sc:Han} \p{script=Han} = Script, \p{scx:Han}; names matched loosely -- (Unicode 16.0, the version of the
// PCRE2 10.46 the reference pins; UCP meanings: \d = Nd, \s = Z + \h + \v, \w = L | N | Mn | Pc), groups ( ) (?: ) (?<name> ),
// (?i) (?s) (?m) (?i: ) (?s: ) -- caseless on ASCII letters incl. U+017F / U+212A; (?m): `^` / `$` at every line break --, alternation, * + ? {m} {m,} {m,n} {,n}
// greedy / lazy, possessive on single-character atoms, ^ $ \A \z \Z \b \B, look-ahead / look-behind (?= ) (?! ) (?<= )
// (?<! ) on one character.  Not supported: back-references, recursion, atomic groups, conditionals, look-around over more than
// one character, (?x), \R \X \K \G, binary properties (\p{Alphabetic} ...).
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

namespace ovtk {

constexpr int kRegexMaxStates = 4096;   // DFA states (a transition is a u16: bit 15 = "a match ends here")
constexpr int kRegexMaxClasses = 250;
constexpr int kRegexMaxCtx = 16;        // "previous character" contexts
constexpr uint16_t kRegexMatchBit = 0x8000;
constexpr uint16_t kRegexStateMask = 0x7FFF;

struct RegexProgram {
    // trans[state * n_syms + sym]: bits 0..14 = next state (0 = dead: stop), bit 15 = the pattern matches the text
    // consumed BEFORE this symbol.  Symbols: character classes 0 .. n_classes-1, then sym_eot (end of subject), then
    // sym_final_nl (a '\n' that is the subject's last character; -1 when the pattern has no `$` / `\Z`).
    std::vector<uint16_t> trans;
    int n_states = 0, n_syms = 0, n_classes = 0;
    int sym_eot = 0, sym_final_nl = -1;
    // Start state by context: ctx 0 = start of the subject, otherwise ctx_of_class[class of the previous character].
    int n_ctx = 1;
    uint16_t start[kRegexMaxCtx] = {};
    std::vector<uint8_t> ctx_of_class;   // [n_classes]
    uint8_t ascii_class[128] = {};
    std::vector<uint16_t> cp_index;      // [0x110000 >> 7]
    std::vector<uint8_t> cp_blocks;      // [n_blocks * 128]
    bool can_match_empty = false;        // some start state accepts before consuming anything
    bool invalid = false;                // PCRE2 itself rejects the pattern: the program never matches (the reference's null pattern,
    std::string invalid_why;             // src/utils.cpp:264-271, 397-399: every string passes through unsplit)
};

// 0, or OVTK_E_UNSUPPORTED with `err` naming what is outside the subset.  A pattern that PCRE2 itself rejects (an unmatched parenthesis,
// a quantifier without an operand, a range out of order ...) compiles -- to a program that never matches, RegexProgram::invalid.
int compile_regex(const std::string& pattern, RegexProgram& out, std::string& err);

// General_Category of every code point, as its index in unicode_gc.inc's order (Cn Lu Ll Lt Lm Lo Mn Mc Me Nd Nl No Pc Pd Ps Pe Pi Pf Po
// Sm Sc Sk So Zs Zl Zp Cc Cf Cs Co): gc[0x110000].  For the class table of span_fam.hpp's scans.
void unicode_general_categories(std::vector<uint8_t>& gc);

}  // namespace ovtk
