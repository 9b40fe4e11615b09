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
// `$`, `\b`) and a small "what stands behind this position" context kept in the state (`^`, look-behind, `\b`).  A look-ahead
// over more than one character travels with its thread as a condition until the characters behind it have decided it; a match
// whose look-ahead was still open is reported late, with the number of characters it ended back (bits 12..14).
//
// Supported syntax (anything else is OVTK_E_UNSUPPORTED, never a wrong result): literals, `.`, escapes \t \n \r \f \v
// \e \a \0 \xHH \x{H..}, \Q..\E, classes [...] with ranges / negation / POSIX [:alpha:] [:digit:] ..., \d \s \w \h \v \N
// and their negations, \p{..} \P{..} \pL by General_Category and (round 5) by script -- \p{Han} \p{Hira} = Script_Extensions as in
// PCRE2 >= 10.40, \p{sc:Han} \p{script=Han} = Script, \p{scx:Han}; names matched loosely -- (Unicode 16.0, the version of the
// PCRE2 10.46 the reference pins; UCP meanings: \d = Nd, \s = Z + \h + \v, \w = L | N | Mn | Pc), groups ( ) (?: ) (?<name> ),
// (?i) (?s) (?m) (?i: ) (?s: ) -- caseless on ASCII letters incl. U+017F / U+212A; (?m): `^` / `$` at every line break --, alternation, * + ? {m} {m,} {m,n} {,n}
// greedy / lazy / possessive, atomic groups (?> ), ^ $ \A \z \Z \b \B \R, (?x), (?| ), quantified assertion groups, look-ahead (?= ) (?! )
// over anything that is decided within kRegexMaxDelay characters of a match's end, look-behind (?<= ) (?<! ) over alternatives of fixed
// sequences of up to kRegexMaxBehind characters.  Not supported: back-references, recursion, conditionals, \X \K \G, option verbs other
// than a leading (*UTF) / (*UCP), binary properties (\p{Alphabetic} ...), caseless matching of cased characters outside ASCII, an atomic
// group that can give characters back inside itself and holds (or stands in) a look-ahead of more than one character.
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

namespace ovtk {

constexpr int kRegexMaxStates = 4096;   // DFA states (a transition is a u16: bits 0..11 the next state)
constexpr int kRegexMaxClasses = 250;
constexpr int kRegexMaxCtx = 64;        // "what stands behind this position" contexts
constexpr int kRegexMaxBehind = 8;      // characters a look-behind may look at
constexpr int kRegexMaxDelay = 7;       // characters a match's end may lie behind the transition that reports it (a look-ahead that was still undecided)
constexpr uint16_t kRegexMatchBit = 0x8000;
constexpr uint16_t kRegexStateMask = 0x0FFF;
constexpr int kRegexDelayShift = 12;    // bits 12..14: the match reported by bit 15 ended this many CHARACTERS before the current position
constexpr uint16_t kRegexDelayMask = 0x7;

struct RegexProgram {
    // trans[state * n_syms + sym]: bits 0..11 = next state (0 = dead: stop), bit 15 = the pattern matches the text
    // consumed BEFORE this symbol -- less the last d characters of it, d = bits 12..14 (a match whose look-ahead was decided d characters late).  Symbols: character classes 0 .. n_classes-1, then sym_eot (end of subject), then
    // sym_final_nl (a '\n' that is the subject's last character; -1 when the pattern has no `$` / `\Z`).
    std::vector<uint16_t> trans;
    int n_states = 0, n_syms = 0, n_classes = 0;
    int sym_eot = 0, sym_final_nl = -1;
    // Start state by context.  The contexts are an automaton of their own over the classes: 0 = start of the subject; 1 = "nothing of
    // interest behind" -- where to start when the `behind_chars` characters in front of a position are re-read; then
    // ctx = ctx_next[ctx * n_classes + class] per character.  (n_ctx == 1: the pattern does not look behind.)
    int n_ctx = 1;
    int behind_chars = 0;
    uint16_t start[kRegexMaxCtx] = {};
    std::vector<uint8_t> ctx_next;       // [n_ctx * n_classes]
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
