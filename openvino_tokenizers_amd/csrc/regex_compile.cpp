// regex_compile.cpp -- see regex_compile.hpp.  Plain host C++ (no HIP).
#include "regex_compile.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <climits>
#include <set>
#include <tuple>
#include <utility>

#include "../../include/ovtk_amd.h"
#include <cctype>

#include "unicode_gc.inc"
#include "unicode_scripts.inc"

namespace ovtk {
namespace {

// ---------------------------------------------------------------------------------------------- character sets
constexpr uint32_t kMaxCp = 0x10FFFF;
enum Gc : int {
    Cn, Lu, Ll, Lt, Lm, Lo, Mn, Mc, Me, Nd, Nl, No, Pc, Pd, Ps, Pe, Pi, Pf, Po, Sm, Sc, Sk, So, Zs, Zl, Zp, Cc, Cf, Cs, Co, kGcCount
};
const char* const kGcNames[kGcCount] = {"Cn", "Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Mc", "Me", "Nd", "Nl", "No", "Pc", "Pd", "Ps",
                                        "Pe", "Pi", "Pf", "Po", "Sm", "Sc", "Sk", "So", "Zs", "Zl", "Zp", "Cc", "Cf", "Cs", "Co"};
constexpr uint32_t gc_bit(int g) { return 1u << g; }
constexpr uint32_t kMaskL = gc_bit(Lu) | gc_bit(Ll) | gc_bit(Lt) | gc_bit(Lm) | gc_bit(Lo);
constexpr uint32_t kMaskM = gc_bit(Mn) | gc_bit(Mc) | gc_bit(Me);
constexpr uint32_t kMaskN = gc_bit(Nd) | gc_bit(Nl) | gc_bit(No);
constexpr uint32_t kMaskP = gc_bit(Pc) | gc_bit(Pd) | gc_bit(Ps) | gc_bit(Pe) | gc_bit(Pi) | gc_bit(Pf) | gc_bit(Po);
constexpr uint32_t kMaskS = gc_bit(Sm) | gc_bit(Sc) | gc_bit(Sk) | gc_bit(So);
constexpr uint32_t kMaskZ = gc_bit(Zs) | gc_bit(Zl) | gc_bit(Zp);
constexpr uint32_t kMaskC = gc_bit(Cn) | gc_bit(Cc) | gc_bit(Cf) | gc_bit(Cs) | gc_bit(Co);

struct CharSet {
    std::vector<std::pair<uint32_t, uint32_t>> r;  // inclusive ranges; sorted and disjoint after normalize()
    void add(uint32_t lo, uint32_t hi) { r.emplace_back(lo, hi); }
    void add(const CharSet& o) { r.insert(r.end(), o.r.begin(), o.r.end()); }
    void normalize() {
        std::sort(r.begin(), r.end());
        std::vector<std::pair<uint32_t, uint32_t>> out;
        for (const auto& x : r) {
            if (!out.empty() && x.first <= out.back().second + 1) out.back().second = std::max(out.back().second, x.second);
            else out.push_back(x);
        }
        r.swap(out);
    }
    CharSet negated() const {  // of a normalized set
        CharSet n;
        uint32_t next = 0;
        for (const auto& x : r) {
            if (x.first > next) n.add(next, x.first - 1);
            next = x.second + 1;
        }
        if (next <= kMaxCp) n.add(next, kMaxCp);
        return n;
    }
    bool has(uint32_t cp) const {
        for (const auto& x : r)
            if (cp >= x.first && cp <= x.second) return true;
        return false;
    }
    CharSet intersected(CharSet o) const {   // a & b = ~(~a | ~b)
        CharSet a = *this;
        a.normalize();
        o.normalize();
        CharSet u = a.negated();
        u.add(o.negated());
        u.normalize();
        return u.negated();
    }
    CharSet minus(CharSet o) const {
        o.normalize();
        return intersected(o.negated());
    }
};

int gc_of(uint32_t cp) {
    const unsigned* p = std::upper_bound(kGcStart, kGcStart + kGcRanges + 1, cp);
    return kGcValue[(p - kGcStart) - 1];
}
// Script / Script_Extensions of a script name (long name or ISO 15924 code, matched loosely as PCRE2 does: case, `_`, `-` and spaces do
// not count); false: no such script.  unicode_scripts.inc (Unicode 16.0; tools/gen_unicode_scripts.py).
bool script_set(const std::string& loose_name, bool extensions, CharSet& out);

CharSet gc_set(uint32_t mask) {
    CharSet s;
    for (unsigned i = 0; i < kGcRanges; ++i)
        if (mask >> kGcValue[i] & 1u) s.add(kGcStart[i], kGcStart[i + 1] - 1);
    s.normalize();
    return s;
}

bool script_set(const std::string& loose_name, bool extensions, CharSet& out) {
    auto loose = [](const char* t) {
        std::string r;
        for (; *t; ++t)
            if (*t != '_' && *t != '-' && *t != ' ') r.push_back(char(std::tolower(static_cast<unsigned char>(*t))));
        return r;
    };
    for (unsigned i = 0; i < kScriptCount; ++i) {
        if (loose(kScriptNames[i][0]) != loose_name && loose(kScriptNames[i][1]) != loose_name) continue;
        const unsigned* offs = extensions ? kScxOffsets : kScOffsets;
        const unsigned* rs = extensions ? kScxRanges : kScRanges;
        for (unsigned k = offs[i]; k < offs[i + 1]; ++k) out.add(rs[2 * k], rs[2 * k + 1] - 1);
        return true;
    }
    return false;
}
CharSet hspace_set() {  // PCRE2 \h
    CharSet s;
    for (uint32_t c : {0x09u, 0x20u, 0xA0u, 0x1680u, 0x180Eu, 0x202Fu, 0x205Fu, 0x3000u}) s.add(c, c);
    s.add(0x2000, 0x200A);
    s.normalize();
    return s;
}
CharSet vspace_set() {  // PCRE2 \v
    CharSet s;
    s.add(0x0A, 0x0D);
    s.add(0x85, 0x85);
    s.add(0x2028, 0x2029);
    s.normalize();
    return s;
}
CharSet space_set() {  // \s under UCP: property Z, or \h, or \v
    CharSet s = gc_set(kMaskZ);
    s.add(hspace_set());
    s.add(vspace_set());
    s.normalize();
    return s;
}
CharSet word_set() {  // \w under UCP in PCRE2 >= 10.43 (the reference pins 10.46): L, N, Mn, Pc
    return gc_set(kMaskL | kMaskN | gc_bit(Mn) | gc_bit(Pc));
}

// ---------------------------------------------------------------------------------------------- syntax tree
enum NodeKind { kEmpty, kSet, kCat, kAlt, kRepeat, kAssert, kLookNode, kAtomic };   // kLookNode: (?=X) / (?!X) over more than one character, kids = {X};
                                                                                     // kAtomic: (?>X) that Parser::atomize could not spell in plain syntax, kids = {X}
enum AssertKind { kBot, kEot, kEotNl, kWordB, kNotWordB, kAhead, kBehind };
struct Node {
    NodeKind kind = kEmpty;
    CharSet set;            // kSet; kAssert kAhead / kBehind
    std::vector<int> kids;  // kCat, kAlt; kRepeat: one
    int min = 0, max = 0;   // kRepeat; max < 0: unbounded
    int mode = 0;           // kRepeat: 0 greedy, 1 lazy, 2 possessive
    int akind = 0;
    bool aneg = false;      // kAssert, kLookNode: negative
    std::vector<std::vector<CharSet>> behind;   // kAssert kBehind over more than one character: the alternatives, each a sequence of sets
};

struct Unsupported {
    std::string what;
};
// What PCRE2 itself rejects (pcre2_compile returns NULL): the reference then keeps a null pattern and every match "fails"
// (src/utils.cpp:264-271, 397-399) -- RegexSplit passes every string through unsplit.  Only the errors that are certain are thrown
// as Invalid; whatever PCRE2 might accept (a property name this table does not know, a group option it does) stays Unsupported.
struct Invalid {
    std::string what;
};

struct Flags {
    bool caseless = false, dotall = false, multiline = false, extended = false;
};

class Parser {
public:
    Parser(const std::string& utf8, std::vector<Node>& nodes) : nodes_(nodes) {
        // the pattern as code points (the subject is UTF-8, the pattern too: PCRE2_UTF)
        for (size_t i = 0; i < utf8.size();) {
            const uint8_t b = uint8_t(utf8[i]);
            int n = b < 0x80 ? 1 : (b >= 0xF0 ? 4 : (b >= 0xE0 ? 3 : (b >= 0xC0 ? 2 : 0)));
            if (n == 0 || i + size_t(n) > utf8.size()) throw Invalid{"pattern is not valid UTF-8"};
            uint32_t cp = n == 1 ? b : (b & (0xFFu >> (n + 1)));
            for (int k = 1; k < n; ++k) {
                const uint8_t c = uint8_t(utf8[i + size_t(k)]);
                if ((c & 0xC0) != 0x80) throw Invalid{"pattern is not valid UTF-8"};
                cp = (cp << 6) | (c & 0x3F);
            }
            p_.push_back(cp);
            i += size_t(n);
        }
    }
    int parse() {
        Flags f;
        for (;;) {   // leading option settings that ask for what the reference sets anyway (PCRE2_UTF | PCRE2_UCP, src/utils.cpp:256-262)
            if (looking_at("(*UTF)")) pos_ += 6;
            else if (looking_at("(*UCP)")) pos_ += 6;
            else break;
        }
        const int root = parse_alt(f);
        if (pos_ != p_.size()) throw Invalid{"unmatched ')'"};
        return root;
    }

private:
    std::vector<Node>& nodes_;
    std::vector<uint32_t> p_;
    size_t pos_ = 0;
    std::vector<std::string> group_names_;   // named groups seen so far (PCRE2 rejects a second use of a name)

    bool more() const { return pos_ < p_.size(); }
    uint32_t peek(size_t k = 0) const { return pos_ + k < p_.size() ? p_[pos_ + k] : 0xFFFFFFFFu; }
    bool eat(uint32_t c) {
        if (peek() == c) { ++pos_; return true; }
        return false;
    }
    bool looking_at(const char* s) const {
        for (size_t k = 0; s[k]; ++k)
            if (peek(k) != uint32_t(uint8_t(s[k]))) return false;
        return true;
    }
    int add(Node n) {
        nodes_.push_back(std::move(n));
        return int(nodes_.size()) - 1;
    }
    int set_node(CharSet s) {
        s.normalize();
        Node n;
        n.kind = kSet;
        n.set = std::move(s);
        return add(std::move(n));
    }

    // Caseless closure of explicitly written characters: ASCII letters, plus the two non-ASCII characters that fold to
    // one (PCRE2 with UTF|UCP: 's' ~ U+017F, 'k' ~ U+212A).  Cased characters beyond ASCII are refused.
    static void close_caseless(CharSet& s) {
        s.normalize();
        CharSet extra;
        for (const auto& x : s.r) {
            if (x.second >= 0x80 && x.second - std::max<uint32_t>(x.first, 0x80) > 0x40000) throw Unsupported{"(?i) over a wide range"};
            for (uint32_t cp = std::max<uint32_t>(x.first, 0x80); cp <= x.second; ++cp) {
                const int g = gc_of(cp);
                if (g == Lu || g == Ll || g == Lt || cp == 0x345) throw Unsupported{"(?i) on a cased character outside ASCII"};
            }
            const uint32_t lo = x.first, hi = x.second;
            auto clip = [&](uint32_t a, uint32_t b, int delta) {
                const uint32_t l = std::max(lo, a), h = std::min(hi, b);
                if (l <= h) extra.add(uint32_t(int(l) + delta), uint32_t(int(h) + delta));
            };
            clip('A', 'Z', 32);
            clip('a', 'z', -32);
        }
        s.add(extra);
        s.normalize();
        if (s.has('s')) s.add(0x17F, 0x17F);
        if (s.has('k')) s.add(0x212A, 0x212A);
        s.normalize();
    }

    int parse_alt(Flags& f) {
        std::vector<int> alts;
        alts.push_back(parse_cat(f));
        while (eat('|')) alts.push_back(parse_cat(f));
        if (alts.size() == 1) return alts[0];
        Node n;
        n.kind = kAlt;
        n.kids = std::move(alts);
        return add(std::move(n));
    }

    int parse_cat(Flags& f) {
        std::vector<int> items;
        for (;;) {
            skip_extended(f);
            if (!(more() && peek() != '|' && peek() != ')')) break;
            const bool group = peek() == '(';
            int atom = parse_atom(f);
            if (atom < 0) continue;  // a flag setting (?i)
            atom = parse_quantifier(atom, group, f);
            items.push_back(atom);
        }
        if (items.empty()) return add(Node{});
        if (items.size() == 1) return items[0];
        Node n;
        n.kind = kCat;
        n.kids = std::move(items);
        return add(std::move(n));
    }

    bool parse_braces(int& mn, int& mx) {  // at '{': {m} {m,} {m,n} {,n}; anything else is a literal brace
        // As PCRE2 10.43 and later read it (the reference pins 10.46, src/CMakeLists.txt:185-189; Perl 5.34): `{,n}` is {0,n}, and
        // blanks may stand behind `{`, in front of `}` and on either side of the comma.  (The image's 10.39 takes both forms as
        // literal text: tests/test_regex_general.py judges them through the spelled-out {0,n}.)
        size_t q = pos_ + 1;
        auto blanks = [&]() {
            while (q < p_.size() && (p_[q] == ' ' || p_[q] == '\t')) ++q;
        };
        auto number = [&](int& v) {
            if (!(q < p_.size() && p_[q] >= '0' && p_[q] <= '9')) return false;
            long long x = 0;
            while (q < p_.size() && p_[q] >= '0' && p_[q] <= '9') {
                x = x * 10 + (p_[q] - '0');
                if (x > 65535) throw Invalid{"number too big in {} quantifier"};
                ++q;
            }
            v = int(x);
            return true;
        };
        blanks();
        const bool has_min = number(mn);
        if (!has_min) mn = 0;
        mx = mn;
        blanks();
        if (q < p_.size() && p_[q] == ',') {
            ++q;
            blanks();
            if (!number(mx)) {
                if (!has_min) return false;   // "{,}" is literal text
                mx = -1;
            }
            blanks();
        } else if (!has_min) {
            return false;
        }
        if (!(q < p_.size() && p_[q] == '}')) return false;
        if (mx >= 0 && mx < mn) throw Invalid{"numbers out of order in {} quantifier"};
        pos_ = q + 1;
        return true;
    }

    void skip_extended(const Flags& f) {   // (?x): white space and `#` comments between the items are not part of the pattern (inside a class they are)
        if (!f.extended) return;
        for (;;) {
            while (more() && (peek() == ' ' || (peek() >= 0x09 && peek() <= 0x0D))) ++pos_;
            if (peek() != '#') break;
            while (more() && peek() != '\n') ++pos_;
        }
    }

    int parse_quantifier(int atom, bool from_group, const Flags& f) {
        for (int count = 0;; ++count) {
            int mn, mx;
            skip_extended(f);
            if (eat('*')) { mn = 0; mx = -1; }
            else if (eat('+')) { mn = 1; mx = -1; }
            else if (eat('?')) { mn = 0; mx = 1; }
            else if (peek() == '{' && parse_braces(mn, mx)) {}
            else return atom;
            if (count) throw Invalid{"quantifier does not follow a repeatable item"};
            if (nodes_[size_t(atom)].kind == kLookNode || nodes_[size_t(atom)].kind == kAssert) {
                const Node& a = nodes_[size_t(atom)];
                if (!(from_group || a.kind == kLookNode || a.akind == kAhead || a.akind == kBehind))
                    throw Invalid{"quantifier does not follow a repeatable item"};   // (^* $* \\b*)
                // a quantified assertion group -- (?=a)*, (^){1,2} --: tried once where the minimum is not zero, and free to fail where it is:
                // nothing is consumed either way and there are no captures to tell the difference
                if (!eat('?')) eat('+');
                atom = mn == 0 ? add(Node{}) : atom;
                continue;
            }
            Node n;
            n.kind = kRepeat;
            n.kids = {atom};
            n.min = mn;
            n.max = mx;
            if (eat('?')) n.mode = 1;
            else if (eat('+')) n.mode = 2;
            if (n.max > 1000 || n.min > 1000) throw Unsupported{"repeat count too large"};
            const bool possessive_group = n.mode == 2 && nodes_[size_t(atom)].kind != kSet;
            if (possessive_group) n.mode = 0;   // X*+ is (?>X*)
            atom = add(std::move(n));
            if (possessive_group) atom = atomize(atom);
        }
    }

    static int hex(uint32_t c) {
        if (c >= '0' && c <= '9') return int(c - '0');
        if (c >= 'a' && c <= 'f') return int(c - 'a' + 10);
        if (c >= 'A' && c <= 'F') return int(c - 'A' + 10);
        return -1;
    }

    // \p / \P at pos_ (behind the letter): the set, by General_Category.
    CharSet parse_property(bool negate, bool caseless = false) {
        std::string name;
        if (eat('{')) {
            if (eat('^')) negate = !negate;
            while (more() && peek() != '}') name.push_back(char(p_[pos_++]));
            if (!eat('}')) throw Invalid{"malformed \\P or \\p sequence"};
        } else if (more()) {
            name.push_back(char(p_[pos_++]));
        }
        std::string key;
        for (char c : name)
            if (c != ' ' && c != '_' && c != '-') key.push_back(c);
        uint32_t mask = 0;
        CharSet s;
        bool direct = false;
        if (key == "L") mask = kMaskL;
        else if (key == "M") mask = kMaskM;
        else if (key == "N") mask = kMaskN;
        else if (key == "P") mask = kMaskP;
        else if (key == "S") mask = kMaskS;
        else if (key == "Z") mask = kMaskZ;
        else if (key == "C") mask = kMaskC;
        else if (key == "L&" || key == "Lc" || key == "LC") mask = gc_bit(Lu) | gc_bit(Ll) | gc_bit(Lt);
        else if (key == "Any" || key == "any") { s.add(0, kMaxCp); direct = true; }
        else if (key == "Xan") mask = kMaskL | kMaskN;
        else if (key == "Xsp" || key == "Xps") { s = space_set(); direct = true; }
        else if (key == "Xwd") { s = word_set(); direct = true; }
        else {
            for (int g = 0; g < kGcCount; ++g)
                if (key == kGcNames[g]) mask = gc_bit(g);
            if (!mask) {
                // a script: \p{Han} is Script_Extensions since PCRE2 10.40 (the reference pins 10.46), \p{sc:Han} / \p{script=Han} the
                // Script property, \p{scx:Han} / \p{scriptextensions=Han} the extensions by name
                std::string low;
                for (char c : key) low.push_back(char(std::tolower(static_cast<unsigned char>(c))));
                bool ext = true;
                const size_t sep = low.find_first_of(":=");
                if (sep != std::string::npos) {
                    const std::string prop = low.substr(0, sep);
                    if (prop == "sc" || prop == "script") ext = false;
                    else if (prop != "scx" && prop != "scriptextensions") throw Unsupported{"\\p{" + name + "}: unknown property"};
                    low = low.substr(sep + 1);
                }
                if (!script_set(low, ext, s)) throw Unsupported{"\\p{" + name + "}: neither a General_Category nor a script of Unicode 16.0"};
                direct = true;
            }
        }
        // \p{Lu} \p{Ll} \p{Lt} match any cased letter under PCRE2's caseless rules: not reproduced
        const uint32_t cased = gc_bit(Lu) | gc_bit(Ll) | gc_bit(Lt);
        if (caseless && !direct && (mask & cased) && (mask & cased) != cased) throw Unsupported{"(?i) with \\p{Lu} / \\p{Ll} / \\p{Lt}"};
        if (!direct) s = gc_set(mask);
        s.normalize();
        return negate ? s.negated() : s;
    }

    // An escape (pos_ behind the backslash).  Returns true and fills `set` for a character-type escape; returns false
    // and fills `cp` for a single character.  in_class: inside [...].
    bool parse_escape(bool in_class, CharSet& set, uint32_t& cp, bool caseless = false) {
        if (!more()) throw Invalid{"\\ at end of pattern"};
        const uint32_t c = p_[pos_++];
        switch (c) {
            case 'd': set = gc_set(gc_bit(Nd)); return true;
            case 'D': set = gc_set(gc_bit(Nd)).negated(); return true;
            case 's': set = space_set(); return true;
            case 'S': set = space_set().negated(); return true;
            case 'w': set = word_set(); return true;
            case 'W': set = word_set().negated(); return true;
            case 'h': set = hspace_set(); return true;
            case 'H': set = hspace_set().negated(); return true;
            case 'v': set = vspace_set(); return true;
            case 'V': set = vspace_set().negated(); return true;
            case 'p': set = parse_property(false, caseless); return true;
            case 'P': set = parse_property(true, caseless); return true;
            case 'N':
                if (in_class || peek() == '{') throw Unsupported{"\\N{...}"};
                set = CharSet{};
                set.add('\n', '\n');
                set = set.negated();
                return true;
            case 't': cp = '\t'; return false;
            case 'n': cp = '\n'; return false;
            case 'r': cp = '\r'; return false;
            case 'f': cp = '\f'; return false;
            case 'e': cp = 0x1B; return false;
            case 'a': cp = 0x07; return false;
            case '0': {
                uint32_t v = 0;
                for (int k = 0; k < 2 && peek() >= '0' && peek() <= '7'; ++k) v = v * 8 + (p_[pos_++] - '0');
                cp = v;
                return false;
            }
            case 'x': {
                uint32_t v = 0;
                if (eat('{')) {
                    int digits = 0;
                    while (more() && hex(peek()) >= 0) {
                        v = v * 16 + uint32_t(hex(p_[pos_++]));
                        if (++digits > 6) throw Unsupported{"\\x{...} too long"};
                    }
                    if (!eat('}') || digits == 0) throw Invalid{"malformed \\x{...}"};
                } else {
                    for (int k = 0; k < 2 && more() && hex(peek()) >= 0; ++k) v = v * 16 + uint32_t(hex(p_[pos_++]));
                }
                if (v > kMaxCp) throw Invalid{"code point value in \\x{} is too large"};
                cp = v;
                return false;
            }
            case 'b':
                if (in_class) { cp = 0x08; return false; }
                break;
            default: break;
        }
        if (c < 0x80 && ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')))
            throw Unsupported{std::string("escape \\") + char(c)};
        cp = c;  // an escaped punctuation / non-ASCII character stands for itself
        return false;
    }

    int parse_class(const Flags& f) {  // pos_ behind '['
        const bool negate = eat('^');
        CharSet explicit_chars, props;
        bool first = true, after_class_item = false;
        for (;;) {
            if (!more()) throw Invalid{"missing terminating ] for character class"};
            uint32_t c = p_[pos_];
            if (c == ']' && !first) { ++pos_; break; }
            first = false;
            // "[\d-x]": PCRE2 rejects a range that starts at a class escape / POSIX class (only "-]" is a literal hyphen there)
            if (c == '-' && after_class_item && peek(1) != ']') throw Invalid{"invalid range in character class"};
            after_class_item = false;
            if (c == '[' && peek(1) == ':') {  // POSIX class; under UCP: Unicode properties
                size_t q = pos_ + 2;
                bool neg = false;
                if (q < p_.size() && p_[q] == '^') { neg = true; ++q; }
                std::string name;
                while (q < p_.size() && p_[q] != ':') name.push_back(char(p_[q++]));
                if (!(q + 1 < p_.size() && p_[q + 1] == ']')) throw Unsupported{"malformed POSIX class"};
                pos_ = q + 2;
                CharSet s;
                if (name == "alpha") s = gc_set(kMaskL);
                else if (name == "lower") s = gc_set(gc_bit(Ll));
                else if (name == "upper") s = gc_set(gc_bit(Lu));
                else if (name == "alnum") s = gc_set(kMaskL | kMaskN);
                else if (name == "digit") s = gc_set(gc_bit(Nd));
                else if (name == "space") s = space_set();
                else if (name == "word") s = word_set();
                else if (name == "ascii") s.add(0, 0x7F);
                else if (name == "blank") s = hspace_set();
                else if (name == "cntrl") s = gc_set(gc_bit(Cc));
                else if (name == "xdigit") { s.add('0', '9'); s.add('a', 'f'); s.add('A', 'F'); }
                else if (name == "graph" || name == "print") {
                    // PCRE2 under UCP (pcre2pattern, "POSIX character classes"): what marks the page -- L, M, N, P, S, Cf without the
                    // invisible format characters U+061C, U+180E, U+2066-2069; [:print:] adds the Zs spaces
                    CharSet hidden;
                    hidden.add(0x61C, 0x61C);
                    hidden.add(0x180E, 0x180E);
                    hidden.add(0x2066, 0x2069);
                    hidden.normalize();
                    s = gc_set(kMaskL | kMaskN | kMaskP | kMaskS | gc_bit(Mn) | gc_bit(Mc) | gc_bit(Me) | gc_bit(Cf));
                    s = s.minus(hidden);
                    if (name == "print") s.add(gc_set(gc_bit(Zs)));
                } else if (name == "punct") {   // P, and the S characters below 128
                    s = gc_set(kMaskP);
                    CharSet sym = gc_set(kMaskS), low;
                    low.add(0, 0x7F);
                    s.add(sym.intersected(low));
                }
                else throw Invalid{"unknown POSIX class name"};
                if ((name == "lower" || name == "upper") && f.caseless) throw Unsupported{"(?i) with [:lower:] / [:upper:]"};
                s.normalize();
                props.add(neg ? s.negated() : s);
                after_class_item = true;
                continue;
            }
            if (c == '[' && (peek(1) == '.' || peek(1) == '=')) throw Invalid{"POSIX collating elements are not supported"};
            uint32_t lo;
            ++pos_;
            if (c == '\\') {
                if (looking_at("Q")) throw Unsupported{"\\Q inside a class"};
                CharSet s;
                if (parse_escape(true, s, lo, f.caseless)) {
                    props.add(s);
                    after_class_item = true;
                    continue;
                }
            } else {
                lo = c;
            }
            uint32_t hi = lo;
            if (peek() == '-' && peek(1) != ']' && pos_ + 1 < p_.size()) {
                ++pos_;
                uint32_t c2 = p_[pos_++];
                if (c2 == '\\') {
                    CharSet s;
                    if (parse_escape(true, s, hi, f.caseless)) throw Invalid{"invalid range in character class"};
                } else if (c2 == '[' && peek() == ':') {
                    throw Unsupported{"class range that ends at a POSIX class"};
                } else {
                    hi = c2;
                }
                if (hi < lo) throw Invalid{"range out of order in character class"};
            }
            explicit_chars.add(lo, hi);
        }
        if (f.caseless) close_caseless(explicit_chars);
        explicit_chars.add(props);
        explicit_chars.normalize();
        return set_node(negate ? explicit_chars.negated() : explicit_chars);
    }

    int assert_node(int kind, bool neg = false, CharSet set = CharSet{}) {
        Node n;
        n.kind = kAssert;
        n.akind = kind;
        n.aneg = neg;
        set.normalize();
        n.set = std::move(set);
        return add(std::move(n));
    }

    // The body of a look-around group as ONE character set (a class, a literal, an alternation of those).
    bool single_char_set(int node, CharSet& out) const {
        const Node& n = nodes_[size_t(node)];
        if (n.kind == kSet) { out.add(n.set); return true; }
        if (n.kind == kAlt) {
            for (int k : n.kids)
                if (!single_char_set(k, out)) return false;
            return true;
        }
        return false;
    }

    // `node` as a sequence of character sets (a literal string, classes, counted repeats of those): the only thing it can match is one
    // character of each, in order.
    bool fixed_sets(int node, std::vector<CharSet>& out) const {
        const Node& n = nodes_[size_t(node)];
        switch (n.kind) {
            case kEmpty: return true;
            case kSet: out.push_back(n.set); return true;
            case kCat:
                for (int k : n.kids)
                    if (!fixed_sets(k, out)) return false;
                return true;
            case kRepeat:
                if (n.min != n.max || n.min > 64) return false;
                for (int k = 0; k < n.min; ++k)
                    if (!fixed_sets(n.kids[0], out)) return false;
                return true;
            default: return false;
        }
    }
    // One way to match at most (no alternative to fall back on inside): sets, assertions, look-arounds, counted repeats and sequences of those.
    bool single_path(int node) const {
        const Node& n = nodes_[size_t(node)];
        switch (n.kind) {
            case kEmpty: case kSet: case kAssert: case kLookNode: case kAtomic: return true;
            case kCat:
                for (int k : n.kids)
                    if (!single_path(k)) return false;
                return true;
            case kRepeat: return n.min == n.max && single_path(n.kids[0]);
            default: return false;
        }
    }
    int look_node(bool neg, int body) {
        CharSet s;
        if (single_char_set(body, s)) return assert_node(kAhead, neg, s);
        Node n;
        n.kind = kLookNode;
        n.aneg = neg;
        n.kids = {body};
        return add(std::move(n));
    }
    // (?>X): X matched the FIRST way its alternatives and repeats allow (in PCRE2's order) and never re-entered -- rewritten into plain
    // syntax where that first way can be told by looking ahead:
    //   one way to match at most            X itself
    //   A1|A2|...                           (?>A1) | (?!A1)(?>A2) | (?!A1)(?!A2)(?>A3) ...  (A2 is tried only where A1 does not match at all)
    //   F{n,m} greedy, F a fixed sequence   F{n,m} whose every early way out is taken only where no further F follows (the possessive form)
    //   F{n,m}? lazy                        F{n}
    //   P X, P with one way to match        P (?>X)
    // Anything else (a repeat in front of something that could make it give characters back inside the group) stays a group of its own kind,
    // kAtomic, for the table builder: the threads inside it are told apart by the entry they came from, and the first of an entry's threads
    // to leave the group ends the ones behind it (compile_regex).
    int atomize(int node) {
        if (single_path(node)) return node;
        const Node n = nodes_[size_t(node)];   // (copy: nodes_ grows below)
        if (n.kind == kAlt) {
            Node alt;
            alt.kind = kAlt;
            for (size_t k = 0; k < n.kids.size(); ++k) {
                const int body = atomize(n.kids[k]);
                if (k == 0) { alt.kids.push_back(body); continue; }
                Node cat;
                cat.kind = kCat;
                for (size_t j = 0; j < k; ++j) cat.kids.push_back(look_node(true, n.kids[j]));
                cat.kids.push_back(body);
                alt.kids.push_back(add(std::move(cat)));
            }
            return add(std::move(alt));
        }
        if (n.kind == kRepeat) {
            std::vector<CharSet> seq;
            if (!fixed_sets(n.kids[0], seq) || seq.empty()) return atomic_node(node);
            Node r = n;
            if (n.mode == 1) r.max = r.min;
            else r.mode = 2;
            return add(std::move(r));
        }
        if (n.kind == kCat) {
            for (size_t k = 0; k + 1 < n.kids.size(); ++k)
                if (!single_path(n.kids[k])) return atomic_node(node);
            Node cat = n;
            cat.kids.back() = atomize(n.kids.back());
            return add(std::move(cat));
        }
        return atomic_node(node);
    }
    int atomic_node(int body) {
        Node n;
        n.kind = kAtomic;
        n.kids = {body};
        return add(std::move(n));
    }

    int parse_group(Flags& outer) {  // pos_ behind '('
        Flags f = outer;
        if (eat('?')) {
            if (eat(':') || eat('|')) {   // (?|...): the branches number their groups alike -- nothing here looks at group numbers
            } else if (peek() == '=' || peek() == '!') {
                const bool neg = p_[pos_++] == '!';
                const int body = parse_alt(f);
                if (!eat(')')) throw Invalid{"missing closing parenthesis"};
                return look_node(neg, body);   // one character: an assertion on the next class; anything longer: decided while the characters go by
            } else if (peek() == '<' && (peek(1) == '=' || peek(1) == '!')) {
                const bool neg = peek(1) == '!';
                pos_ += 2;
                const int body = parse_alt(f);
                if (!eat(')')) throw Invalid{"missing closing parenthesis"};
                CharSet s;
                if (single_char_set(body, s)) return assert_node(kBehind, neg, s);
                // alternatives of fixed sequences of sets, kRegexMaxBehind characters at most (PCRE2 wants every alternative of a
                // fixed length; anything else here is its error 125 or outside this subset)
                std::vector<std::vector<CharSet>> alts;
                const Node& b = nodes_[size_t(body)];
                for (int k : b.kind == kAlt ? b.kids : std::vector<int>{body}) {
                    std::vector<CharSet> seq;
                    if (!fixed_sets(k, seq) || seq.empty()) throw Unsupported{"look-behind on something other than fixed sequences of characters"};
                    if (seq.size() > size_t(kRegexMaxBehind)) throw Unsupported{"look-behind over more than 8 characters"};
                    alts.push_back(std::move(seq));
                }
                const int a = assert_node(kBehind, neg);
                nodes_[size_t(a)].behind = std::move(alts);
                return a;
            } else if (eat('>')) {   // atomic group
                const int body = parse_alt(f);
                if (!eat(')')) throw Invalid{"missing closing parenthesis"};
                return atomize(body);
            } else if (peek() == '<' || looking_at("P<") || peek() == '\'') {  // named capture: a plain group here
                const uint32_t close = peek() == '\'' ? '\'' : '>';
                if (looking_at("P<")) pos_ += 2;
                else ++pos_;
                std::string name;
                while (more() && peek() != close) {
                    const uint32_t ch = p_[pos_++];
                    if (ch >= 0x80) throw Unsupported{"group name outside ASCII"};
                    name.push_back(char(ch));
                }
                // PCRE2: a name is word characters, not starting with a digit, at most 32 of them, and not used twice (error 142 / 144 / 143)
                bool ok = eat(close) && !name.empty() && name.size() <= 32 && !(name[0] >= '0' && name[0] <= '9');
                for (char ch : name) ok = ok && ((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '_');
                if (!ok) throw Invalid{"syntax error in subpattern name"};
                for (const std::string& seen : group_names_)
                    if (seen == name) throw Invalid{"two named subpatterns have the same name"};
                group_names_.push_back(name);
            } else {
                // flag settings: (?i) (?s) (?is) (?-i) ... and the scoped form (?i: ... )
                bool on = true, any = false;
                Flags g = f;
                while (more() && peek() != ')' && peek() != ':') {
                    const uint32_t c = p_[pos_++];
                    if (c == '-') on = false;
                    else if (c == 'i') g.caseless = on;
                    else if (c == 's') g.dotall = on;
                    else if (c == 'm') g.multiline = on;
                    else if (c == 'x') g.extended = on;
                    // a letter PCRE2 has no option for is its error 111; anything else here -- `>` `|` `(` `R` digits `&` `#` `C` `+` ...: atomic
                    // groups, branch reset, conditions, recursion, comments, callouts -- is syntax it knows and this subset does not cover
                    else if (((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) && !std::strchr("imnsxJUarRPC", int(c))) throw Invalid{"unrecognized character after (? or (?-"};
                    else throw Unsupported{std::string("group option (?") + char(c < 0x80 ? c : '?') + ")"};
                    any = true;
                }
                if (!any) throw Unsupported{"unsupported group type"};
                if (eat(')')) {  // applies to the rest of the enclosing group, later alternatives included
                    outer = g;
                    return -1;
                }
                if (!eat(':')) throw Invalid{"missing closing parenthesis"};   // (the loop above stops at `)`, `:` or the pattern's end)
                f = g;
            }
        } else if (peek() == '*') {
            // (*VERB), (*VERB:NAME), (*alpha_assertion:...): what PCRE2 knows is outside the subset; anything else it rejects (error 160 / 195)
            static const char* const known[] = {"UTF", "UCP", "ANY", "ANYCRLF", "CR", "LF", "CRLF", "NUL", "BSR_ANYCRLF", "BSR_UNICODE", "NO_AUTO_POSSESS",
                                                "NO_DOTSTAR_ANCHOR", "NO_JIT", "NO_START_OPT", "NOTEMPTY", "NOTEMPTY_ATSTART", "LIMIT_HEAP", "LIMIT_MATCH",
                                                "LIMIT_DEPTH", "LIMIT_RECURSION", "ACCEPT", "COMMIT", "FAIL", "F", "PRUNE", "SKIP", "THEN", "MARK", "",
                                                "pla", "plb", "nla", "nlb", "napla", "naplb", "positive_lookahead", "positive_lookbehind", "negative_lookahead",
                                                "negative_lookbehind", "non_atomic_positive_lookahead", "non_atomic_positive_lookbehind", "atomic", "sr",
                                                "asr", "script_run", "atomic_script_run", "scan_substring", "scs"};
            std::string name;
            for (size_t q = pos_ + 1; q < p_.size() && ((p_[q] >= 'A' && p_[q] <= 'Z') || (p_[q] >= 'a' && p_[q] <= 'z') || p_[q] == '_'); ++q) name.push_back(char(p_[q]));
            for (const char* k : known)
                if (name == k) throw Unsupported{"(*VERB)"};
            throw Invalid{"(*VERB) not recognized"};
        }
        const int body = parse_alt(f);
        if (!eat(')')) throw Invalid{"missing closing parenthesis"};
        return body;
    }

    int literal(uint32_t cp, const Flags& f) {
        CharSet s;
        s.add(cp, cp);
        if (f.caseless) close_caseless(s);
        return set_node(s);
    }

    int parse_atom(Flags& f) {
        const uint32_t c = p_[pos_++];
        switch (c) {
            case '(': return parse_group(f);
            case '[': return parse_class(f);
            case '.': {
                CharSet s;
                if (f.dotall) s.add(0, kMaxCp);
                else { s.add('\n', '\n'); s = s.negated(); }
                return set_node(s);
            }
            // (?m): `^` also behind every \n, `$` also in front of every \n (and at the very end: PCRE2_MULTILINE drops "before a final \n"
            // as a case of its own -- it is one of "in front of a \n") -- each an alternation of two one-character assertions
            case '^': {
                if (!f.multiline) return assert_node(kBot);
                CharSet nl;
                nl.add('\n', '\n');
                Node n;
                n.kind = kAlt;
                n.kids = {assert_node(kBot), assert_node(kBehind, false, nl)};
                return add(std::move(n));
            }
            case '$': {
                if (!f.multiline) return assert_node(kEotNl);
                CharSet nl;
                nl.add('\n', '\n');
                Node n;
                n.kind = kAlt;
                n.kids = {assert_node(kAhead, false, nl), assert_node(kEot)};
                return add(std::move(n));
            }
            case '*': case '+': case '?': throw Invalid{"quantifier does not follow a repeatable item"};
            case '\\': {
                if (eat('Q')) {  // \Q ... \E: literal text
                    std::vector<int> items;
                    while (more() && !looking_at("\\E")) items.push_back(literal(p_[pos_++], f));
                    if (more()) pos_ += 2;
                    if (items.empty()) return -1;
                    if (items.size() == 1) return items[0];
                    // quantifiers apply to the last character only: return a concatenation whose last element the
                    // caller cannot separate -- refuse the rare quantified form
                    if (peek() == '*' || peek() == '+' || peek() == '?' || peek() == '{') throw Unsupported{"quantifier behind \\E"};
                    Node n;
                    n.kind = kCat;
                    n.kids = std::move(items);
                    return add(std::move(n));
                }
                if (eat('E')) return -1;
                switch (peek()) {
                    case 'A': ++pos_; return assert_node(kBot);
                    case 'z': ++pos_; return assert_node(kEot);
                    case 'Z': ++pos_; return assert_node(kEotNl);
                    case 'b': ++pos_; return assert_node(kWordB);
                    case 'B': ++pos_; return assert_node(kNotWordB);
                    case 'R': {   // (?>\r\n|\n|\x0b|\f|\r|\x85|\x{2028}|\x{2029}): any Unicode newline sequence, \r\n never split
                        ++pos_;
                        Node crlf;
                        crlf.kind = kCat;
                        CharSet cr, lf;
                        cr.add('\r', '\r');
                        lf.add('\n', '\n');
                        crlf.kids = {set_node(cr), set_node(lf)};
                        Node alt;
                        alt.kind = kAlt;
                        alt.kids = {add(std::move(crlf)), set_node(vspace_set())};
                        return atomize(add(std::move(alt)));
                    }
                    case 'G': case 'K': case 'X': case 'C': case 'g': case 'k':
                        throw Unsupported{std::string("escape \\") + char(peek())};
                    case 'L': case 'l': case 'U': case 'u': case 'F':
                        throw Invalid{"PCRE2 does not support \\F, \\L, \\l, \\U, or \\u"};
                    default: break;
                }
                if (peek() >= '1' && peek() <= '9') throw Unsupported{"back-reference"};
                CharSet s;
                uint32_t cp = 0;
                if (parse_escape(false, s, cp, f.caseless)) return set_node(s);
                return literal(cp, f);
            }
            default: return literal(c, f);
        }
    }
};

// ---------------------------------------------------------------------------------------------- backtracking automaton
enum Op : uint8_t { kChar, kSplit, kJmp, kAssertOp, kMatch, kLook, kLookAccept, kAtomEnter, kAtomExit };
struct Inst {
    Op op;
    int x = 0, y = 0;  // kChar: x = set id, y = next; kSplit: x preferred, y other; kJmp: x; kAssertOp: x = assert id, y = next;
                       // kLook: x = entry of the looked-for X (which ends in kLookAccept), y = next; kAtomEnter / kAtomExit: x = next
    // kSplit at the head of a greedy / possessive unbounded repeat (x = one more round of the body, y = the way out).  When
    // the body gets back here without having consumed a character, PCRE2 ends the repeat THERE -- what follows the repeat is
    // tried before the body's remaining alternatives ((a??)+ on "a" matches the empty string, not "a").
    bool loop = false;
    bool neg = false;   // kLook
};
struct AssertInfo {
    int kind;
    bool neg;
    int set;  // index into sets, or -1
    std::vector<std::vector<int>> behind;   // kBehind over sequences: per alternative the set ids, first character first
};

struct Builder {
    const std::vector<Node>& nodes;
    std::vector<Inst> prog;
    std::vector<CharSet> sets;
    std::vector<AssertInfo> asserts;

    int set_id(const CharSet& s) {
        for (size_t i = 0; i < sets.size(); ++i)
            if (sets[i].r == s.r) return int(i);
        sets.push_back(s);
        return int(sets.size()) - 1;
    }
    int emit(Inst i) {
        if (prog.size() > 20000) throw Unsupported{"pattern too large"};
        prog.push_back(i);
        return int(prog.size()) - 1;
    }
    int emit_assert(int kind, bool neg, const CharSet* s, int next) {
        asserts.push_back(AssertInfo{kind, neg, s ? set_id(*s) : -1, {}});
        return emit(Inst{kAssertOp, int(asserts.size()) - 1, next});
    }
    int emit_look(bool neg, int body, int next) {
        const int accept = emit(Inst{kLookAccept});
        const int entry = gen(body, accept);
        Inst in{kLook, entry, next};
        in.neg = neg;
        return emit(in);
    }
    // Code that matches node `n` and continues at `next`; returns its entry.
    int gen(int n, int next) {
        const Node& nd = nodes[size_t(n)];
        switch (nd.kind) {
            case kEmpty: return next;
            case kSet: return emit(Inst{kChar, set_id(nd.set), next});
            case kCat: {
                int e = next;
                for (size_t k = nd.kids.size(); k-- > 0;) e = gen(nd.kids[k], e);
                return e;
            }
            case kAlt: {
                std::vector<int> entries;
                for (int k : nd.kids) entries.push_back(gen(k, next));
                int e = entries.back();
                for (size_t k = entries.size() - 1; k-- > 0;) e = emit(Inst{kSplit, entries[k], e});
                return e;
            }
            case kAssert: {
                if (nd.akind == kBehind && !nd.behind.empty()) {
                    AssertInfo a{kBehind, nd.aneg, -1, {}};
                    for (const auto& seq : nd.behind) {
                        a.behind.emplace_back();
                        for (const CharSet& cs : seq) a.behind.back().push_back(set_id(cs));
                    }
                    asserts.push_back(std::move(a));
                    return emit(Inst{kAssertOp, int(asserts.size()) - 1, next});
                }
                return emit_assert(nd.akind, nd.aneg, (nd.akind == kAhead || nd.akind == kBehind) ? &nd.set : nullptr, next);
            }
            case kLookNode: return emit_look(nd.aneg, nd.kids[0], next);
            case kAtomic: {
                const int leave = emit(Inst{kAtomExit, next, 0});
                const int body = gen(nd.kids[0], leave);
                return emit(Inst{kAtomEnter, body, 0});
            }
            case kRepeat: {
                const int child = nd.kids[0];
                // where the "no further repetition" edge goes: straight on, or (possessive X: never give a character
                // back) through "the next character is not an X"
                auto skip_target = [&](int to) {
                    if (nd.mode != 2) return to;
                    if (nodes[size_t(child)].kind == kSet) return emit_assert(kAhead, true, &nodes[size_t(child)].set, to);
                    return emit_look(true, child, to);   // (a fixed sequence of sets: Parser::atomize)
                };
                auto split = [&](int body, int skip) {
                    return nd.mode == 1 ? emit(Inst{kSplit, skip, body}) : emit(Inst{kSplit, body, skip});
                };
                int e = next;
                if (nd.max < 0) {
                    const int loop = emit(Inst{kSplit, 0, 0});
                    const int body = gen(child, loop);
                    const int skip = skip_target(next);
                    prog[size_t(loop)] = nd.mode == 1 ? Inst{kSplit, skip, body} : Inst{kSplit, body, skip, true};
                    e = loop;
                } else {
                    for (int k = nd.min; k < nd.max; ++k) {
                        const int body = gen(child, e);
                        e = split(body, skip_target(next));
                    }
                }
                for (int k = 0; k < nd.min; ++k) e = gen(child, e);
                return e;
            }
        }
        return next;
    }
};

}  // namespace

int compile_regex(const std::string& pattern, RegexProgram& out, std::string& err) {
    try {
        std::vector<Node> nodes;
        Parser parser(pattern, nodes);
        const int root = parser.parse();
        Builder b{nodes, {}, {}, {}};
        const int match_pc = b.emit(Inst{kMatch});
        const int entry = b.gen(root, match_pc);

        // ---- what the assertions look at behind a position: sequences of sets (one set long for `\b`, one-character look-behind and
        // (?m)^; up to kRegexMaxBehind for (?<=abc)), first character first
        bool uses_bot = false, uses_word = false, uses_final_nl = false;
        std::vector<std::vector<int>> seqs;
        auto seq_id = [&](const std::vector<int>& q) {
            for (size_t i = 0; i < seqs.size(); ++i)
                if (seqs[i] == q) return int(i);
            seqs.push_back(q);
            return int(seqs.size()) - 1;
        };
        int word_set_id = -1, word_seq = -1;
        for (const AssertInfo& a : b.asserts)
            if (a.kind == kWordB || a.kind == kNotWordB) uses_word = true;
        if (uses_word) {
            word_set_id = b.set_id(word_set());
            word_seq = seq_id({word_set_id});
        }
        std::vector<std::vector<int>> seqs_of_assert(b.asserts.size());
        for (size_t i = 0; i < b.asserts.size(); ++i) {
            const AssertInfo& a = b.asserts[i];
            if (a.kind == kBot) uses_bot = true;
            if (a.kind == kEotNl) uses_final_nl = true;
            if (a.kind != kBehind) continue;
            if (a.behind.empty()) seqs_of_assert[i].push_back(seq_id({a.set}));
            for (const auto& q : a.behind) seqs_of_assert[i].push_back(seq_id(q));
        }
        int behind_chars = 0;
        for (const auto& q : seqs) behind_chars = std::max(behind_chars, int(q.size()));

        // ---- partition of the code points: signature = membership in every set
        const size_t n_sets = b.sets.size();
        std::vector<uint32_t> cuts{0, 0x80, kMaxCp + 1};  // (ASCII / non-ASCII is not a class border by itself; harmless)
        for (const CharSet& s : b.sets)
            for (const auto& r : s.r) {
                cuts.push_back(r.first);
                cuts.push_back(r.second + 1);
            }
        std::sort(cuts.begin(), cuts.end());
        cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
        std::map<std::vector<bool>, int> class_of_sig;
        std::vector<std::vector<bool>> sig_of_class;
        std::vector<uint8_t> cls(size_t(kMaxCp) + 1, 0);
        for (size_t i = 0; i + 1 < cuts.size(); ++i) {
            const uint32_t lo = cuts[i], hi = cuts[i + 1] - 1;
            std::vector<bool> sig(n_sets);
            for (size_t s = 0; s < n_sets; ++s) sig[s] = b.sets[s].has(lo);
            auto it = class_of_sig.find(sig);
            if (it == class_of_sig.end()) {
                if (sig_of_class.size() >= size_t(kRegexMaxClasses)) throw Unsupported{"too many character classes"};
                it = class_of_sig.emplace(sig, int(sig_of_class.size())).first;
                sig_of_class.push_back(sig);
            }
            std::fill(cls.begin() + lo, cls.begin() + hi + 1, uint8_t(it->second));
        }
        const int n_classes = int(sig_of_class.size());
        const int sym_eot = n_classes, sym_final_nl = uses_final_nl ? n_classes + 1 : -1;
        const int n_syms = n_classes + 1 + (uses_final_nl ? 1 : 0);
        const int nl_class = cls['\n'];
        auto sym_class = [&](int sym) { return sym == sym_final_nl ? nl_class : (sym < n_classes ? sym : -1); };

        // ---- contexts: a small automaton over the classes that knows, for every sequence, which of its prefixes the text behind the
        // position ends in (bit l-1 of its mask: the last l characters are the sequence's first l).  Context 0 = start of the subject,
        // context 1 = nothing known to match (also where the kernels start when they re-read the kRegexMaxBehind characters behind a
        // position in the middle of a string: a prefix of l characters depends on the last l characters only).
        const bool track_ctx = uses_bot || !seqs.empty();
        std::vector<std::vector<uint16_t>> ctx_masks{std::vector<uint16_t>(seqs.size(), 0)};
        std::vector<uint8_t> ctx_next;
        int n_ctx = 1;
        if (!track_ctx) ctx_next.assign(size_t(n_classes), 0);
        if (track_ctx) {
            std::map<std::vector<uint16_t>, int> ids;
            ctx_masks.push_back(ctx_masks[0]);
            ids.emplace(ctx_masks[1], 1);
            n_ctx = 2;
            for (int x = 0; x < n_ctx; ++x) {
                ctx_next.resize(size_t(x + 1) * size_t(n_classes));
                for (int c = 0; c < n_classes; ++c) {
                    std::vector<uint16_t> m(seqs.size());
                    for (size_t j = 0; j < seqs.size(); ++j) {
                        uint16_t fits = 0;
                        for (size_t l = 0; l < seqs[j].size(); ++l)
                            if (sig_of_class[size_t(c)][size_t(seqs[j][l])]) fits |= uint16_t(1u << l);
                        m[j] = uint16_t(((ctx_masks[size_t(x)][j] << 1) | 1u) & fits);
                    }
                    auto it = ids.find(m);
                    if (it == ids.end()) {
                        if (n_ctx >= kRegexMaxCtx) throw Unsupported{"too many look-behind contexts"};
                        it = ids.emplace(m, n_ctx++).first;
                        ctx_masks.push_back(m);
                    }
                    ctx_next[size_t(x) * size_t(n_classes) + size_t(c)] = uint8_t(it->second);
                }
            }
        }
        auto ctx_ends_in = [&](int ctx, int seq) { return ((ctx_masks[size_t(ctx)][size_t(seq)] >> (seqs[size_t(seq)].size() - 1)) & 1u) != 0; };
        auto assert_ok = [&](int id, int ctx, int sym, int c) {
            const AssertInfo& a = b.asserts[size_t(id)];
            switch (a.kind) {
                case kBot: return ctx == 0;
                case kEot: return sym == sym_eot;
                case kEotNl: return sym == sym_eot || sym == sym_final_nl;
                case kAhead: return (c >= 0 && sig_of_class[size_t(c)][size_t(a.set)]) != a.neg;
                case kBehind: {
                    bool any = false;
                    for (int q : seqs_of_assert[size_t(id)]) any = any || ctx_ends_in(ctx, q);
                    return any != a.neg;
                }
                case kWordB:
                case kNotWordB: {
                    const bool wp = ctx_ends_in(ctx, word_seq);
                    const bool wn = c >= 0 && sig_of_class[size_t(c)][size_t(word_set_id)];
                    return (wp != wn) == (a.kind == kWordB);
                }
            }
            return false;
        };

        // ---- subset construction over ordered lists of threads.  A thread is a position of the backtracking automaton; behind a
        // look-ahead over more than one character it carries that look-ahead as a CONDITION -- the set of positions of the looked-for
        // X that the characters read since have led to -- which travels with it, character by character, until X has matched (positive:
        // the condition is dropped; negative: the thread dies) or cannot any more (the other way round).  A thread that reaches the
        // pattern's end while it still carries conditions is a TENTATIVE match: it stays in the list, ages by one per character, cuts
        // nothing yet; when its last condition is dropped the transition reports the match with that age as the delay (the match ended
        // `age` characters back) and everything of lower priority is cut, as for an immediate match.
        //   An atomic group the parser could not rewrite (kAtomEnter .. kAtomExit) gives every thread that enters it a TAG of its own,
        // inherited by the threads that come of it.  When one of them leaves the group, the threads with the same tag BEHIND it in the
        // list (the ways PCRE2 would only try by backtracking into the group) end; the ones in FRONT of it might still leave the group
        // later, and then this one must not have been: it carries them as a negative condition -- "none of these positions reaches the
        // group's exit".
        struct Cond {
            bool neg;
            int accept;             // -1: the looked-for X ends in its own kLookAccept; else the kAtomExit that counts
            std::vector<int> pcs;   // sorted
            bool operator<(const Cond& o) const { return neg != o.neg ? neg < o.neg : (accept != o.accept ? accept < o.accept : pcs < o.pcs); }
            bool operator==(const Cond& o) const { return neg == o.neg && accept == o.accept && pcs == o.pcs; }
        };
        using Conds = std::vector<Cond>;   // sorted, no duplicates
        struct Item {
            int pc;    // >= 0: a thread about to run from this position; -1: a tentative match
            int age;   // tentative match: characters read since its end
            Conds conds;
            std::vector<int> tags;   // the atomic groups the thread is inside, outermost first (numbers of no meaning beyond "same entry")
            bool operator==(const Item& o) const { return pc == o.pc && age == o.age && conds == o.conds && tags == o.tags; }
        };
        // a state's key: the context and the items flattened (a plain thread is its position alone, as before look-aheads had conditions)
        using Key = std::pair<int, std::vector<int>>;
        auto flatten = [](const std::vector<Item>& items) {
            std::vector<int> v;
            std::vector<int> seen_tags;   // tags renumbered in the order they appear: equal states get equal keys
            for (const Item& it : items) {
                if (it.pc >= 0 && it.conds.empty() && it.tags.empty()) { v.push_back(it.pc); continue; }
                v.push_back(-1);
                v.push_back(it.pc);
                v.push_back(it.age);
                v.push_back(int(it.conds.size()));
                for (const Cond& cd : it.conds) {
                    v.push_back(cd.neg ? 1 : 0);
                    v.push_back(cd.accept);
                    v.push_back(int(cd.pcs.size()));
                    v.insert(v.end(), cd.pcs.begin(), cd.pcs.end());
                }
                v.push_back(int(it.tags.size()));
                for (int t : it.tags) {
                    size_t k = 0;
                    while (k < seen_tags.size() && seen_tags[k] != t) ++k;
                    if (k == seen_tags.size()) seen_tags.push_back(t);
                    v.push_back(int(k));
                }
            }
            return v;
        };
        auto unflatten = [](const std::vector<int>& v) {
            std::vector<Item> items;
            for (size_t i = 0; i < v.size();) {
                if (v[i] >= 0) { items.push_back(Item{v[i++], 0, {}, {}}); continue; }
                Item it{v[i + 1], v[i + 2], {}, {}};
                const int n = v[i + 3];
                i += 4;
                for (int k = 0; k < n; ++k) {
                    Cond cd{v[i] != 0, v[i + 1], {}};
                    const int m = v[i + 2];
                    cd.pcs.assign(v.begin() + long(i) + 3, v.begin() + long(i) + 3 + m);
                    i += 3 + size_t(m);
                    it.conds.push_back(std::move(cd));
                }
                const int nt = v[i++];
                it.tags.assign(v.begin() + long(i), v.begin() + long(i) + nt);
                i += size_t(nt);
                items.push_back(std::move(it));
            }
            return items;
        };
        std::map<Key, int> ids;
        std::vector<Key> states;
        states.push_back(Key{0, {}});  // state 0: dead
        ids.emplace(states[0], 0);
        auto intern = [&](Key k) {
            if (k.second.empty()) return 0;
            if (!track_ctx) k.first = 0;
            auto it = ids.find(k);
            if (it != ids.end()) return it->second;
            if (states.size() >= size_t(kRegexMaxStates)) throw Unsupported{"pattern needs more than 4096 DFA states"};
            const int id = int(states.size());
            ids.emplace(k, id);
            states.push_back(std::move(k));
            return id;
        };
        for (int c = 0; c < n_ctx; ++c) out.start[c] = uint16_t(intern(Key{c, {entry}}));
        std::vector<uint16_t> trans;
        std::vector<uint8_t> open(b.prog.size());
        std::set<std::tuple<int, uint64_t, int, int>> visited;
        long long closure_steps = 0;
        constexpr long long kMaxClosureSteps = 50'000'000;
        auto count_step = [&]() {
            if (++closure_steps > kMaxClosureSteps) throw Unsupported{"pattern too intricate for the table builder"};
        };

        // The looked-for X from the raw positions `raw`, at a position whose context and next symbol are known: has it matched
        // (`accept`), and which character positions wait for the next character (`chars`, sorted).
        auto look_close = [&](const std::vector<int>& raw, int accept_pc, int ctx, int sym, int c, bool& accept, std::vector<int>& chars) {
            accept = false;
            chars.clear();
            std::vector<int> st(raw.rbegin(), raw.rend());
            std::set<int> seen;
            while (!st.empty()) {
                const int pc = st.back();
                st.pop_back();
                if (!seen.insert(pc).second) continue;
                count_step();
                const Inst& in = b.prog[size_t(pc)];
                switch (in.op) {
                    case kJmp: st.push_back(in.x); break;
                    case kSplit: st.push_back(in.y); st.push_back(in.x); break;
                    case kChar: chars.push_back(pc); break;
                    case kLookAccept: accept = true; break;
                    case kAssertOp: if (assert_ok(in.x, ctx, sym, c)) st.push_back(in.y); break;
                    case kLook: throw Unsupported{"look-around inside a look-ahead of more than one character, or inside an atomic group that can give characters back"};
                    case kAtomExit:
                        if (pc == accept_pc) { accept = true; break; }
                        [[fallthrough]];
                    case kAtomEnter: throw Unsupported{"atomic group that can give characters back, inside a look-ahead or another such group"};
                    case kMatch: break;
                }
            }
            std::sort(chars.begin(), chars.end());
        };
        // ... from raw positions, before the next symbol is known: reached the end through jumps alone?
        auto look_done = [&](const std::vector<int>& raw, int accept_pc) {
            std::vector<int> st(raw);
            std::set<int> seen;
            while (!st.empty()) {
                const int pc = st.back();
                st.pop_back();
                if (!seen.insert(pc).second) continue;
                const Inst& in = b.prog[size_t(pc)];
                if (in.op == kLookAccept || pc == accept_pc) return true;
                if (in.op == kJmp) st.push_back(in.x);
                if (in.op == kSplit) { st.push_back(in.y); st.push_back(in.x); }
            }
            return false;
        };
        // conditions at a position (context, next symbol): false = the thread that carries them dies
        auto close_conds = [&](const Conds& in, int ctx, int sym, int c, Conds& res) {
            res.clear();
            bool accept = false;
            std::vector<int> chars;
            for (const Cond& cd : in) {
                look_close(cd.pcs, cd.accept, ctx, sym, c, accept, chars);
                if (accept) {
                    if (cd.neg) return false;
                } else if (chars.empty() || c < 0) {
                    if (!cd.neg) return false;
                } else {
                    res.push_back(Cond{cd.neg, cd.accept, chars});
                }
            }
            std::sort(res.begin(), res.end());
            res.erase(std::unique(res.begin(), res.end()), res.end());
            return true;
        };
        // ... over the character of class c
        auto step_conds = [&](const Conds& in, int c, Conds& res) {
            res.clear();
            for (const Cond& cd : in) {
                std::vector<int> raw;
                for (int pc : cd.pcs) {
                    const Inst& q = b.prog[size_t(pc)];
                    if (sig_of_class[size_t(c)][size_t(q.x)]) raw.push_back(q.y);
                }
                std::sort(raw.begin(), raw.end());
                raw.erase(std::unique(raw.begin(), raw.end()), raw.end());
                if (raw.empty()) {
                    if (!cd.neg) return false;
                } else if (look_done(raw, cd.accept)) {
                    if (cd.neg) return false;
                } else {
                    res.push_back(Cond{cd.neg, cd.accept, std::move(raw)});
                }
            }
            std::sort(res.begin(), res.end());
            res.erase(std::unique(res.begin(), res.end()), res.end());
            return true;
        };

        struct Entry { int pc, cid, age, tid; };   // pc == INT_MIN: a tentative match of the state; pc < 0 otherwise: "the body of repeat -pc - 1 has been walked"
        struct Found { int pc, cid, age, tid; };   // pc >= 0: a thread waiting at a character position; -1: a tentative match
        for (size_t s = 0; s < states.size(); ++s) {
            trans.resize((s + 1) * size_t(n_syms), 0);
            if (s == 0) continue;
            const int st_ctx = states[s].first;
            const std::vector<Item> st_items = unflatten(states[s].second);
            for (int sym = 0; sym < n_syms; ++sym) {
                const int c = sym_class(sym);
                std::vector<Conds> cl{Conds{}};   // the condition lists at hand in this transition; 0 = none
                auto cond_id = [&](const Conds& x) {
                    for (size_t i = 0; i < cl.size(); ++i)
                        if (cl[i] == x) return int(i);
                    cl.push_back(x);
                    return int(cl.size()) - 1;
                };
                std::vector<std::vector<int>> tl{std::vector<int>{}};   // ... and the tag lists; 0 = outside every atomic group
                auto tags_id = [&](const std::vector<int>& x) {
                    for (size_t i = 0; i < tl.size(); ++i)
                        if (tl[i] == x) return int(i);
                    tl.push_back(x);
                    return int(tl.size()) - 1;
                };
                int fresh_tag = 0;
                for (const Item& it : st_items)
                    for (int t : it.tags) fresh_tag = std::max(fresh_tag, t + 1);
                std::vector<int> ended_tags;   // entries of atomic groups one of whose threads has left the group in this transition
                auto ended = [&](int tid) {
                    for (int t : tl[size_t(tid)])
                        if (std::find(ended_tags.begin(), ended_tags.end(), t) != ended_tags.end()) return true;
                    return false;
                };
                bool matched = false;
                int delay = 0;
                std::vector<Found> found;
                auto have = [&](int pc, int cid, int age, int tid) {
                    for (const Found& f : found)
                        if (f.pc == pc && f.age == age && f.tid == tid && (f.cid == cid || (pc >= 0 && f.cid == 0))) return true;   // (the same position without conditions, earlier: all this one could do)
                    return false;
                };
                // the closure at this position: context st_ctx behind it, symbol sym next
                std::vector<Entry> stack;
                {
                    std::vector<Entry> first;
                    Conds res;
                    for (const Item& it : st_items) {
                        if (!close_conds(it.conds, st_ctx, sym, c, res)) continue;
                        first.push_back(Entry{it.pc >= 0 ? it.pc : INT_MIN, cond_id(res), it.age, tags_id(it.tags)});
                    }
                    stack.assign(first.rbegin(), first.rend());
                }
                std::fill(open.begin(), open.end(), 0);
                visited.clear();
                uint64_t open_sig = 0;  // which repeats are open on the way to the node at hand (order-independent sum)
                auto sig_of = [](int loop_pc) { return (uint64_t(loop_pc) + 1) * 0x9E3779B97F4A7C15ull; };
                while (!stack.empty() && !matched) {
                    const Entry e = stack.back();
                    stack.pop_back();
                    const int pc = e.pc;
                    if (pc == INT_MIN) {  // a tentative match from the characters before
                        if (cl[size_t(e.cid)].empty()) {
                            matched = true;
                            delay = e.age;
                        } else if (!have(-1, e.cid, e.age, 0)) {
                            found.push_back(Found{-1, e.cid, e.age, 0});
                        }
                        continue;
                    }
                    if (pc < 0) {  // the body of repeat -pc - 1 has been walked
                        open[size_t(-pc - 1)] = 0;
                        open_sig -= sig_of(-pc - 1);
                        continue;
                    }
                    if (ended(e.tid)) continue;   // a way into an atomic group that a thread in front of this one has left
                    if (open[size_t(pc)]) {  // an empty round of an open repeat: PCRE2 leaves the repeat here
                        stack.push_back(Entry{b.prog[size_t(pc)].y, e.cid, 0, e.tid});
                        continue;
                    }
                    // A node reached again is the same thread only if the same repeats are open (what follows an empty round
                    // depends on them) and it carries the same conditions.  (Character threads are told apart by `found` below: the first one wins.)
                    if (!visited.insert(std::make_tuple(pc, open_sig, e.cid, e.tid)).second) continue;
                    count_step();
                    const Inst& in = b.prog[size_t(pc)];
                    switch (in.op) {
                        case kJmp: stack.push_back(Entry{in.x, e.cid, 0, e.tid}); break;
                        case kSplit:
                            stack.push_back(Entry{in.y, e.cid, 0, e.tid});
                            if (in.loop) {
                                stack.push_back(Entry{-pc - 1, 0, 0, 0});
                                open[size_t(pc)] = 1;
                                open_sig += sig_of(pc);
                            }
                            stack.push_back(Entry{in.x, e.cid, 0, e.tid});
                            break;
                        case kChar:
                            if (!have(pc, e.cid, 0, e.tid)) found.push_back(Found{pc, e.cid, 0, e.tid});
                            break;
                        case kMatch:
                            if (cl[size_t(e.cid)].empty()) matched = true;   // everything of lower priority is cut
                            else if (!have(-1, e.cid, 0, 0)) found.push_back(Found{-1, e.cid, 0, 0});
                            break;
                        case kAssertOp:
                            if (assert_ok(in.x, st_ctx, sym, c)) stack.push_back(Entry{in.y, e.cid, 0, e.tid});
                            break;
                        case kAtomEnter: {
                            std::vector<int> tags = tl[size_t(e.tid)];
                            tags.push_back(fresh_tag++);
                            stack.push_back(Entry{in.x, e.cid, 0, tags_id(tags)});
                            break;
                        }
                        case kAtomExit: {
                            std::vector<int> tags = tl[size_t(e.tid)];
                            const int tag = tags.back();
                            tags.pop_back();
                            // the threads of the same entry in front of this one, still inside the group: this exit holds only if none of them gets here
                            std::vector<int> ahead;
                            for (const Found& f : found) {
                                const std::vector<int>& ft = tl[size_t(f.tid)];
                                if (std::find(ft.begin(), ft.end(), tag) == ft.end()) continue;
                                if (ft.back() != tag || f.cid != e.cid)
                                    throw Unsupported{"atomic group that can give characters back, with a look-ahead or another such group inside it"};
                                if (c >= 0 && sig_of_class[size_t(c)][size_t(b.prog[size_t(f.pc)].x)]) ahead.push_back(f.pc);   // (the others end at this character)
                            }
                            ended_tags.push_back(tag);   // ... and the ones behind it end here
                            int cid = e.cid;
                            if (!ahead.empty() && !tags.empty()) throw Unsupported{"atomic group that can give characters back, with a look-ahead or another such group inside it"};
                            if (!ahead.empty()) {
                                std::sort(ahead.begin(), ahead.end());
                                ahead.erase(std::unique(ahead.begin(), ahead.end()), ahead.end());
                                Conds with = cl[size_t(e.cid)];
                                with.push_back(Cond{true, pc, ahead});
                                std::sort(with.begin(), with.end());
                                with.erase(std::unique(with.begin(), with.end()), with.end());
                                cid = cond_id(with);
                            }
                            stack.push_back(Entry{in.x, cid, 0, tags_id(tags)});
                            break;
                        }
                        case kLook: {
                            bool accept = false;
                            std::vector<int> chars;
                            look_close({in.x}, -1, st_ctx, sym, c, accept, chars);
                            if (accept) {
                                if (!in.neg) stack.push_back(Entry{in.y, e.cid, 0, e.tid});
                            } else if (chars.empty() || c < 0) {
                                if (in.neg) stack.push_back(Entry{in.y, e.cid, 0, e.tid});
                            } else {
                                // (inside such an atomic group the thread would leave the group -- and end the ways behind it -- before the
                                // look-ahead is decided)
                                if (!tl[size_t(e.tid)].empty()) throw Unsupported{"atomic group that can give characters back, with a look-ahead or another such group inside it"};
                                Conds with = cl[size_t(e.cid)];
                                with.push_back(Cond{in.neg, -1, chars});
                                std::sort(with.begin(), with.end());
                                with.erase(std::unique(with.begin(), with.end()), with.end());
                                stack.push_back(Entry{in.y, cond_id(with), 0, e.tid});
                            }
                            break;
                        }
                        case kLookAccept: break;
                    }
                }
                // ... and over the character: the threads whose set holds it move on, conditions and tentative matches with them
                std::vector<Item> next;
                if (c >= 0) {
                    Conds res;
                    for (const Found& f : found) {
                        if (f.pc >= 0 && !sig_of_class[size_t(c)][size_t(b.prog[size_t(f.pc)].x)]) continue;
                        if (!step_conds(cl[size_t(f.cid)], c, res)) continue;
                        Item it{f.pc >= 0 ? b.prog[size_t(f.pc)].y : -1, f.pc >= 0 ? 0 : f.age + 1, res, tl[size_t(f.tid)]};
                        if (f.pc < 0 && res.empty()) {   // decided by this very character: a match that ended f.age characters back
                            matched = true;               // (of higher priority than one the closure found: that one came later in the list)
                            delay = f.age;
                            break;
                        }
                        if (it.age > kRegexMaxDelay) throw Unsupported{"a look-ahead that stays undecided for more than 7 characters behind the end of a match"};
                        bool dup = false;
                        for (const Item& o : next) dup = dup || o == it || (it.pc >= 0 && o.pc == it.pc && o.conds.empty() && o.tags == it.tags);
                        if (!dup) next.push_back(std::move(it));
                    }
                }
                const int next_ctx = c >= 0 && track_ctx ? int(ctx_next[size_t(st_ctx) * size_t(n_classes) + size_t(c)]) : 0;
                trans[s * size_t(n_syms) + size_t(sym)] =
                    uint16_t(intern(Key{next_ctx, flatten(next)})) | (matched ? uint16_t(kRegexMatchBit | (delay << kRegexDelayShift)) : uint16_t(0));
            }
        }
        out.trans = std::move(trans);
        out.n_states = int(states.size());
        out.n_syms = n_syms;
        out.n_classes = n_classes;
        out.sym_eot = sym_eot;
        out.sym_final_nl = sym_final_nl;
        out.n_ctx = n_ctx;
        out.ctx_next = std::move(ctx_next);
        out.behind_chars = behind_chars;
        out.can_match_empty = false;
        for (int c = 0; c < n_ctx; ++c)
            for (int sym = 0; sym < n_syms; ++sym)
                if (out.trans[size_t(out.start[c]) * size_t(n_syms) + size_t(sym)] & kRegexMatchBit) out.can_match_empty = true;
        std::memcpy(out.ascii_class, cls.data(), 128);
        // two-level table over blocks of 128 code points
        const size_t n_blocks_all = (size_t(kMaxCp) + 1) >> 7;
        out.cp_index.assign(n_blocks_all, 0);
        out.cp_blocks.clear();
        std::map<std::vector<uint8_t>, int> block_ids;
        for (size_t blk = 0; blk < n_blocks_all; ++blk) {
            std::vector<uint8_t> v(cls.begin() + long(blk * 128), cls.begin() + long(blk * 128 + 128));
            auto it = block_ids.find(v);
            if (it == block_ids.end()) {
                it = block_ids.emplace(v, int(block_ids.size())).first;
                out.cp_blocks.insert(out.cp_blocks.end(), v.begin(), v.end());
            }
            out.cp_index[blk] = uint16_t(it->second);
        }
        return OVTK_OK;
    } catch (const Unsupported& u) {
        err = "RegexSplit: pattern outside the subset compiled for the GPU (" + u.what + "); PCRE2 is not executed on the device";
        return OVTK_E_UNSUPPORTED;
    } catch (const Invalid& v) {
        // The reference keeps a null pattern (src/utils.cpp:264-271): nothing ever matches.  One dead state, one class.
        out = RegexProgram{};
        out.invalid = true;
        out.invalid_why = v.what;
        out.n_classes = 1;
        out.sym_eot = 1;
        out.n_syms = 2;
        out.n_states = 1;
        out.trans.assign(2, 0);
        out.ctx_next.assign(1, 0);
        out.cp_index.assign((size_t(kMaxCp) + 1) >> 7, 0);
        out.cp_blocks.assign(128, 0);
        return OVTK_OK;
    }
}

void unicode_general_categories(std::vector<uint8_t>& gc) {
    gc.assign(0x110000, 0);
    for (unsigned i = 0; i < kGcRanges; ++i)
        for (unsigned cp = kGcStart[i]; cp < kGcStart[i + 1] && cp < 0x110000u; ++cp) gc[cp] = uint8_t(kGcValue[i]);
}

}  // namespace ovtk
