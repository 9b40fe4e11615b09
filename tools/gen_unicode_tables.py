"""Generates the Unicode tables of the product from Unicode 16.0 General_Category data:

  openvino_tokenizers_amd/csrc/unicode_tables.inc   per-code-point property nibbles of the hand-written scanners
  openvino_tokenizers_amd/csrc/unicode_gc.inc       General_Category of every code point as a range list (host side:
                                                    the regex compiler builds \\p{..}, \\w, \\d, \\s sets from it)

Why 16.0: the reference pins PCRE2 10.46 (src/CMakeLists.txt:185-189), whose tables are Unicode 16.0; a match under
PCRE2_UTF|PCRE2_UCP (src/utils.cpp:259-261) depends on the code points' General_Category only (for the patterns this
library supports: \\p{..} by category, \\d = Nd, \\w = L|N|Mn|Pc, \\s = Z + the control white space below).

Offline source of the 16.0 data: the Oniguruma engine inside HF `tokenizers` 0.22.2 (pre_tokenizers.Split(Regex)),
which carries exactly Unicode 16.0 -- verified below: it knows U+1C89 (new in 16.0) and does not know U+A7CE / U+1E6C0
(new in 17.0).  Cross-checks printed and asserted:
  * Python `regex` (Unicode 17.0 here): may differ only on code points Oniguruma reports as unassigned (17.0 additions);
  * the image's PCRE2 (10.39, Unicode 14.0 -- the oracle's matcher): may differ only where PCRE2 reports Cn (assigned
    after 14.0), apart from the handful of category changes between 14.0 and 16.0 that are listed.

nibble bits: [1:0] class  0 = other, 1 = \\p{L}, 2 = \\p{N}, 3 = \\s      (L, N, \\s are disjoint)
             [2]   \\p{P}
Layout: two-level table over 128-code-point blocks; identical blocks are shared.
  uc_index[cp >> 7]  -> block id (u16),   uc_blocks[block*64 + ((cp & 127) >> 1)] -> two nibbles.

Usage: python -m tools.gen_unicode_tables
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

CSRC = Path(__file__).resolve().parent.parent / "openvino_tokenizers_amd" / "csrc"
OUT = CSRC / "unicode_tables.inc"
OUT_GC = CSRC / "unicode_gc.inc"
NCP = 0x110000

# order = the enum ovtk::GeneralCategory in csrc/regex_compile.hpp
GC_NAMES = ["Cn", "Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Mc", "Me", "Nd", "Nl", "No", "Pc", "Pd", "Ps", "Pe", "Pi", "Pf", "Po",
            "Sm", "Sc", "Sk", "So", "Zs", "Zl", "Zp", "Cc", "Cf", "Cs", "Co"]
# PCRE2 \s under UCP (PT_SPACE): property Z, or one of the \h / \v characters -- of those only the controls and
# U+180E MONGOLIAN VOWEL SEPARATOR (Cf since Unicode 6.3, still in PCRE2's HSPACE list) are not Z themselves
SPACE_CONTROLS = [0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x85, 0x180E]


def onig_mask(pattern: str) -> np.ndarray:
    """bool[NCP]: code point matches the single-character class `pattern` in Oniguruma (HF tokenizers)."""
    from tokenizers import Regex, pre_tokenizers
    split = pre_tokenizers.Split(Regex("(?:" + pattern + ")+"), "removed", invert=True)  # keeps the matches only
    mask = np.zeros(NCP, dtype=bool)
    allcps = np.concatenate([np.arange(0, 0xD800), np.arange(0xE000, NCP)])
    step = 1 << 15
    for s0 in range(0, len(allcps), step):
        cps = allcps[s0:s0 + step]
        text = "".join(map(chr, cps.tolist()))
        for _, (a, b) in split.pre_tokenize_str(text):
            mask[cps[a:b]] = True
    return mask


def general_categories() -> np.ndarray:
    """u8[NCP]: index into GC_NAMES, from Oniguruma's Unicode 16.0 tables."""
    gc = np.full(NCP, 255, dtype=np.uint8)
    for k, name in enumerate(GC_NAMES):
        if name == "Cs":
            continue
        m = onig_mask(r"\p{" + name + "}")
        assert not (gc[m] != 255).any(), f"{name} overlaps an earlier category"
        gc[m] = k
    gc[0xD800:0xE000] = GC_NAMES.index("Cs")  # surrogates cannot be put into a UTF-8 probe string
    assert not (gc == 255).any(), "code points without a category"
    return gc


def check_version(gc):
    """The engine is at Unicode 16.0: knows 16.0's additions, not 17.0's."""
    n = {name: k for k, name in enumerate(GC_NAMES)}
    assert gc[0x1C89] == n["Lu"] and gc[0x1C8A] == n["Ll"] and gc[0x10D40] == n["Nd"] and gc[0xA7CB] == n["Lu"], "pre-16.0 tables"
    assert gc[0xA7CE] == n["Cn"] and gc[0x1E6C0] == n["Cn"] and gc[0x088F] == n["Cn"], "post-16.0 tables"
    assigned = int((gc != n["Cn"]).sum())
    # Unicode 16.0: 154 998 graphic + format characters, + 65 controls + 137 468 private use + 2 048 surrogates
    assert assigned == 154998 + 65 + 137468 + 2048, assigned


def cross_check_regex(gc):
    import regex
    n_cn = GC_NAMES.index("Cn")
    bad = 0
    for k, name in enumerate(GC_NAMES):
        if name in ("Cs", "Cn"):
            continue
        r = regex.compile(r"\p{" + name + "}")
        for cp in np.flatnonzero(gc == k):
            if not r.match(chr(int(cp))):
                bad += 1  # a category change between 16.0 and 17.0 of an assigned character
    newer = 0
    r_cn = regex.compile(r"\p{Cn}")
    for cp in np.flatnonzero(gc == n_cn):
        if not (0xD800 <= cp < 0xE000) and not r_cn.match(chr(int(cp))):
            newer += 1
    return bad, newer


def cross_check_pcre2(gc):
    """Against the oracle's PCRE2 (Unicode 14.0): differences only on code points PCRE2 holds for unassigned, apart from
    category changes of existing characters (returned)."""
    from oracle import oracle as O

    def pcre2_mask(pattern):
        mask = np.zeros(NCP, dtype=bool)
        rs = O.RegexSplit("(?:" + pattern + ")++", "isolate")
        allcps = np.concatenate([np.arange(0, 0xD800), np.arange(0xE000, NCP)])
        for s0 in range(0, len(allcps), 2048):
            cps = allcps[s0:s0 + 2048]
            text = "".join(map(chr, cps.tolist())).encode("utf-8")
            lens = np.where(cps < 0x80, 1, np.where(cps < 0x800, 2, np.where(cps < 0x10000, 3, 4)))
            offs = np.concatenate([[0], np.cumsum(lens)])
            start = 0
            while True:
                m = rs.match(text, start)
                if m is None:
                    break
                mask[cps[np.searchsorted(offs, m[0]):np.searchsorted(offs, m[1])]] = True
                start = m[1]
        return mask

    cn14 = pcre2_mask(r"\p{Cn}")
    changed = []
    for group in "LNPSZMC":
        ours = np.isin(gc, [k for k, nm in enumerate(GC_NAMES) if nm[0] == group])
        theirs = pcre2_mask(r"\p{" + group + "}")
        d = np.flatnonzero((ours != theirs) & ~cn14)
        d = d[(d < 0xD800) | (d >= 0xE000)]
        changed += [(int(cp), group) for cp in d]
    return int(cn14.sum()), changed


def nibble_table(gc):
    first = np.array([nm[0] for nm in GC_NAMES])
    L = first[gc] == "L"
    N = first[gc] == "N"
    P = first[gc] == "P"
    S = first[gc] == "Z"
    S[SPACE_CONTROLS] = True
    assert not (L & N).any() and not (L & S).any() and not (N & S).any()
    assert L[0x41] and L[0xE9] and L[0x4E00] and L[0x9FA5] and not L[0x1F600] and L[0x2CEA1] and L[0x31350]
    assert N[0x30] and N[0x0663] and N[0xBD] and S[0x20] and S[0x3000] and S[0x85] and P[0xFF0C] and P[0x2E]
    nib = np.zeros(NCP, dtype=np.uint8)
    nib[L] = 1
    nib[N] = 2
    nib[S] = 3
    nib[P] |= 4
    blocks = nib.reshape(-1, 128)
    uniq, inv = np.unique(blocks, axis=0, return_inverse=True)
    packed = (uniq[:, 0::2] | (uniq[:, 1::2] << 4)).astype(np.uint8)  # low nibble = even cp
    return inv.astype(np.uint16), packed, dict(L=int(L.sum()), N=int(N.sum()), S=int(S.sum()), P=int(P.sum()))


def main():
    import tokenizers
    gc = general_categories()
    check_version(gc)
    bad17, newer17 = cross_check_regex(gc)
    cn14, changed14 = cross_check_pcre2(gc)
    print(f"vs regex (Unicode 17.0): {bad17} category changes of assigned characters, {newer17} characters added after 16.0")
    print(f"vs PCRE2 (Unicode 14.0): {len(changed14)} category-group changes of characters assigned in 14.0: "
          f"{[(hex(c), g) for c, g in changed14[:40]]}")
    assert newer17 == 4803, newer17   # Unicode 17.0 added 4 803 characters
    src = f"Oniguruma of HF tokenizers {tokenizers.__version__} (Unicode 16.0.0: the version of PCRE2 10.46, which the reference pins)"
    index, packed, counts = nibble_table(gc)
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_unicode_tables.py -- do not edit.\n")
        f.write(f"// Source: {src}; counts {counts}.\n")
        f.write("// nibble: bits[1:0] 0 other / 1 \\p{L} / 2 \\p{N} / 3 \\s ; bit 2 \\p{P}.\n")
        f.write(f"static const unsigned kUcNumBlocks = {packed.shape[0]};\n")
        f.write(f"static const unsigned short kUcIndex[{len(index)}] = {{\n")
        for i in range(0, len(index), 32):
            f.write(",".join(str(int(x)) for x in index[i:i + 32]) + ",\n")
        f.write("};\n")
        flat = packed.reshape(-1)
        f.write(f"static const unsigned char kUcBlocks[{len(flat)}] = {{\n")
        for i in range(0, len(flat), 64):
            f.write(",".join(str(int(x)) for x in flat[i:i + 64]) + ",\n")
        f.write("};\n")
    starts = np.flatnonzero(np.concatenate([[True], gc[1:] != gc[:-1]]))
    with open(OUT_GC, "w") as f:
        f.write("// GENERATED by tools/gen_unicode_tables.py -- do not edit.\n")
        f.write(f"// Source: {src}.\n")
        f.write("// General_Category as a range list: code points [kGcStart[i], kGcStart[i+1]) have category kGcValue[i]\n")
        f.write("// (index into: " + " ".join(GC_NAMES) + ").\n")
        f.write(f"static const unsigned kGcRanges = {len(starts)};\n")
        f.write(f"static const unsigned kGcStart[{len(starts) + 1}] = {{\n")
        vals = list(starts) + [NCP]
        for i in range(0, len(vals), 16):
            f.write(",".join(str(int(x)) for x in vals[i:i + 16]) + ",\n")
        f.write("};\n")
        f.write(f"static const unsigned char kGcValue[{len(starts)}] = {{\n")
        for i in range(0, len(starts), 48):
            f.write(",".join(str(int(gc[s])) for s in starts[i:i + 48]) + ",\n")
        f.write("};\n")
    print(f"wrote {OUT} ({packed.shape[0]} blocks, {counts}) and {OUT_GC} ({len(starts)} ranges)")


if __name__ == "__main__":
    main()
