"""csrc/unicode_tables.inc / unicode_gc.inc (Unicode 16.0, the version of the PCRE2 10.46 the reference pins) against
their sources, and the scanners on code points the oracle's older PCRE2 (10.39, Unicode 14.0) does not know."""
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

CSRC = Path(__file__).resolve().parent.parent / "openvino_tokenizers_amd" / "csrc"
GC_NAMES = ["Cn", "Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Mc", "Me", "Nd", "Nl", "No", "Pc", "Pd", "Ps", "Pe", "Pi", "Pf", "Po",
            "Sm", "Sc", "Sk", "So", "Zs", "Zl", "Zp", "Cc", "Cf", "Cs", "Co"]


def _ints(txt, name):
    body = re.search(name + r"\[\d+\] = \{(.*?)\};", txt, re.S).group(1)
    return np.array(body.replace("\n", "").rstrip(",").split(","), dtype=np.int64)


def load_tables():
    txt = (CSRC / "unicode_tables.inc").read_text()
    idx, blk = _ints(txt, "kUcIndex"), _ints(txt, "kUcBlocks")
    cps = np.arange(0x110000)
    byte = blk[idx[cps >> 7] * 64 + ((cps & 127) >> 1)]
    return np.where(cps & 1, byte >> 4, byte & 15)


def load_gc():
    txt = (CSRC / "unicode_gc.inc").read_text()
    start, val = _ints(txt, "kGcStart"), _ints(txt, "kGcValue")
    return np.repeat(val, np.diff(start))


@pytest.fixture(scope="module")
def pcre2_assigned():
    from tools.gen_unicode_tables import NCP  # noqa: F401
    from tools import gen_unicode_tables as G
    # \P{Cn} in the oracle's PCRE2: the code points its Unicode 14.0 tables know
    rs_mask = np.zeros(0x110000, dtype=bool)
    rs = O.RegexSplit(r"(?:\P{Cn})++", "isolate")
    allcps = np.concatenate([np.arange(0, 0xD800), np.arange(0xE000, 0x110000)])
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
            rs_mask[cps[np.searchsorted(offs, m[0]):np.searchsorted(offs, m[1])]] = True
            start = m[1]
    return rs_mask, G


def test_tables_equal_pcre2_where_it_knows_the_code_point(pcre2_assigned):
    """Oracle-side check: on every code point assigned in Unicode <= 14.0 the product's classes are the ones the oracle's
    PCRE2 matches with (so device == oracle on text made of such characters)."""
    known, G = pcre2_assigned
    nib = load_tables()
    for bits, pat in ((1, r"\p{L}"), (2, r"\p{N}"), (3, r"\s")):
        want = np.zeros(0x110000, dtype=bool)
        rs = O.RegexSplit("(?:" + pat + ")++", "isolate")
        allcps = np.flatnonzero(known)
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
                want[cps[np.searchsorted(offs, m[0]):np.searchsorted(offs, m[1])]] = True
                start = m[1]
        assert np.array_equal(((nib & 3) == bits) & known, want), pat


def test_nibbles_follow_the_general_categories():
    nib, gc = load_tables(), load_gc()
    first = np.array([n[0] for n in GC_NAMES])[gc]
    assert np.array_equal((nib & 3) == 1, first == "L") and np.array_equal((nib & 3) == 2, first == "N")
    assert np.array_equal((nib & 4) != 0, first == "P")
    space = first == "Z"
    space[[9, 10, 11, 12, 13, 0x85, 0x180E]] = True
    assert np.array_equal((nib & 3) == 3, space)


def test_general_categories_are_unicode_16():
    """16.0's additions are there (U+1C89 CYRILLIC CAPITAL LETTER TJE ...), 17.0's are not; against Python `regex`
    (Unicode 17.0 in this image) the only differences are code points still unassigned in 16.0."""
    regex = pytest.importorskip("regex")
    gc = load_gc()
    n = {name: k for k, name in enumerate(GC_NAMES)}
    assert gc[0x1C89] == n["Lu"] and gc[0x10D40] == n["Nd"] and gc[0xA7CE] == n["Cn"] and gc[0x1E6C0] == n["Cn"]
    assert int((gc != n["Cn"]).sum()) == 154998 + 65 + 137468 + 2048   # Unicode 16.0's character count
    pats = {"L": regex.compile(r"\p{L}"), "N": regex.compile(r"\p{N}"), "P": regex.compile(r"\p{P}")}
    first = np.array([x[0] for x in GC_NAMES])[gc]
    for cp in range(0x110000):
        if 0xD800 <= cp < 0xE000 or gc[cp] == n["Cn"]:
            continue
        ch = chr(cp)
        for g, r in pats.items():
            assert bool(r.match(ch)) == (first[cp] == g), hex(cp)


def test_scanners_on_code_points_newer_than_the_oracle(backend, pcre2_assigned):
    """Code points assigned in Unicode 15.0 / 15.1 / 16.0 (unknown to the oracle's PCRE2): the device splits a string
    holding them exactly as the oracle splits the same string with each of them replaced by a long-established character
    of the same class (piece boundaries compared in characters)."""
    from openvino_tokenizers_amd.ops import RegexSplit
    from tests.util import one_string_per_row
    from tools.make_tokenizers import GPT2_PATTERN, LLAMA3_PATTERN
    known, _ = pcre2_assigned
    gc = load_gc()
    first = np.array([x[0] for x in GC_NAMES])[gc]
    new = np.flatnonzero(~known & (gc != 0) & (gc != GC_NAMES.index("Cs")) & (gc != GC_NAMES.index("Co")))
    new = new[(first[new] == "L") | (first[new] == "N") | (first[new] == "P") | (first[new] == "S") | (first[new] == "M")]
    rng = np.random.default_rng(16)
    n_each, n_rand = (12, 40) if backend.name == "emu" else (40, 150)   # the emulator is slow
    pick = np.concatenate([new[first[new] == g][:n_each] for g in "LNPSM"] + [rng.choice(new, n_rand, replace=False)])
    stand_in = {"L": "é", "N": "٣", "P": "，", "S": "€", "M": "́"}   # old characters of the same group ('other' for P/S/M)
    frames = ["a{}b", " {}{} x", "1{} {}", "{}'s", "\n{}\n", "元{}元 {}"]
    real, fake = [], []
    for cp in pick:
        for f in frames:
            real.append(f.replace("{}", chr(int(cp))))
            fake.append(f.replace("{}", stand_in[first[cp]]))
    for pattern in (GPT2_PATTERN, LLAMA3_PATTERN):
        pat = np.frombuffer(pattern.encode(), np.uint8)
        ri, fi = one_string_per_row(real), one_string_per_row(fake)
        ref = O.RegexSplit(pattern, "isolate")(*fi)
        got = RegexSplit("isolate", lib=backend.lib).evaluate(backend.data(ri) + [pat])
        gb, ge = backend.host(got[2]), backend.host(got[3])
        grb, gre = backend.host(got[0]), backend.host(got[1])
        for i, (r, f) in enumerate(zip(real, fake)):
            rbytes, fbytes = r.encode(), f.encode()
            to_char_r = {len(r[:k].encode()): k for k in range(len(r) + 1)}
            to_char_f = {len(f[:k].encode()): k for k in range(len(f) + 1)}
            want = [(to_char_f[a - fi[2][i]], to_char_f[b - fi[2][i]])
                    for a, b in zip(ref[2][ref[0][i]:ref[1][i]], ref[3][ref[0][i]:ref[1][i]])]
            have = [(to_char_r[a - ri[2][i]], to_char_r[b - ri[2][i]]) for a, b in zip(gb[grb[i]:gre[i]], ge[grb[i]:gre[i]])]
            assert want == have, (r, rbytes, fbytes, want, have)
