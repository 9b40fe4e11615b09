"""csrc/unicode_tables.inc (the property nibbles the scanners use) against its sources."""
import re
import unicodedata
from pathlib import Path

import numpy as np
import pytest

INC = Path(__file__).resolve().parent.parent / "openvino_tokenizers_amd" / "csrc" / "unicode_tables.inc"


def load_tables():
    txt = INC.read_text()
    idx = np.array(re.search(r"kUcIndex\[\d+\] = \{(.*?)\};", txt, re.S).group(1).replace("\n", "").rstrip(",").split(","), dtype=np.int64)
    blk = np.array(re.search(r"kUcBlocks\[\d+\] = \{(.*?)\};", txt, re.S).group(1).replace("\n", "").rstrip(",").split(","), dtype=np.int64)
    cps = np.arange(0x110000)
    byte = blk[idx[cps >> 7] * 64 + ((cps & 127) >> 1)]
    return np.where(cps & 1, byte >> 4, byte & 15)


def test_tables_equal_pcre2():
    """Regenerating from the PCRE2 the oracle uses gives the committed table (so scanner == PCRE2 on every code point)."""
    from tools.gen_unicode_tables import property_mask
    nib = load_tables()
    for bits, pat in ((1, r"\p{L}"), (2, r"\p{N}"), (3, r"\s")):
        assert np.array_equal((nib & 3) == bits, property_mask(pat)), pat
    assert np.array_equal((nib & 4) != 0, property_mask(r"\p{P}"))


def test_differences_to_newer_unicode_are_new_code_points_only():
    """The reference pins PCRE2 10.46 (Unicode 16); the tables come from the image's PCRE2 (Unicode 14).  Against the
    newest tables available offline (Python `regex`), every differing code point must be one that was still
    unassigned in Unicode 13 (Python's unicodedata) -- i.e. text made of long-established characters, which is what
    tests and bench generate, is classified identically by both."""
    regex = pytest.importorskip("regex")
    nib = load_tables()
    pats = {1: regex.compile(r"\p{L}"), 2: regex.compile(r"\p{N}")}
    diff = []
    for cp in range(0x110000):
        if 0xD800 <= cp < 0xE000:
            continue
        ch = chr(cp)
        want = 1 if pats[1].match(ch) else (2 if pats[2].match(ch) else None)
        have = nib[cp] & 3
        if want is not None and have != want or want is None and have in (1, 2):
            diff.append(cp)
    assert all(unicodedata.category(chr(cp)) == "Cn" for cp in diff), [hex(c) for c in diff[:10]]
    print(f"{len(diff)} code points differ between PCRE2's Unicode 14 tables and regex {regex.__version__}")
