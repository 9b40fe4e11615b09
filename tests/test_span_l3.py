"""lookup_span_kernel<kSpanLlama3> (csrc/span_l3.hpp): the Llama-3 family's rule algebra on bit masks, several rows per 2 048-byte
scan block, whatever the script (round 5).

Two tiers of evidence:
  * `test_rule_algebra_against_the_literal_matcher` builds tests/emu/l3_flags_fuzz.cpp (the scanner alone on the SIMT emulator, one
    block per case) and runs it: random blocks of rows -- digit runs of any length, line breaks and indentation inside white space,
    contractions in both cases, non-ASCII letters / white space / punctuation, broken UTF-8, rows of one byte, blocks that begin
    inside a row and blocks cut inside one -- against llama3_match_end on the COMPLETE rows (that matcher is pinned against PCRE2 by
    tests/test_split_rules.py).  `tools/fuzz_span.py`'s counterpart for this scanner; thousands of seeds were run while it was written.
  * the fused encode (`ovtk_encode_run`) against the oracle chain (PCRE2 + the BPE restatement) on batches that take the span kernel
    (> 256 rows): what is not local in the rules at sizes that cross lanes and blocks (forty digits, a hundred line breaks, 2 100
    blanks), rows that end where the rules look ahead, text the algebra hands to the literal matcher (non-ASCII digits, U+017F),
    pieces longer than a block; the three patterns of the family.  On the emulator every block is also checked against the literal
    matcher inside the kernel (span_kernel.hpp, OVTK_SIMT_EMULATOR).
Reference behaviour: src/regex_split.cpp:205-324 runs the pattern per string, PCRE2_UTF | PCRE2_UCP (src/utils.cpp:256-272).
"""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.test_span_kernel import _filler, fused_vs_oracle, rows_of
from tests.util import BpeTok
from tools.workloads import MODEL_PATTERNS, TextModel, ragged_rows

ROOT = Path(__file__).resolve().parent.parent


def test_rule_algebra_against_the_literal_matcher():
    build = ROOT / "tests" / "emu" / "build"
    build.mkdir(parents=True, exist_ok=True)
    exe = build / "l3_flags_fuzz"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-Wno-attributes", "-I", str(ROOT / "tests" / "emu"),
                    "-I", str(ROOT / "openvino_tokenizers_amd" / "csrc"), str(ROOT / "tests" / "emu" / "l3_flags_fuzz.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), "0", "24", "400"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "cases" in r.stdout.splitlines()[-1]


def _pattern(name, tok):
    return MODEL_PATTERNS.get(name, tok.pattern)


def _unlocal_rows(rng):
    """What the floods carry from lane to lane and from block to block."""
    nb = " "
    rows = [b"1" * n for n in (1, 2, 3, 4, 31, 32, 33, 40, 63, 64, 65, 96, 97, 100, 2047, 2048, 2049)]
    rows += [b"x" * k + b"7" * n + b"y" for k in (29, 30, 31, 32) for n in (3, 4, 5, 34, 67)]
    rows += [b"\n" * n + b"a" for n in (1, 4, 5, 31, 32, 33, 64, 100)] + [b"!" + b"\n" * n + b" a" for n in (1, 5, 32, 70)]
    rows += [b"a" + b" " * n + b"\n" + b" " * m + b"b" for n in (0, 1, 30, 31, 32, 64) for m in (0, 1, 4, 31, 33)]
    rows += [b"def f(x):\n" + b"    " * d + b"return x\n\n\n" + b"\t" * d + b"y = 1\r\n\r\n" for d in range(0, 12)]
    rows += [b"a" + b" " * n for n in (1, 2, 31, 32, 33, 2046, 2047, 2100)] + [b" " * n + b"a" for n in (31, 32, 33, 64, 2047, 2048)]
    rows += [b"x" * k + b" \n \n  \n" + b"z" * 5 for k in (24, 25, 26, 27, 28, 29, 30, 31, 32)]
    rows += [("　" * n + "あ").encode() for n in (1, 10, 11, 12, 30)] + [("a" + nb * n + "b").encode() for n in (1, 15, 16, 17, 40)]
    rows += [b"x" * k + "it's We'LL I'M they'Re '".encode() + b"s" for k in range(20, 34)]
    rows += [("é" * n).encode() for n in (15, 16, 17, 1023, 1024, 1025)] + [("日" * n + "!").encode() for n in (10, 11, 682, 683, 700)]
    rows += [b"!" * n + b"a" for n in (1, 2, 31, 32, 33)] + [("—" * n + "a").encode() for n in (1, 2, 10, 11)] + ["€a €b €€c".encode()]
    rows += [_filler(rng, int(rng.integers(1, 400))) for _ in range(40)]
    return rows


@pytest.mark.parametrize("name", ["llama3", "qwen2", "cl100k"])
def test_what_is_not_local(backend, name):
    """Digit groups, line-break runs, white-space runs and characters that cross lanes (32 bytes) and blocks (2 048 bytes)."""
    if backend.name == "emu" and name == "qwen2":
        pytest.skip("the emulator leg of this pattern runs on the GPU tier")
    tok = BpeTok.load("llama3_small")
    rng = np.random.default_rng(5)
    base = _unlocal_rows(rng)
    order = rng.permutation(len(base))
    strings = [base[i] for i in order] + [base[i] for i in order[::-1]][: max(0, 300 - len(base))]
    fused_vs_oracle(backend, tok, rows_of(strings), pattern=_pattern(name, tok), what=f"{name}: floods across lanes and blocks")


def test_rows_that_end_where_the_rules_look_ahead(backend):
    """A row's last bytes against the next row's first: nothing may be looked at across the boundary (contraction letters, the
    character behind a single O character, the line breaks behind an O run, white space behind a line break, digits)."""
    ends = [b"abc ", b"abc  ", b"abc\n", b"abc \n", b"abc\n ", b"abc'", b"abc 'L", b"abc'R", b"12", b"1234", b"x!", b"x !", b"x\t", b"!\n", b"!\n\n", b" ", b"\n", b"'",
            "café".encode(), "x ".encode(), "日".encode(), b"a\r", b"a \r\n "]
    starts = [b"s next", b"LL be", b"e there", b"t", b" x", b"  x", b"'s", b"'ll go", b"34", b"5", b"d", b"\nq", b"\n\n q", b" ", b"!", b"a", "été".encode(),
              "　x".encode(), b"\r\nz", b"1", b"\t\tx"]
    strings = []
    for a in ends:
        for s in starts:
            strings += [a, s]
    tok = BpeTok.load("llama3_small")
    for name in ("llama3", "cl100k"):
        fused_vs_oracle(backend, tok, rows_of(strings[:600]), pattern=_pattern(name, tok), what=f"{name}: row boundaries")
        if backend.name == "emu":
            break


def test_blocks_the_literal_matcher_takes(backend):
    """Non-ASCII digits (digit groups count characters) and U+017F (folds to `s`) are outside the algebra: such a BLOCK is matched by
    lane 0; the blocks around it are not.  And a character that a cut block's end cuts must not look like one of them."""
    rng = np.random.default_rng(9)
    odd = ["١٢٣٤", "x²", "１２３４５", "it'ſ", "ſt", "½", "12٣٤4"]
    strings = []
    for i in range(320):
        s = _filler(rng, int(rng.integers(1, 500)))
        if i % 9 == 0:
            s += odd[(i // 9) % len(odd)].encode() + _filler(rng, int(rng.integers(0, 80)))
        if i % 17 == 0:   # U+5FFF .. cut by a block end: e5 bf | bf reads as U+017F
            s = _filler(rng, 2046) + "忿忿".encode() + _filler(rng, 50)
        strings.append(s)
    tok = BpeTok.load("llama3_small")
    fused_vs_oracle(backend, tok, rows_of(strings), what="llama3: blocks with non-ASCII digits / U+017F")


def test_mixed_text_and_long_rows(backend):
    tok = BpeTok.load("llama3_small")
    b, e, c = TextModel(77, "mixed").batch(288, 512)
    rb, re_ = ragged_rows(288)
    fused_vs_oracle(backend, tok, [rb, re_, b, e, c], what="llama3: mixed text at config 4's row length")
    if backend.name == "emu":
        return
    for name in ("qwen2", "cl100k"):
        fused_vs_oracle(backend, tok, [rb, re_, b, e, c], pattern=_pattern(name, tok), what=f"{name}: mixed text")
    b, e, c = TextModel(78, "mixed").batch(300, 5000)
    rb, re_ = ragged_rows(300)
    fused_vs_oracle(backend, tok, [rb, re_, b, e, c], what="llama3: mixed text, rows of 5 000 bytes")
