"""DeepSeek-V3's main split pattern and o200k_base on lookup_span_kernel (csrc/span_fam.hpp, csrc/fam_literal.hpp; round 5).

Three tiers of evidence, each against the one above it:
  * the oracle's PCRE2 (PCRE2_UTF | PCRE2_UCP, as src/utils.cpp:256-272 compiles the pattern) + the BPE restatement: the truth;
  * the LITERAL matchers ds3_match_end / o200k_match_end (the alternatives in order, one position at a time): calls of a few rows run
    them and nothing else (encode_small_kernel<kFusedSeq>), so `test_literal_matchers_against_pcre2` pins them -- crafted rows for every
    rule of the two patterns (marks on both sides of the letter / punctuation line, the upper part giving back to its last Lm / Lo / M
    character, contractions behind words only, slashes behind line breaks, gaps) and random rows of the fuzzers' fragments;
  * the rule ALGEBRA on bit masks (span_flags_ds3 / span_flags_o200k): `test_rule_algebra_against_the_literal_matchers` runs
    tests/emu/l3_flags_fuzz.cpp for the two families (a block per case against the literal matcher on the complete rows), the batch
    tests below run the kernel (> 256 rows) against the oracle -- and the emulator build checks every block it scans against the literal
    matcher inside the kernel.
Reference behaviour: src/regex_split.cpp:205-324 (`isolate`: the text between two matches is a piece of its own, :262-284).
"""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.test_span_kernel import _filler, fused_vs_oracle, rows_of
from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
from oracle import oracle as O
from tests.util import BpeTok, assert_same
from tools.workloads import MODEL_PATTERNS, TextModel, ragged_rows

ROOT = Path(__file__).resolve().parent.parent
FAMILIES = ["deepseek-v3", "o200k"]

FRAGS = ["the", "token", "a", "I", "x", "Zq", "hello", "World", " ", " ", " ", "  ", "   ", "\n", "\n", "\r\n", "\r", "\t", "\n\n", " \n", "\n ", "\n    ", "\x0b",
         ",", ".", "!", "?", "!?", "...", "--", "(", ")", "\"", "'", "''", "$", "%", "#", "@", "/", "\\", "_", "-", "*",
         "'s", "'t", "'m", "'d", "'re", "'ve", "'ll", "'S", "'T", "'RE", "'Ve", "'lL", "'r", "'x", "don't", "we'll", "I'M", "it's",
         "1", "12", "123", "1234", "12345", "1234567", "3.14", "1,000", "a1b2", "9x",
         "é", "naïve", "straße", "×", " ", "«", "»", "—", "…", "€", " ", " ", "　", "", "привет", "Ωμέγα", "日本語", "の", "。", "😀", "😀😁", "𐐀", "K",
         "！", "שלום", "سلام",
         "A", "B", "AB", "ABC", "HTTP", "Camel", "camelCase", "XMLHttpRequest", "iPhone", "aB", "Ab", "aBc", "ABc", "abC", "A1", "Z",
         "́", "́́", "é", "É", "́a", "́A", "!́", "!!́", " ́", "⃝", "ः",
         "ǅ", "ʰ", "日", "日A", "A日", "A日B", "日Ab", "Пр", "ПР", "пР", "É", "Été", "Ω", "A's", "a'T", "B'Re", "日's", "́'s", "'́",
         "//", "\n/", "\n//", "!\n/", "*/\n/*", "/\n", "\n/\n", "\r\n/", "!\n/!\n/a", "*/", "/*",
         "\x01", "\x7f", "\x1b[0m", "­", "​", "‍", "﻿", "", "\U000e0001",
         "1a", "1A", "12ab", "a1", "­a", "​B", "\x01a", "!a", "!ab", "!A", "!é", "!aé", "?b́", "#tag", "@user", "$x", "_id", "-v", "(a", " !a", "!!a", ".com",
         "٣", "²", "１", "½", "٣a", "ſ", "'ſ"]

CRAFTED = [
    # o200k: the upper part gives back to its last character of both kinds; CamelCase; upper-case runs in front of a lower-case letter
    "中A", "中文A", "A中", "A中B", "ABC中DEF ", "CamelCaseWord", "HTTPServer", "XMLHttpRequest x", "ABC", "ABCd", "aBC", "aBC's", "ABC's", "ABC'S x",
    "́AB ", "́ab", " ́AB", "ÁB", "áB", "AB́", "AB́C", "日本語ABC", "ABC日本語", "日本語abc", "abc日本語ABCdef",
    # marks on both sides
    "!́", "!!́", "!́!́", "!!́!́", " !́", "!́a", "!!́a", "́", "́́", "á", "1́", "1́a", "\ńa", "\t́",
    "!!́'s", "á's", "́'s", "!́'s",
    # contractions: behind a word only; chains
    "don't", "DON'T", "don'T", "don'tX", "don'Tx", "don'TX", "don't́X", "it's's", "it's's's", "a's'S'll", "a'll'll", " 's", "'s", "x 's", "1's", "!'s", "a''s",
    "a'rE're", "t'lL ", "a'r", "a'l", "a'v", "we'LLgo", "we'llGo",
    # slashes behind line breaks
    "!\n/", "!\n/x", "!\n//!", "*/\n/* x */\n/* y", "!\n/\n", "!\n/!\n/a", "a\n/", "1\n/", "\n/", "! \n/", "!\r\n/\r\n//x", "/\n/", "//\n//\n//", "!\n/a", "!\n/ a",
    # DeepSeek-V3: ASCII punctuation + ASCII letters, gaps, the optional character
    "!ab", "!abé", "!ab́", "!éa", "!!ab", " !ab", "!ab!cd", "_id", ".com", "#tag1", "@user_name", "(a)", "a.b.c", "¡ab", "—ab",
    "1", "12", "123 456", "12ab", "1 ab", "1\nab", "a1b2c3", "\x01\x02ab", "\x01\x02", "\x01 \x02", "٣a", "٣٣a", "٣ a", "1!a", "1!!", "1 !", "­a", "­­", "​B​",
    "a­", "a­b", "2024-01-01", "3.14", "1,000", "x=1;", "\x7f", "\x7fa", "1́", "١٢٣",
    # white space
    "a  b", "a \n b", "a\n\nb", "a \n", " \n ", "\n \n", "a   ", "   a", "\t\ta", "a b", "a  b", "!\n\n x", "!\n \n", "! \n", "!\r\n\r\na",
]


def test_rule_algebra_against_the_literal_matchers():
    build = ROOT / "tests" / "emu" / "build"
    build.mkdir(parents=True, exist_ok=True)
    exe = build / "l3_flags_fuzz"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-Wno-attributes", "-I", str(ROOT / "tests" / "emu"),
                    "-I", str(ROOT / "openvino_tokenizers_amd" / "csrc"), str(ROOT / "tests" / "emu" / "l3_flags_fuzz.cpp"), "-o", str(exe)], check=True)
    for family in ("3", "4"):
        r = subprocess.run([str(exe), "0", "8", "300", family], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:]
        assert "cases" in r.stdout.splitlines()[-1]


def _random_rows(rng, n, max_frags):
    rows = []
    for _ in range(n):
        k = int(rng.integers(1, max_frags))
        s = "".join(FRAGS[int(i)] for i in rng.integers(0, len(FRAGS), size=k))
        if rng.random() < 0.3:
            s = s.replace(" ", "")
        rows.append(s.encode())
    return rows


@pytest.mark.parametrize("name", FAMILIES)
def test_literal_matchers_against_pcre2(backend, name):
    """Calls of at most 200 rows: one launch of encode_small_kernel<kFusedSeq>, lane 0 walking each row with the literal matcher."""
    tok = BpeTok.load("llama3_small")
    pattern = MODEL_PATTERNS[name]
    crafted = [s.encode() for s in CRAFTED]
    for at in range(0, len(crafted), 150):
        fused_vs_oracle(backend, tok, rows_of(crafted[at:at + 150]), pattern=pattern, what=f"{name}: crafted rows from {at}")
    rng = np.random.default_rng(31)
    for rnd in range(6 if backend.name == "emu" else 30):
        rows = _random_rows(rng, 180, 14)
        fused_vs_oracle(backend, tok, rows_of(rows), pattern=pattern, what=f"{name}: random rows, round {rnd}")


@pytest.mark.parametrize("name", FAMILIES)
def test_the_scan_on_batches(backend, name):
    """More than 256 rows: lookup_span_kernel<kSpanDs3 / kSpanO200k>, the rows back to back in its 2 048-byte blocks."""
    tok = BpeTok.load("llama3_small")
    pattern = MODEL_PATTERNS[name]
    rng = np.random.default_rng(32)
    crafted = [s.encode() for s in CRAFTED]
    rows = crafted + _random_rows(rng, 300, 40)
    order = rng.permutation(len(rows))
    fused_vs_oracle(backend, tok, rows_of([rows[i] for i in order]), pattern=pattern, what=f"{name}: crafted and random rows in one batch")
    b, e, c = TextModel(79, "mixed").batch(288, 512)
    rb, re_ = ragged_rows(288)
    fused_vs_oracle(backend, tok, [rb, re_, b, e, c], pattern=pattern, what=f"{name}: mixed text at config 4's row length")


@pytest.mark.parametrize("name", FAMILIES)
def test_what_is_not_local(backend, name):
    """Runs that cross lanes (32 bytes) and blocks (2 048 bytes): upper-case runs (o200k: whether a piece starts at a run's first letter
    depends on its end), digit runs, marks, white space, line breaks with slashes; pieces longer than a block."""
    if backend.name == "emu" and name == "deepseek-v3":
        pytest.skip("the emulator leg of this pattern runs on the GPU tier")
    tok = BpeTok.load("llama3_small")
    pattern = MODEL_PATTERNS[name]
    rng = np.random.default_rng(33)
    rows = [b"A" * n for n in (1, 31, 32, 33, 64, 100, 2047, 2048, 2049, 5000)] + [b"A" * n + b"b" for n in (1, 31, 32, 33, 2047, 2048, 2100, 4100)]
    rows += [("日" + "A" * n).encode() for n in (1, 30, 31, 700, 2100)] + [("日" + "A" * n + "b").encode() for n in (1, 30, 31, 700, 2100)]
    rows += [b"x" * k + b"ABCDEFGH" * m + t for k in (20, 28, 30) for m in (1, 4, 9) for t in (b"", b"i", b" ", b"'s", b"'S")]
    rows += [b"1" * n for n in (1, 3, 4, 40, 64, 65, 2047, 2049)] + [b"1" * n + b"ab" for n in (1, 2, 31, 32, 33, 2048)]
    rows += [("e" + "́" * n).encode() for n in (1, 15, 16, 17, 1100)] + [("!!" + "́" * n + "a").encode() for n in (1, 15, 16, 1100)]
    rows += [b"!" + b"\n/" * n + b"x" for n in (1, 15, 16, 17, 40)] + [b"!\n" + b"/" * n + b"!" for n in (1, 31, 32, 33, 70)]
    rows += [b"!" * n + b"ab" for n in (1, 2, 31, 32, 33)] + [b"a" + b" " * n + b"B" for n in (1, 2, 31, 32, 33, 2046, 2047, 2100)]
    rows += [b"\n" * n + b"a" for n in (1, 5, 32, 100)] + [b"a" + b" " * n + b"\n" + b" " * m + b"b" for n in (0, 1, 31, 32) for m in (0, 1, 31, 33)]
    rows += [b"x" * k + "it's We'LL I'M they'Re A'S '".encode() + b"s" for k in range(20, 34)]
    rows += [_filler(rng, int(rng.integers(1, 400))) for _ in range(60)]
    order = rng.permutation(len(rows))
    strings = [rows[i] for i in order]
    strings += strings[: max(0, 300 - len(strings))]
    fused_vs_oracle(backend, tok, rows_of(strings), pattern=pattern, what=f"{name}: runs across lanes and blocks")


@pytest.mark.parametrize("name", FAMILIES)
def test_rows_the_scan_leaves_and_blocks_it_declines(backend, name):
    """Skipped strings and empty rows go to the generic kernel (literal matcher); o200k blocks with a non-ASCII digit or U+017F go to lane
    0 of the span kernel; a character the block's end cuts must not look like one of them."""
    tok = BpeTok.load("llama3_small")
    pattern = MODEL_PATTERNS[name]
    rng = np.random.default_rng(34)
    odd = ["١٢٣٤", "x²", "１２３４５", "it'ſ", "ſt", "½", "12٣٤4", "a's's's's's's's's's"]
    strings = []
    for i in range(320):
        s = _filler(rng, int(rng.integers(1, 500)))
        if i % 9 == 0:
            s += odd[(i // 9) % len(odd)].encode() + _filler(rng, int(rng.integers(0, 80)))
        if i % 17 == 0:
            s = _filler(rng, 2046) + "忿忿".encode() + _filler(rng, 50)
        if i % 23 == 0:
            s = b""
        strings.append(s)
    skips = np.zeros(len(strings), np.uint8)
    skips[::7] = 1
    fused_vs_oracle(backend, tok, rows_of(strings), skips=skips, pattern=pattern, what=f"{name}: skipped / empty rows, blocks for lane 0")


def test_a_handle_without_a_memo_takes_the_compiled_dfa(backend):
    """cache_capacity = 0: no piece memo, no span kernel -- the pattern runs as any other pattern does."""
    tok = BpeTok.load("llama3_small")
    attrs = dict(tok.attrs)
    attrs["cache_capacity"] = 0
    tok0 = BpeTok(tok.vocab, tok.merges, tok.added, tok.pattern, **attrs)
    rng = np.random.default_rng(35)
    rows = _random_rows(rng, 300, 20)
    fused_vs_oracle(backend, tok0, rows_of(rows), pattern=MODEL_PATTERNS["o200k"], what="o200k without a memo")


def test_deepseek_v3_three_splits_in_a_row(backend):
    """DeepSeek-V3's pre-tokenizer is a Sequence of three `Split`s (tokenizer.json: digits of 1-3, CJK runs, the main pattern; each
    `isolated`), i.e. three RegexSplit nodes (tokenizer_pipeline.py:392-457 builds one per Split): the third one's rows hold SEVERAL
    strings each, which the span kernel leaves to the compiled DFA (api_encode.cpp start_encode).  Oracle: the same three in a row."""
    tok = BpeTok.load("llama3_small")
    pats = [r"\p{N}{1,3}", "[一-龥぀-ゟ゠-ヿ]+", MODEL_PATTERNS["deepseek-v3"]]
    rng = np.random.default_rng(36)
    rows = [s.encode() for s in CRAFTED] + _random_rows(rng, 300, 30)
    inputs = rows_of(rows)
    ref = [np.asarray(x) for x in inputs]
    got = backend.data(inputs)
    for pat in pats[:2]:
        ref = list(O.RegexSplit(pat, "isolate")(*ref[:5])[:5])
        got = list(RegexSplit("isolate", lib=backend.lib).evaluate(list(got[:5]) + [np.frombuffer(pat.encode(), np.uint8)])[:5])
    assert len(ref[2]) > len(rows)          # rows of several strings reach the third split
    ref_ids = tok.oracle()(*O.RegexSplit(pats[2], "isolate")(*ref)[:5])
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    got_ids = fused.evaluate(list(got) + [np.frombuffer(pats[2].encode(), np.uint8)], tok.consts)
    assert_same(ref_ids, got_ids, backend.host, "DeepSeek-V3: three splits in a row")
