"""The GPU split scanners vs PCRE2 (the oracle runs the real matcher with PCRE2_UTF|PCRE2_UCP, src/utils.cpp:259-261).

The scanners replace regex matching by a local piece-start predicate (csrc/split_device.hpp); this file is the
evidence that the predicate is equivalent: every string up to 4 symbols over an alphabet with one representative
per behaviour class (letters incl. the contraction letters, digit, ASCII space, tab, newline, NBSP, apostrophe,
punctuation, 2/3/4-byte letters/symbols), plus random long strings that cross the 768-byte LDS chunk boundary.
"""
import itertools

import numpy as np
import pytest

from openvino_tokenizers_amd.ops import RegexSplit
from oracle import oracle as O
from tests.util import assert_same, one_string_per_row
from tools.make_tokenizers import GPT2_PATTERN, LLAMA3_PATTERN

DIGITS_PATTERN = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"
ALPHABET = ["a", "s", "t", "r", "e", "l", "v", "1", " ", "\t", "\n", " ", "'", "!", "é", "元", "，", "😀", "٣", "　"]


def check(backend, pattern, strings, behaviour="isolate", invert=False, max_splits=-1, ref_pattern=None):
    """ref_pattern: what the oracle's PCRE2 (10.39) must be given to mean what `pattern` means in the reference's 10.46."""
    inputs = one_string_per_row(strings)
    ref = O.RegexSplit(ref_pattern or pattern, behaviour, invert, max_splits)(*inputs)
    got = RegexSplit(behaviour, invert, max_splits, lib=backend.lib).evaluate(
        backend.data(inputs) + [np.frombuffer(pattern.encode(), np.uint8)])
    try:
        assert_same(ref[:4], got[:4], backend.host, "RegexSplit")
    except AssertionError:
        # name the first offending string
        rb, re_ = ref[0], ref[1]
        gb, ge = backend.host(got[0]), backend.host(got[1])
        for i, s in enumerate(strings):
            r = list(zip(ref[2][rb[i]:re_[i]].tolist(), ref[3][rb[i]:re_[i]].tolist()))
            g = list(zip(backend.host(got[2])[gb[i]:ge[i]].tolist(), backend.host(got[3])[gb[i]:ge[i]].tolist()))
            base = int(inputs[2][i])
            if [(a - base, b - base) for a, b in r] != [(a - base, b - base) for a, b in g]:
                raise AssertionError(f"{s!r}: PCRE2 {[(a - base, b - base) for a, b in r]} scanner "
                                     f"{[(a - base, b - base) for a, b in g]}")
        raise


@pytest.mark.parametrize("pattern", [GPT2_PATTERN, DIGITS_PATTERN], ids=["gpt2", "gpt2-digits"])
def test_exhaustive_short_strings(backend, pattern):
    if backend.name == "emu":  # the emulator is slow: all 1- and 2-grams + a seeded sample of the 3- and 4-grams
        rng = np.random.default_rng(7)
        strings = ["".join(t) for k in (1, 2) for t in itertools.product(ALPHABET, repeat=k)]
        strings += ["".join(rng.choice(ALPHABET, size=int(k))) for k in rng.integers(3, 5, size=1200)]
    else:
        strings = ["".join(t) for k in range(1, 5) for t in itertools.product(ALPHABET, repeat=k)]
    check(backend, pattern, strings)


@pytest.mark.parametrize("pattern", [GPT2_PATTERN, DIGITS_PATTERN], ids=["gpt2", "gpt2-digits"])
def test_random_strings(backend, pattern):
    rng = np.random.default_rng(11)
    p = np.array([6, 2, 2, 2, 3, 2, 1, 3, 8, 1, 1, 0.5, 3, 2, 1, 1, 0.5, 0.5, 0.5, 0.5])
    p = p / p.sum()
    n = 300 if backend.name == "emu" else 6000
    strings = ["".join(rng.choice(ALPHABET, size=int(rng.integers(1, 60)), p=p)) for _ in range(n)]
    strings += ["".join(rng.choice(ALPHABET, size=int(rng.integers(400, 1500)), p=p)) for _ in range(12)]
    check(backend, pattern, strings)


def test_chunk_boundaries(backend):
    """Pieces and special sequences placed right at the chunk seams (the scan window is 768 bytes; 512 is where the
    packed-byte scanner goes from 8 to 12 bytes per lane), long single-class runs."""
    strings = []
    for k in list(range(500, 530, 3)) + list(range(742, 772)):
        strings += ["a" * k + "'s b", "x" * k + "  y", " " * k + "z", "a" * k + " 'll", "é" * (k // 2) + "'t元",
                    "1" * k + "a" * 600, ("ab " * 200)[:k] + "\t\t" + "c" * 20, "a" * k + "\n\n" + "b" * k + " "]
    check(backend, GPT2_PATTERN, strings)


@pytest.mark.parametrize("pattern", [GPT2_PATTERN, DIGITS_PATTERN], ids=["gpt2", "gpt2-digits"])
def test_every_ascii_byte(backend, pattern):
    """The scanners classify ASCII arithmetically (split_device.hpp ascii_class): every byte 1..127 in letter, digit,
    space and string-edge contexts."""
    strings = []
    for c in range(1, 128):
        ch = chr(c)
        strings += [ch, "a" + ch + "a", " " + ch + " ", "1" + ch + ch + "1", ch + " x", "x " + ch, "'" + ch + "b", "x'" + ch]
    check(backend, pattern, strings)


# ------------------------------------------------------------------ BERT patterns (tokenizer_pipeline.py:392-426)
BERT_WS = r"\s+"
BERT_PUNCT = "|".join([r"[!-/]", r"[:-@]", r"[\[-`]", r"[{-~]", r"[\p{P}]", r"[\x{4E00}-\x{9FFF}]", r"[\x{3400}-\x{4DBF}]",
                       r"[\x{20000}-\x{2A6DF}]", r"[\x{2A700}-\x{2B73F}]", r"[\x{2B740}-\x{2B81F}]",
                       r"[\x{2B820}-\x{2CEAF}]", r"[\x{F900}-\x{FAFF}]", r"[\x{2F800}-\x{2FA1F}]"])
BERT_ALPHABET = ["a", "b", "1", " ", "\t", "\n", "\u00a0", "\u3000", "!", "/", ":", "@", "[", "`", "{", "~", "_", "$", "é", "元",
                 "㐀", "\U00020000", "豈", "，", "«", "😀", "٣"]


@pytest.mark.parametrize("pattern,behaviour,invert", [(BERT_WS, "remove", False), (BERT_WS, "isolate", False),
                                                      (BERT_WS, "remove", True), (BERT_PUNCT, "isolate", False),
                                                      (BERT_PUNCT, "remove", False), (BERT_PUNCT, "remove", True)])
def test_bert_patterns(backend, pattern, behaviour, invert):
    rng = np.random.default_rng(23)
    strings = ["".join(t) for k in (1, 2) for t in itertools.product(BERT_ALPHABET, repeat=k)]
    n = 300 if backend.name == "emu" else 5000
    strings += ["".join(rng.choice(BERT_ALPHABET, size=int(rng.integers(1, 50)))) for _ in range(n)]
    strings += ["".join(rng.choice(BERT_ALPHABET, size=int(rng.integers(450, 1400)))) for _ in range(8)]
    strings += ["", " ", "   ", "a" * 600, " " * 700 + "x", "!" * 520, "ab " * 400]
    check(backend, pattern, strings, behaviour, invert)


@pytest.mark.parametrize("pattern,behaviour,invert", [(BERT_WS, "remove", False), (BERT_WS, "isolate", False),
                                                      (BERT_PUNCT, "isolate", False), (BERT_PUNCT, "remove", True)])
def test_bert_patterns_ascii_windows(backend, pattern, behaviour, invert):
    """ASCII-only strings take the packed-byte scanner of the class patterns (class_packed_starts: 4, 8 or 12 bytes per
    lane by window length): every byte 1..127 in context, and lengths around the 256 / 512 / 768-byte steps."""
    rng = np.random.default_rng(29)
    strings = []
    for c in range(1, 128):
        ch = chr(c)
        strings += [ch, "a" + ch + "a", " " + ch + " ", ch + ch, "x " + ch, ch + " x"]
    ascii_alphabet = ["a", "b", "1", " ", " ", "\t", "\n", "!", "/", ":", "@", "[", "`", "{", "~", "_", "$", "-", "'"]
    lengths = [3, 60, 250, 255, 256, 257, 300, 500, 511, 512, 513, 600, 760, 767, 768, 769, 800, 1500, 2100]
    for n in lengths[: 9 if backend.name == "emu" else None] + lengths[-4:]:
        strings += ["".join(rng.choice(ascii_alphabet, size=n)) for _ in range(2)]
    strings += ["ab " * 90, "!" * 300, " " * 255 + "x", "a" * 256 + "!", "word, " * 100]
    check(backend, pattern, strings, behaviour, invert)


def test_bert_patterns_max_splits(backend):
    strings = ["one two  three four", "a,b,c,d", "x", "  lead and trail  ", ",,,"]
    for ms in (1, 2, 5):
        check(backend, BERT_WS, strings, "remove", False, ms)
        check(backend, BERT_PUNCT, strings, "isolate", False, ms)


def test_bert_chain_matches_reference_kat(backend):
    """tests/layer_tests.py:340 ("Hello     world!" -> bert whitespace) and the chained delimiter split on the pieces."""
    inputs = one_string_per_row(["Hello     world!", "don't stop-me,now", "  多语言 text。"])
    ws = np.frombuffer(BERT_WS.encode(), np.uint8)
    pu = np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    ref1 = O.RegexSplit(BERT_WS, "remove")(*inputs)
    got1 = RegexSplit("remove", lib=backend.lib).evaluate(backend.data(inputs) + [ws])
    assert_same(ref1[:4], got1[:4], backend.host, "bert whitespace")
    assert O.unpack_strings(ref1[2], ref1[3], ref1[4])[:2] == [b"Hello", b"world!"]
    ref2 = O.RegexSplit(BERT_PUNCT, "isolate")(*ref1[:5])
    got2 = RegexSplit("isolate", lib=backend.lib).evaluate(list(got1[:5]) + [pu])
    assert_same(ref2[:4], got2[:4], backend.host, "bert delimiters on the whitespace pieces")


# ------------------------------------------------------------------ Llama-3 pattern (sequential matcher, lane per row)
LLAMA3_ALPHABET = ["a", "s", "S", "t", "r", "E", "l", "L", "v", "m", "d", "ſ", "1", "٣", " ", "\t", "\n", "\r", "\u00a0", "'", "!",
                   "é", "元", "😀"]


def test_llama3_exhaustive_short_strings(backend):
    rng = np.random.default_rng(5)
    if backend.name == "emu":
        strings = ["".join(t) for k in (1, 2) for t in itertools.product(LLAMA3_ALPHABET, repeat=k)]
        strings += ["".join(rng.choice(LLAMA3_ALPHABET, size=int(k))) for k in rng.integers(3, 6, size=1500)]
    else:
        strings = ["".join(t) for k in range(1, 4) for t in itertools.product(LLAMA3_ALPHABET, repeat=k)]
        strings += ["".join(rng.choice(LLAMA3_ALPHABET, size=int(k))) for k in rng.integers(4, 7, size=60000)]
    check(backend, LLAMA3_PATTERN, strings)


def test_llama3_random_and_long(backend):
    rng = np.random.default_rng(17)
    p = np.array([6, 2, 1, 2, 2, 1, 2, 1, 1, 1, 1, 0.3, 4, 1, 8, 1, 1.5, 0.7, 0.5, 2.5, 2, 1, 1, 0.5])
    p = p / p.sum()
    n = 300 if backend.name == "emu" else 6000
    strings = ["".join(rng.choice(LLAMA3_ALPHABET, size=int(rng.integers(1, 80)), p=p)) for _ in range(n)]
    strings += ["".join(rng.choice(LLAMA3_ALPHABET, size=int(rng.integers(400, 1500)), p=p)) for _ in range(10)]
    strings += ["", "1" * 700, " " * 300 + "x", "\n" * 20 + " a", "it's IT'S 'Tis don'T we'LL 12345 6,789.10\r\n\r\n  end  "]
    check(backend, LLAMA3_PATTERN, strings)
    check(backend, LLAMA3_PATTERN, strings[:50], max_splits=3)


def test_llama3_bitparallel_paths(backend):
    """Strings without the chars that send a window to the literal matcher (non-ASCII digits, U+017F): the bit-parallel
    rules and their ripples across 64-byte words -- digit groups at every phase and offset, white-space runs with line
    breaks at every place, line breaks behind an O char, contractions, multi-byte chars next to all of them, chunk seams."""
    rng = np.random.default_rng(23)
    alpha = [a for a in LLAMA3_ALPHABET if a not in ("ſ", "٣")] + ["é", "元", "😀", "\u3000", ".", "-"]
    n = 150 if backend.name == "emu" else 20000
    strings = ["".join(rng.choice(alpha, size=int(rng.integers(1, 40)))) for _ in range(n)]
    strings += ["".join(rng.choice(alpha, size=int(rng.integers(300, 1600)))) for _ in range(6 if backend.name == "emu" else 300)]
    for pre in (0, 1, 2, 61, 62, 63, 64, 65, 127, 500, 509, 510, 511, 512):       # digit runs / white space across word and chunk borders
        for k in (1, 2, 3, 4, 7, 130):
            strings.append("x" * pre + "1" * k + "y")
            strings.append("x" * pre + " " * k + "y")
            strings.append("x" * pre + "!" + "\n" * k + " " * (k % 3) + "y")
            strings.append("x" * pre + " " * k + "\n" + " " * k + "y")
            strings.append("x" * pre + "é" * k + "'LL" + "元" * k + " " + "😀" * k + "\r\n")
    strings += [" " * 2000, "\n" * 1200 + "a", "a" + " \n" * 400, "1" * 1300 + " " + "2" * 5, "!" * 900 + "\n" * 700 + "?", "x" * 3000 + "'s"]
    check(backend, LLAMA3_PATTERN, strings)


def test_llama3_rows_with_several_strings_and_skips(backend):
    strings = [b"Hello world's 1234", b"<|begin_of_text|>", b"  two  spaces\n\nnew", b"", b"x"]
    b, e, c = O.pack_strings(strings)
    rb, re_ = np.array([0, 2, 2], np.int32), np.array([2, 2, 5], np.int32)
    skips = np.array([0, 1, 0, 0, 0], np.uint8)
    pat = np.frombuffer(LLAMA3_PATTERN.encode(), np.uint8)
    ref = O.RegexSplit(LLAMA3_PATTERN, "isolate")(rb, re_, b, e, c, skips=skips)
    got = RegexSplit("isolate", lib=backend.lib).evaluate(backend.data([rb, re_, b, e, c, skips]) + [pat])
    assert_same(ref[:4] + [ref[5]], list(got[:4]) + [got[5]], backend.host, "llama3 ragged rows + skips")


# ------------------------------------------------------------------ the reference's own RegexSplit known answers on the device
from tests.golden.reference_kats import REGEX_SPLIT_KATS  # noqa: E402


@pytest.mark.parametrize("text, expected, layer", REGEX_SPLIT_KATS)
def test_reference_regex_split_kats(backend, text, expected, layer):
    """tests/layer_tests.py:331-389 through ovtk_regex_split_run: the unpacked pieces equal the reference's expected
    strings (the reference test packs outputs 2..4 with StringTensorPack and compares)."""
    pattern, behaviour, invert = layer
    from openvino_tokenizers_amd import _lib as L
    split = RegexSplit(behaviour, invert, lib=backend.lib)
    try:
        out = split.evaluate(backend.data(one_string_per_row([text])) + [np.frombuffer(pattern.encode(), np.uint8)])
    except L.OvtkError as err:
        if err.code == L.E_UNSUPPORTED:
            pytest.skip("pattern / behaviour has no device matcher yet")
        raise
    got = tuple(s.decode("utf-8") for s in O.unpack_strings(*[backend.host(x) for x in out[2:5]]))
    assert got == expected


@pytest.mark.parametrize("pattern,behaviour", [(GPT2_PATTERN, "isolate"), (r"\s+", "remove")])
def test_legacy_nine_input_form(backend, pattern, behaviour):
    """RegexSplit of old IRs (regex_split.cpp:102,164-179,235-238): inputs 6-8 hold "skip tokens"; a string that EQUALS one
    passes through unsplit, everything else is split; five outputs.  Expected value: the oracle given the membership flags as
    its skips input (the same pass-through branch, :231-238)."""
    from tests.util import pack_strings
    rows = [["<s>", "hello world", "</s>"], ["<s> x", "", "<mask>"], ["</s>", "</s> ", "don't <s>"], []]
    skip_tokens = ["<s>", "</s>", "<mask>", "<s>"]
    flat = [s for r in rows for s in r]
    b, e, c = pack_strings(flat)
    rends = np.cumsum([len(r) for r in rows]).astype(np.int32)
    rbeg = (rends - np.asarray([len(r) for r in rows], np.int32)).astype(np.int32)
    pat = np.frombuffer(pattern.encode(), np.uint8)
    flags = np.asarray([s in set(skip_tokens) for s in flat])
    ref = O.RegexSplit(pattern, behaviour)(rbeg, rends, b, e, c, skips=flags)
    got = RegexSplit(behaviour, lib=backend.lib).evaluate(backend.data([rbeg, rends, b, e, c]) + [pat] + list(pack_strings(skip_tokens)))
    assert len(got) == 5
    assert_same(list(ref[:4]), got[:4], backend.host, "RegexSplit, 9 inputs")
    # no skip tokens: plain six-input behaviour
    ref0 = O.RegexSplit(pattern, behaviour)(rbeg, rends, b, e, c)
    got0 = RegexSplit(behaviour, lib=backend.lib).evaluate(backend.data([rbeg, rends, b, e, c]) + [pat] + list(pack_strings([])))
    assert_same(list(ref0[:4]), got0[:4], backend.host, "RegexSplit, 9 inputs, empty skip set")
