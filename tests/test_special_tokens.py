"""SpecialTokensSplit (src/special_tokens_split.cpp): the reference's known answers through the kernel, and the kernel
against the oracle (PCRE2 with the generated pattern) on token sets that exercise strip_left / strip_right,
alternation order, whitespace back-off and the chain SpecialTokensSplit -> RegexSplit -> BPETokenizer."""
import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import BPETokenizer, RegexSplit, SpecialTokensSplit
from oracle import oracle as O
from tests.golden.reference_kats import SPECIAL_TOKENS_KATS
from tests.util import BpeTok, assert_same, one_string_per_row


def u8(s):
    return np.frombuffer(s.encode("utf-8"), np.uint8)


@pytest.mark.parametrize("tokens, text, expected, expected_skips", SPECIAL_TOKENS_KATS)
def test_reference_kats(backend, tokens, text, expected, expected_skips):
    """tests/layer_tests.py:405-457."""
    pat = O.special_tokens_pattern(tokens)
    got = SpecialTokensSplit(lib=backend.lib).evaluate(backend.data(one_string_per_row([text])) + [u8(pat)])
    pieces = tuple(s.decode("utf-8") for s in O.unpack_strings(backend.host(got[2]), backend.host(got[3]), backend.host(got[4])))
    assert pieces == expected and backend.host(got[5]).tolist() == expected_skips


TOKEN_SETS = [
    [("<|endoftext|>", False, False)],
    [("<s>", False, False), ("</s>", False, True), ("<pad>", True, False), ("[MASK]", True, True), ("<s", False, False)],
    [("ab", False, False), ("abc", False, False), ("b", True, True)],           # alternation order: first listed wins
    [(" x", True, False), ("x", False, False), ("\u00a0y", False, True)],        # tokens that start with whitespace
    [("<｜begin▁of▁sentence｜>", False, False), ("▁", True, False), ("。", False, True)],
]


@pytest.mark.parametrize("tokens", TOKEN_SETS)
def test_random_texts(backend, tokens):
    rng = np.random.default_rng(len(tokens) * 7 + len(tokens[0][0]))
    alphabet = [t for t, _, _ in tokens] * 3 + [" ", " ", "\t", "\n", "\u00a0", "\u3000", "a", "b", "c", "x", "y", "<", ">", "s", "|",
                                               "é", "元", "▁", "<s", "</", "[MASK"]
    n = 200 if backend.name == "emu" else 4000
    strings = ["".join(rng.choice(alphabet, size=int(rng.integers(0, 24)))) for _ in range(n)]
    strings += ["", " ", "   ", " " * 300 + tokens[0][0] + " " * 300, "a" * 1000 + tokens[-1][0]]
    inputs = one_string_per_row(strings)
    pat = O.special_tokens_pattern(tokens)
    ref = O.SpecialTokensSplit(pat)(*inputs)
    got = SpecialTokensSplit(lib=backend.lib).evaluate(backend.data(inputs) + [u8(pat)])
    assert_same(ref[:4] + [ref[5]], list(got[:4]) + [got[5]], backend.host, "SpecialTokensSplit")


def test_skips_input_and_ragged_rows(backend):
    tokens = [("<|endoftext|>", False, False), ("<pad>", True, True)]
    pat = O.special_tokens_pattern(tokens)
    strings = [b"hello <|endoftext|> world", b"<pad>", b"", b"  <pad>  x<|endoftext|>", b"keep <|endoftext|> whole"]
    b, e, c = O.pack_strings(strings)
    rb, re_ = np.array([0, 2, 2], np.int32), np.array([2, 2, 5], np.int32)
    skips = np.array([0, 0, 0, 0, 1], np.uint8)
    ref = O.SpecialTokensSplit(pat)(rb, re_, b, e, c, skips=skips)
    got = SpecialTokensSplit(lib=backend.lib).evaluate(backend.data([rb, re_, b, e, c, skips]) + [u8(pat)])
    assert_same(ref[:4] + [ref[5]], list(got[:4]) + [got[5]], backend.host, "7-input form")


def test_chain_with_regex_split_and_bpe(backend):
    """SpecialTokensSplit -> RegexSplit(skips) -> BPETokenizer: the special token reaches BPE whole and maps to its id."""
    tok = BpeTok.load("gpt2_small")
    pat = O.special_tokens_pattern([("<|endoftext|>", False, False)])
    inputs = one_string_per_row(["some text<|endoftext|> more text here<|endoftext|>", "no specials", "<|endoftext|>"])
    s_ref = O.SpecialTokensSplit(pat)(*inputs)
    r_ref = O.RegexSplit(tok.pattern, "isolate")(*s_ref[:5], skips=s_ref[5])
    ref = tok.oracle()(*r_ref[:5])
    s_got = SpecialTokensSplit(lib=backend.lib).evaluate(backend.data(inputs) + [u8(pat)])
    r_got = RegexSplit("isolate", lib=backend.lib).evaluate(list(s_got[:6]) + [tok.pattern_u8()])
    got = BPETokenizer(**tok.attrs, lib=backend.lib).evaluate(list(r_got[:5]) + tok.consts)
    assert_same(ref, got, backend.host, "special -> split -> bpe")
    assert (ref[2] == tok.added[b"<|endoftext|>"]).sum() == 3


def test_unsupported_patterns(backend):
    for pat in ("[a-z]+", "(a|b)+", "(\\d)", "(?:\\s*)(", ""):
        with pytest.raises(L.OvtkError) as ei:
            SpecialTokensSplit(lib=backend.lib).evaluate(backend.data(one_string_per_row(["x"])) + [u8(pat) if pat else np.zeros(0, np.uint8)])
        assert ei.value.code == L.E_UNSUPPORTED


def _texts_with_specials(rng, n, every=7):
    """Text rows of the zipf model, one row in `every` with special tokens inside (also first / last, and two in a row)."""
    from tools.workloads import TextModel
    b, e, c = TextModel(int(rng.integers(1 << 30)), "zipf").batch(n, 200)
    rows = [bytes(c[b[i]:e[i]]) for i in range(n)]
    sp = b"<|endoftext|>"
    for i in range(0, n, every):
        k = i // every % 5
        r = rows[i]
        cut = int(rng.integers(0, len(r) + 1))
        rows[i] = [r[:cut] + sp + r[cut:], sp + r, r + sp, r[:cut] + sp + sp + r[cut:], sp][k]
    rows[1] = b"a < b <| c <|endoftext| d <|endoftext|"   # the token's first bytes without the token
    return rows


@pytest.mark.parametrize("n_rows", [40, 320])
def test_fused_special_split_bpe(backend, n_rows):
    """ovtk_encode_special_run (SpecialTokensSplit -> RegexSplit -> BPETokenizer in one call, tokenizer_pipeline.py:1613-1636) = the
    oracle chain = the three ops one after the other; small batches (one launch) and batches that take the span kernel; blocking and
    in two halves (device tensors)."""
    from openvino_tokenizers_amd.ops import FusedSpecialSplitBPE
    tok = BpeTok.load("gpt2_small")
    pat = O.special_tokens_pattern([("<|endoftext|>", False, False)])
    rng = np.random.default_rng(n_rows)
    inputs = one_string_per_row(_texts_with_specials(rng, n_rows))
    s_ref = O.SpecialTokensSplit(pat)(*inputs)
    r_ref = O.RegexSplit(tok.pattern, "isolate")(*s_ref[:5], skips=s_ref[5])
    ref = tok.oracle()(*r_ref[:5])
    assert (ref[2] == tok.added[b"<|endoftext|>"]).sum() >= n_rows // 7
    fused = FusedSpecialSplitBPE(SpecialTokensSplit(lib=backend.lib), RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    data = backend.data(inputs)
    for call in range(2):
        assert_same(ref, fused.evaluate(data + [u8(pat)], tok.pattern_u8(), tok.consts), backend.host, f"fused special -> split -> bpe, call {call}")
    if backend.name == "hip-device":
        assert_same(ref, fused.enqueue(data + [u8(pat)], tok.pattern_u8(), tok.consts)(), backend.host, "the same in two halves")
    # strip_left / strip_right tokens, and a skips input (7-input form)
    pat2 = O.special_tokens_pattern([("<|endoftext|>", False, False), ("<pad>", True, True)])
    strings = [b"hello <|endoftext|> world", b"<pad>", b"", b"  <pad>  x<|endoftext|>", b"keep <|endoftext|> whole"] * (n_rows // 5)
    inputs2 = one_string_per_row(strings)
    skips = (np.arange(len(strings)) % 5 == 4).astype(np.uint8)
    s_ref = O.SpecialTokensSplit(pat2)(*inputs2, skips=skips)
    ref2 = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(*s_ref[:5], skips=s_ref[5])[:5])
    fused2 = FusedSpecialSplitBPE(SpecialTokensSplit(lib=backend.lib), RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    assert_same(ref2, fused2.evaluate(backend.data(inputs2 + [skips]) + [u8(pat2)], tok.pattern_u8(), tok.consts), backend.host, "strip flags + skips input")
    # rows of several strings, an empty row, strings that are not one stretch of text (the wave's sweep does not apply)
    strings3 = [b"hello <|endoftext|> world", b"<pad>", b"", b"  <pad>  x<|endoftext|>", b"keep <|endoftext|> whole", b"tail"] * 50
    b3, e3, c3 = O.pack_strings(strings3)
    b3, e3 = b3[::-1].copy(), e3[::-1].copy()   # (reversed: string i + 1 lies in front of string i)
    rb3 = np.arange(0, 300, 3, dtype=np.int32)
    re3 = rb3 + np.asarray([3, 0, 2, 3] * 25, np.int32)
    s_ref = O.SpecialTokensSplit(pat2)(rb3, re3, b3, e3, c3)
    ref3 = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(*s_ref[:5], skips=s_ref[5])[:5])
    assert_same(ref3, fused2.evaluate(backend.data([rb3, re3, b3, e3, c3]) + [u8(pat2)], tok.pattern_u8(), tok.consts), backend.host, "ragged rows, scattered strings")
