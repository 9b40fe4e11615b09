"""TrieTokenizer (src/trie_tokenizer.cpp, RWKV world tokenizer): greedy longest match; kernel vs oracle, and the oracle
vs a ten-line pure-Python statement of the same rule."""
import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import TrieTokenizer
from oracle import oracle as O
from tests.util import assert_same


def rwkv_like_vocab(rng, n_extra=3000):
    """All 256 single bytes (ids 1..256, like the RWKV vocabulary file) + random multi-byte entries, some of them
    prefixes of others, one duplicate string (the later id wins, trie_tokenizer.cpp:40-43)."""
    vocab = [bytes([b]) for b in range(256)]
    alphabet = list(b"abcdeft \n") + [0xE4, 0xB8, 0xAD, 0xE6, 0x96, 0x87]
    seen = set(vocab)
    while len(vocab) < 256 + n_extra:
        w = bytes(rng.choice(alphabet, size=int(rng.integers(2, 9))).tolist())
        if w not in seen:
            seen.add(w)
            vocab.append(w)
    vocab.append(vocab[300])
    indices = np.arange(1, len(vocab) + 1, dtype=np.int32)
    return vocab, indices


def py_greedy(vocab, indices, s):
    table = {}
    for w, i in zip(vocab, indices.tolist()):
        table[w] = i
    longest = max(map(len, table))
    out, i = [], 0
    while i < len(s):
        for ln in range(min(longest, len(s) - i), 0, -1):
            if s[i:i + ln] in table:
                out.append(table[s[i:i + ln]])
                i += ln
                break
        else:
            raise ValueError("no match")
    return out


def test_oracle_against_plain_python():
    rng = np.random.default_rng(1)
    vocab, indices = rwkv_like_vocab(rng, 500)
    strings = [bytes(rng.choice(list(b"abcdeft \n\xe4\xb8\xad\xe6\x96\x87xyz"), size=int(rng.integers(0, 60))).tolist()) for _ in range(200)]
    b, e, c = O.pack_strings(strings)
    rb = np.arange(len(strings), dtype=np.int32)
    ob, oe, ids = O.TrieTokenizer(vocab, indices)(rb, rb + 1, b, e, c)
    for i, s in enumerate(strings):
        assert ids[ob[i]:oe[i]].tolist() == py_greedy(vocab, indices, s)


def test_kernel_matches_oracle(backend):
    rng = np.random.default_rng(2)
    vocab, indices = rwkv_like_vocab(rng)
    n = 120 if backend.name == "emu" else 20000
    strings = [bytes(rng.choice(list(b"abcdeft \n\xe4\xb8\xad\xe6\x96\x87xyz\x00\xff"), size=int(rng.integers(0, 90))).tolist()) for _ in range(n)]
    strings[3] = b""
    b, e, c = O.pack_strings(strings)
    # ragged rows: 0, 1 or 2 strings per row
    cuts = np.unique(np.concatenate([[0, len(strings)], rng.integers(0, len(strings), len(strings) // 2)])).astype(np.int32)
    rb, re_ = cuts[:-1], cuts[1:]
    rb, re_ = np.concatenate([rb, [5]]).astype(np.int32), np.concatenate([re_, [5]]).astype(np.int32)   # + an empty row
    ref = O.TrieTokenizer(vocab, indices)(rb, re_, b, e, c)
    vb, ve, vc = O.pack_strings(vocab)
    got = TrieTokenizer(lib=backend.lib).evaluate(backend.data([rb, re_, b, e, c]) + [vb, ve, vc, indices])
    assert_same(list(ref), got, backend.host, "TrieTokenizer")


def test_errors(backend):
    vb, ve, vc = O.pack_strings([b"a", b"ab"])
    idx = np.array([7, 9], np.int32)
    b, e, c = O.pack_strings([b"abz"])
    one = np.array([0], np.int32)
    with pytest.raises(L.OvtkError) as ei:   # 'z' matches nothing: the reference would never return
        TrieTokenizer(lib=backend.lib).evaluate([one, one + 1, b, e, c, vb, ve, vc, idx])
    assert ei.value.code == L.E_VOCAB
    with pytest.raises(O.OracleError):
        O.TrieTokenizer([b"a", b"ab"], idx)(one, one + 1, b, e, c)
    with pytest.raises(L.OvtkError, match="Vocab size must be equal to Indices size"):
        TrieTokenizer(lib=backend.lib).evaluate([one, one + 1, b, e, c, vb, ve, vc, idx[:1]])
    ok = TrieTokenizer(lib=backend.lib).evaluate([one, one + 1, b, np.array([2], np.int32), c, vb, ve, vc, idx])
    assert backend.host(ok[2]).tolist() == [9]


def test_rows_that_share_strings(backend):
    """Rows may name the same strings again (the reference only dereferences offsets): the rows' bytes then outnumber the chars tensor's and
    the staging buffer of the one-walk form is grown for a second attempt; window edges of the 16-byte text loads at every offset."""
    rng = np.random.default_rng(4)
    vocab, indices = rwkv_like_vocab(rng, 800)
    words = vocab[256:]
    strings = [b"".join(words[int(k)] for k in rng.integers(0, len(words), int(rng.integers(1, 14)))) for _ in range(60)] + [b"a" * k for k in range(1, 20)]
    b, e, c = O.pack_strings(strings)
    n = len(strings)
    # every string a row, a third of them once more, and one row of many strings: the rows' bytes exceed the chars tensor's (the staging
    # buffer's first size), the ids do not (the reference sizes its output by the chars tensor, trie_tokenizer.cpp:57-59)
    rb = np.concatenate([np.arange(n), np.arange(n // 3), [n // 2]]).astype(np.int32)
    re_ = np.concatenate([np.arange(n) + 1, np.arange(n // 3) + 1, [n // 2 + 9]]).astype(np.int32)
    ref = O.TrieTokenizer(vocab, indices)(rb, re_, b, e, c)
    assert int((e - b)[rb[n:n + n // 3]].sum()) > 0 and len(ref[2]) <= len(c)
    _run_shared(backend, vocab, indices, rb, re_, b, e, c, ref)
    # round 6: a row's stretch is its bytes in whole 64-byte segments and the first size has room for that -- three rows that each name
    # EVERY string, and a few of one string, outgrow it
    rb = np.array([0, 0, 0, 3, 4, 50], np.int32)
    re_ = np.array([n, n, n, 4, 5, 51], np.int32)
    ref = O.TrieTokenizer(vocab, indices)(rb, re_, b, e, c)
    assert 3 * len(c) > len(c) + 64 * (len(rb) + 1) and len(ref[2]) <= len(c)
    _run_shared(backend, vocab, indices, rb, re_, b, e, c, ref)


def _run_shared(backend, vocab, indices, rb, re_, b, e, c, ref):
    vb, ve, vc = O.pack_strings(vocab)
    op = TrieTokenizer(lib=backend.lib)
    for call in range(2):
        got = op.evaluate(backend.data([rb, re_, b, e, c]) + [vb, ve, vc, indices])
        assert_same(list(ref), got, backend.host, f"shared strings, call {call}")


def test_rows_of_many_segments(backend):
    """The segmented walk (round 6: a lane per 64-byte segment walks the chain that starts at the segment's first byte, a lane per row stitches
    them): tokens across segment borders, a token longer than a segment (and than two), chains that never meet the guess ({a, aaa} on a run of
    a's: 64 is not a multiple of 3), a guess that starts on a byte no token starts with (`z` only ever inside `az`: the guess breaks there,
    the true chain never starts a token there -- no error), every length around the segment size, rows of several strings among them."""
    rng = np.random.default_rng(5)
    vocab = [bytes([b]) for b in range(256) if b != ord("z")] + [b"aaa", b"az", b"ab" * 40, b"c" * 150, b"hello", b" world", b" wor", b"ld", b"lo w", b"the quick", b" brown fox"]
    vocab += [bytes(rng.choice(list(b"abc dehlorw"), size=int(rng.integers(2, 12))).tolist()) for _ in range(400)]
    vocab = list(dict.fromkeys(vocab))
    indices = np.arange(10, 10 + len(vocab), dtype=np.int32)
    strings = [b"a" * k for k in (1, 2, 3, 63, 64, 65, 127, 128, 129, 191, 200, 700)]
    strings += [b"x" * k + b"az" * 5 + b"x" * 70 for k in range(60, 68)]   # `z` on either side of a border
    strings += [b"x" * k + b"ab" * 45 + b"c" * 320 + b"hello world" * 9 for k in range(0, 70, 7)]
    words = [b"hello", b" world", b"the quick", b" brown fox", b"aaa", b" ", b"lo w", b"c" * 150, b"ab" * 40] + vocab[-60:]
    n = 40 if backend.name == "emu" else 6000
    strings += [b"".join(words[int(k)] for k in rng.integers(0, len(words), int(rng.integers(1, 120)))) for _ in range(n)]
    strings += [bytes(rng.choice(list(b"abc dehlorw"), size=int(k)).tolist()) for k in rng.integers(0, 600, n)]
    strings.append(b"")
    b, e, c = O.pack_strings(strings)
    m = len(strings)
    cuts = np.unique(np.concatenate([[0, m], np.arange(0, 30), rng.integers(0, m, m - m // 6)])).astype(np.int32)   # mostly one string per row, some 2-4
    rb, re_ = cuts[:-1], cuts[1:]
    ref = O.TrieTokenizer(vocab, indices)(rb, re_, b, e, c)
    vb, ve, vc = O.pack_strings(vocab)
    op = TrieTokenizer(lib=backend.lib)
    got = op.evaluate(backend.data([rb, re_, b, e, c]) + [vb, ve, vc, indices])
    assert_same(list(ref), got, backend.host, "rows of many segments")
    # ... and where the TRUE chain starts a token on `z`, the error is raised wherever the segment borders fall
    for k in (0, 63, 64, 65, 200):
        bb, ee, cc = O.pack_strings([b"a" * 300, b"b" * k + b"z" + b"a" * 100])
        with pytest.raises(L.OvtkError) as ei:
            op.evaluate(backend.data([np.array([0, 1], np.int32), np.array([1, 2], np.int32), bb, ee, cc]) + [vb, ve, vc, indices])
        assert ei.value.code == L.E_VOCAB
