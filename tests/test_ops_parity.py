"""WordpieceTokenizer / VocabEncoder / RaggedToDense / VocabDecoder / ByteFallback / FuzeRagged / fused detokenizer:
the oracle against HF golden vectors and the reference's known answers, and the kernels against the oracle
(`backend` = SIMT emulator on CPU, libovtk_amd.so with host buffers, with device buffers).  Bit-exact throughout."""
import json

import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import (ByteFallback, FusedDetokenizer, FusedSplitWordpiece, FuzeRagged, RaggedToDense,
                                         RegexSplit, VocabDecoder, VocabEncoder, WordpieceTokenizer)
from oracle import oracle as O
from tests.golden.reference_kats import RAGGED_TO_DENSE_KATS
from tests.util import assert_same, one_string_per_row, pack_strings
from tools.make_tokenizers import load_tokenizer
from tools.workloads import TextModel, ragged_rows

GOLDEN = __import__("pathlib").Path(__file__).parent / "golden"
BERT_WS = r"\s+"
BERT_PUNCT = "|".join([r"[!-/]", r"[:-@]", r"[\[-`]", r"[{-~]", r"[\p{P}]", r"[\x{4E00}-\x{9FFF}]", r"[\x{3400}-\x{4DBF}]",
                       r"[\x{20000}-\x{2A6DF}]", r"[\x{2A700}-\x{2B73F}]", r"[\x{2B740}-\x{2B81F}]",
                       r"[\x{2B820}-\x{2CEAF}]", r"[\x{F900}-\x{FAFF}]", r"[\x{2F800}-\x{2FA1F}]"])


def bert_words(inputs):
    """The two chained RegexSplit ops of the BERT pipeline (tokenizer_pipeline.py:392-435), on the oracle."""
    s1 = O.RegexSplit(BERT_WS, "remove")(*inputs)
    return O.RegexSplit(BERT_PUNCT, "isolate")(*s1[:5])[:5]


def wp_consts(tok):
    return list(pack_strings(tok["vocab"])) + [np.asarray(tok["unk_id"], np.int32)]


# ------------------------------------------------------------------ oracle pinned by HF (CPU)
def test_oracle_wordpiece_matches_hf_golden():
    z = np.load(GOLDEN / "golden_wordpiece_bert_small.npz")
    tok = load_tokenizer("bert_small")
    rb, re_ = ragged_rows(len(z["begins"]))
    words = bert_words([rb, re_, z["begins"], z["ends"], z["chars"]])
    ob, oe, ids = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*words, tok["unk_id"])
    assert np.array_equal(ob, z["id_begins"]) and np.array_equal(oe, z["id_ends"]) and np.array_equal(ids, z["ids"])


def test_oracle_wordpiece_matches_hf_live():
    tokenizers = pytest.importorskip("tokenizers")
    hf = tokenizers.Tokenizer.from_file(str(GOLDEN / "tok_bert_small.hf.json"))
    tok = load_tokenizer("bert_small")
    wp = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])
    b, e, c = TextModel(77, "zipf").batch(200, 180)
    c = np.frombuffer(c.tobytes().lower(), np.uint8)
    rb, re_ = ragged_rows(len(b))
    ob, oe, ids = wp(*bert_words([rb, re_, b, e, c]), tok["unk_id"])
    raw = c.tobytes()
    for i in range(len(b)):
        assert hf.encode(raw[b[i]:e[i]].decode(), add_special_tokens=False).ids == ids[ob[i]:oe[i]].tolist()


# ------------------------------------------------------------------ WordpieceTokenizer kernel
@pytest.mark.parametrize("name,n,target", [("bert_small", 40, 200), ("bert", 16, 256)])
def test_wordpiece(backend, name, n, target):
    tok = load_tokenizer(name)
    b, e, c = TextModel(21, "zipf").batch(n, target)
    c = np.frombuffer(c.tobytes().lower(), np.uint8)
    rb, re_ = ragged_rows(n)
    words = bert_words([rb, re_, b, e, c])
    ref = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*words, tok["unk_id"])
    op = WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=backend.lib)
    got = op.evaluate(backend.data(words) + wp_consts(tok))
    assert_same(ref, got, backend.host, "WordpieceTokenizer")
    assert (ref[2] == tok["unk_id"]).sum() < len(ref[2]) // 2


@pytest.mark.parametrize("name,kind,n,target", [("bert_small", "zipf", 40, 200), ("bert_small", "mixed", 24, 300),
                                                 ("bert", "zipf", 12, 256)])
def test_fused_split_wordpiece(backend, name, kind, n, target):
    """The three-op BERT chain on the device (two RegexSplit scanners + WordpieceTokenizer) and the fused path, both
    against the oracle chain (PCRE2 splits + trie WordPiece)."""
    tok = load_tokenizer(name)
    b, e, c = TextModel(31, kind).batch(n, target)
    c = np.frombuffer(c.tobytes().decode().lower().encode(), np.uint8) if kind == "zipf" else c
    if kind == "mixed":  # lower() may change byte lengths of non-ASCII text: keep the original bytes
        pass
    rb, re_ = ragged_rows(n)
    inputs = [rb, re_, b, e, c]
    words = bert_words(inputs)
    ref = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*words, tok["unk_id"])
    ws_pat, pu_pat = np.frombuffer(BERT_WS.encode(), np.uint8), np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    data = backend.data(inputs)
    ws = RegexSplit("remove", lib=backend.lib)
    pu = RegexSplit("isolate", lib=backend.lib)
    wp = WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=backend.lib)
    s1 = ws.evaluate(data + [ws_pat])
    s2 = pu.evaluate(list(s1[:5]) + [pu_pat])
    assert_same(words[:4], s2[:4], backend.host, "BERT split chain")
    got = wp.evaluate(list(s2[:5]) + wp_consts(tok))
    assert_same(ref, got, backend.host, "WordpieceTokenizer on device-split words")
    chain = FusedSplitWordpiece(ws, pu, wp)
    fused = chain.evaluate(data, ws_pat, pu_pat, wp_consts(tok)[0:3] + [np.asarray(tok["unk_id"], np.int32)])
    assert_same(ref, fused, backend.host, "fused split + WordPiece")
    # once more on what the word store learned, and with another unk_token_id (input 8 is read every call,
    # wordpiece_tokenizer.cpp:74: nothing that was filed may depend on it)
    fused = chain.evaluate(data, ws_pat, pu_pat, wp_consts(tok)[0:3] + [np.asarray(tok["unk_id"], np.int32)])
    assert_same(ref, fused, backend.host, "fused split + WordPiece, second call")
    ref7 = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*words, 7)
    fused = chain.evaluate(data, ws_pat, pu_pat, wp_consts(tok)[0:3] + [np.asarray(7, np.int32)])
    assert_same(ref7, fused, backend.host, "fused split + WordPiece, unk_token_id = 7")


def test_fused_split_wordpiece_edge_cases(backend):
    """Unknown / over-long words, > 15-byte words (memo misses), empty rows, the all-empty batch, > 512-byte strings."""
    vocab = [b"[UNK]", b"un", b"##aff", b"##able", b"aff", b"a", b"##b", b"able", b"unaffable", b"x" * 20, b"##" + b"x" * 20,
             b",", b"!", "é".encode(), "##é".encode(), "元".encode()]
    strings = ["unaffable, unaffablez! a ab abz", "", "   ", "x" * 20 + " " + "x" * 40 + "," + "x" * 41, "éé éz 元元 a元b",
               "a " * 400, "able" * 200, ",,,,", " lead", "trail "]
    inputs = one_string_per_row(strings)
    ws_pat, pu_pat = np.frombuffer(BERT_WS.encode(), np.uint8), np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    consts = list(pack_strings(vocab)) + [np.asarray(0, np.int32)]
    for max_bytes in (100, 40, 3):
        ref = O.WordpieceTokenizer(vocab, "##", max_bytes)(*bert_words(inputs), 0)
        fused = FusedSplitWordpiece(RegexSplit("remove", lib=backend.lib), RegexSplit("isolate", lib=backend.lib),
                                    WordpieceTokenizer("##", max_bytes, lib=backend.lib))
        assert_same(ref, fused.evaluate(backend.data(inputs), ws_pat, pu_pat, consts), backend.host, f"max_bytes={max_bytes}")
        assert_same(ref, fused.evaluate(backend.data(inputs), ws_pat, pu_pat, consts), backend.host, f"max_bytes={max_bytes}, word store warm")
    empty = one_string_per_row(["", ""])
    ref = O.WordpieceTokenizer(vocab, "##", 100)(*bert_words(empty), 0)
    fused = FusedSplitWordpiece(RegexSplit("remove", lib=backend.lib), RegexSplit("isolate", lib=backend.lib),
                                WordpieceTokenizer("##", 100, lib=backend.lib))
    assert_same(ref, fused.evaluate(backend.data(empty), ws_pat, pu_pat, consts), backend.host, "all-empty batch")
    with pytest.raises(L.OvtkError) as ei:  # any other split pair is refused, never approximated
        FusedSplitWordpiece(RegexSplit("isolate", lib=backend.lib), RegexSplit("isolate", lib=backend.lib),
                            WordpieceTokenizer("##", 100, lib=backend.lib)).evaluate(backend.data(inputs), ws_pat, pu_pat, consts)
    assert ei.value.code == L.E_UNSUPPORTED


def test_wordpiece_edge_cases(backend):
    """Unknown words (one unk for the whole word, also after partial matches), words over max_bytes_per_word,
    empty rows, > 64 words per row, a vocabulary where a prefix matches but the continuation does not."""
    vocab = [b"[UNK]", b"un", b"##aff", b"##able", b"##a", b"aff", b"a", b"##b", b"able", b"unaffable", b"##" + b"x" * 30,
             b"x" * 30, "é".encode(), "##é".encode()]
    rows = [[b"unaffable", b"unaffablez", b"zzz", b"a", b"ab", b"abz", b"unaff"],
            [],
            [b"x" * 30, b"x" * 60, b"x" * 61, b"x" * 90, b"x" * 91, "éé".encode(), "éz".encode()],
            [b"a"] * 150 + [b"q"] + [b"able"] * 10]
    flat = [w for r in rows for w in r]
    b, e, c = pack_strings(flat)
    cnt = np.array([len(r) for r in rows])
    re_ = np.cumsum(cnt).astype(np.int32)
    inputs = [(re_ - cnt).astype(np.int32), re_, b, e, c]
    for max_bytes in (100, 60, 2):
        ref = O.WordpieceTokenizer(vocab, "##", max_bytes)(*inputs, 0)
        got = WordpieceTokenizer("##", max_bytes, lib=backend.lib).evaluate(
            backend.data(inputs) + list(pack_strings(vocab)) + [np.asarray(0, np.int32)])
        assert_same(ref, got, backend.host, f"max_bytes={max_bytes}")


# ------------------------------------------------------------------ VocabEncoder
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_vocab_encoder(backend, dtype):
    tok = load_tokenizer("bert_small")
    keys = list(tok["vocab"]) + [tok["vocab"][5], b""]  # a duplicate key (the first value wins) and the empty string
    values = (np.arange(len(keys)) * 7 - 3).astype(dtype)
    rng = np.random.default_rng(3)
    queries = [keys[i] for i in rng.integers(0, len(keys), 500)] + [b"not-in-vocab", b"", b"zz", keys[5], b"\xff\xfe"]
    # queries laid out with gaps and in reverse order
    qb, qe, qc = pack_strings(queries[::-1])
    qb, qe = qb[::-1].copy(), qe[::-1].copy()
    ref = O.VocabEncoder(keys, values)(qb, qe, qc, -1)
    got = VocabEncoder(lib=backend.lib).evaluate(backend.data([qb, qe, qc]) + list(pack_strings(keys)) + [values, np.asarray(-1, dtype)])
    assert_same([ref], got, backend.host, "VocabEncoder")
    assert ref[-2] == values[5] and ref[-5] == -1


# ------------------------------------------------------------------ RaggedToDense
@pytest.mark.parametrize("inp, attr_pad_right, input_pad_right, expected", RAGGED_TO_DENSE_KATS)
def test_ragged_to_dense_reference_kats(backend, inp, attr_pad_right, input_pad_right, expected):
    """tests/layer_tests.py:497-598 through the kernel (pad_right input 5 overrides the attribute)."""
    inputs = backend.data([np.asarray(inp["begins"], np.int32), np.asarray(inp["ends"], np.int32),
                           np.asarray(inp["data"], np.int32)]) + [np.asarray(inp["padding_size"], np.int32),
                                                                   np.asarray(inp["value"], np.int32)]
    if input_pad_right is not None:
        inputs.append(np.asarray(input_pad_right, np.bool_))
    dense, mask = RaggedToDense(pad_right=attr_pad_right, lib=backend.lib).evaluate(inputs)
    assert np.array_equal(backend.host(dense), np.asarray(expected, np.int32))
    pad_right = attr_pad_right if input_pad_right is None else input_pad_right
    ref_dense, ref_mask = O.ragged_to_dense(inp["begins"], inp["ends"], np.asarray(inp["data"], np.int32),
                                            inp["padding_size"], inp["value"], pad_right=pad_right)
    assert np.array_equal(backend.host(mask), ref_mask)


@pytest.mark.parametrize("dtype,inner", [(np.int32, ()), (np.int64, ()), (np.uint8, ()), (np.int32, (3,)), (np.int16, (2, 2)),
                                         (np.int16, (2,)), (np.uint8, (4,)), (np.uint8, (2, 2))])   # 4-byte cells of several elements
@pytest.mark.parametrize("pad_right", [True, False])
def test_ragged_to_dense_shapes(backend, dtype, inner, pad_right):
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 40, 50)
    lens[3] = 0
    ends = np.cumsum(lens).astype(np.int32)
    begins = (ends - lens).astype(np.int32)
    data = rng.integers(0, 120, (int(lens.sum()),) + inner).astype(dtype)
    for target in (int(lens.max()), 7, 64):
        ref = O.ragged_to_dense(begins, ends, data, target, 9, pad_right=pad_right)
        got = RaggedToDense(pad_right=pad_right, lib=backend.lib).evaluate(
            backend.data([begins, ends, data]) + [np.asarray(target, np.int32), np.asarray(9, dtype)])
        assert_same(list(ref), got, backend.host, f"RaggedToDense T={target}")


def test_ragged_to_dense_pad_max_length(backend):
    """m_pad_max_length copies `target` elements whatever the row holds (ragged_to_dense.cpp:132-133): rows bleed into
    the following data; a row whose copy would leave the tensor is an error here (undefined in the reference)."""
    begins = np.array([0, 3, 4], np.int32)
    ends = np.array([3, 4, 6], np.int32)
    data = np.arange(12, dtype=np.int32)
    ref = O.ragged_to_dense(begins, ends, data, 5, -1, pad_right=True, pad_max_length=True)
    got = RaggedToDense(pad_right=True, pad_max_length=True, lib=backend.lib).evaluate(
        backend.data([begins, ends, data]) + [np.asarray(5, np.int32), np.asarray(-1, np.int32)])
    assert_same(list(ref), got, backend.host, "pad_max_length")
    with pytest.raises(L.OvtkError) as ei:
        RaggedToDense(pad_right=True, pad_max_length=True, lib=backend.lib).evaluate(
            backend.data([begins, ends, data[:7]]) + [np.asarray(5, np.int32), np.asarray(-1, np.int32)])
    assert ei.value.code == L.E_RANGE


# ------------------------------------------------------------------ VocabDecoder / ByteFallback / FuzeRagged
def detok_vocab():
    tok = load_tokenizer("gpt2_small")
    vocab = list(tok["vocab"])
    vocab[10:10] = []
    # byte-fallback spellings and near misses (byte_fallback.cpp:37, sentence_piece.cpp:27-46)
    extra = [b"<0x41>", b"<0xE2>", b"<0x0A>", b"<0xe2>", b"<0xZZ>", b"<abcd>", b"<0x4>", b"<<x41>", b"<0x41>>", b"<0x<1>", b""]
    return vocab + extra, len(vocab)


@pytest.mark.parametrize("B,S", [(7, 33), (1, 1), (5, 0), (3, 2050)])
def test_vocab_decoder_chain(backend, B, S):
    vocab, n_base = detok_vocab()
    V = len(vocab)
    rng = np.random.default_rng(B * 1000 + S)
    ids = rng.integers(0, V, (B, S)).astype(np.int32)
    if S:
        ids[0, 0] = -5           # negative and >= V ids decode to "" (unsigned compare, vocab_decoder.cpp:71)
        ids[-1, -1] = V + 3
        ids[B // 2, S // 2] = n_base  # "<0x41>"
    skips = [3, 17, n_base + 2, V + 100, -1]
    vconst = list(pack_strings(vocab))
    for skip_attr, skip_in in ((skips, None), ((), None), (skips, np.zeros(0, np.int32)), ((), np.asarray(skips, np.int32))):
        eff = list(skip_attr) if skip_in is None else skip_in.tolist()
        ref = O.vocab_decoder(ids, vocab, eff)
        inputs = backend.data([ids]) + vconst + ([skip_in] if skip_in is not None else [])
        dec = VocabDecoder(skip_tokens=skip_attr, lib=backend.lib)
        got = dec.evaluate(inputs)
        assert_same(list(ref), got, backend.host, "VocabDecoder")
        # ByteFallback on the decoded tokens, then FuzeRagged
        ref_bf = O.byte_fallback(*ref[2:5])
        got_bf = ByteFallback(lib=backend.lib).evaluate(got[2:5])
        assert_same(list(ref_bf), got_bf, backend.host, "ByteFallback")
        ref_fz = O.fuze(ref[0], ref[1], ref_bf[0], ref_bf[1])
        got_fz = FuzeRagged(lib=backend.lib).evaluate(list(got[:2]) + list(got_bf[:2]))
        assert_same(list(ref_fz), got_fz, backend.host, "FuzeRagged")
        # the fused detokenizer gives the same strings in one pass, with and without ByteFallback
        fused = FusedDetokenizer(dec, byte_fallback=True).evaluate(inputs)
        assert_same(list(ref_fz) + [ref_bf[2]], fused, backend.host, "fused detokenizer (byte_fallback)")
        ref_plain = O.fuze(ref[0], ref[1], ref[2], ref[3])
        fused2 = FusedDetokenizer(dec, byte_fallback=False).evaluate(inputs)
        assert_same(list(ref_plain) + [ref[4]], fused2, backend.host, "fused detokenizer")


@pytest.mark.parametrize("bytes_per_id", [None, 0.4])
def test_detokenize_chunked(backend, bytes_per_id):
    """FusedDetokenizer.evaluate_chunked: the caller's loop over row chunks that BASELINE config 5 needs (the reference
    counts chars in int32, src/vocab_decoder.cpp:62-80) -- at a chunk size small enough to see it work: at least three
    chunks, every chunk below the limit, a deliberately low estimate so that a chunk overflows its buffer and is cut
    again; the chunks back to back are the oracle chain's strings."""
    vocab, n_base = detok_vocab()
    V = len(vocab)
    B, S = (90, 48) if backend.name == "emu" else (3000, 96)
    limit = 6000 if backend.name == "emu" else 300_000
    ids = np.random.default_rng(31).integers(-1, V + 1, (B, S)).astype(np.int32)
    ids[B // 3] = 5                      # a row of short tokens next to ...
    ids[B // 3 + 1] = n_base - 1         # ... a row of one long token: the estimate is wrong for both
    inputs = backend.data([ids]) + list(pack_strings(vocab))
    fused = FusedDetokenizer(VocabDecoder(skip_tokens=[3, 17], lib=backend.lib), byte_fallback=True)
    chunks = fused.evaluate_chunked(inputs, chunk_chars=limit, bytes_per_id=bytes_per_id)
    r = O.vocab_decoder(ids, vocab, [3, 17])
    bf = O.byte_fallback(*r[2:5])
    fb, fe = O.fuze(r[0], r[1], bf[0], bf[1])
    want = O.unpack_strings(fb, fe, bf[2])
    got, at = [], 0
    for a, b, cb, ce, cc in chunks:
        assert a == at and b > a
        at = b
        cb, ce, cc = backend.host(cb), backend.host(ce), backend.host(cc)
        assert len(cc) <= limit and cb[0] == 0 and ce[-1] == len(cc) and np.all(cb[1:] == ce[:-1])
        got += O.unpack_strings(cb, ce, cc)
    assert at == B and got == want
    assert len(chunks) >= 3
    overflowed = [c for c in fused.chunk_log if c[3] is None]
    if bytes_per_id:
        assert overflowed, "the low estimate must have produced a chunk that was cut again"
    # with a sink nothing is kept: the chunks arrive in row order
    seen = []
    n = fused.evaluate_chunked(inputs, chunk_chars=limit, bytes_per_id=bytes_per_id,
                               sink=lambda a, b, cb, ce, cc: seen.append((a, b, len(backend.host(cc)))))
    assert n == len(seen) and [x[:2] for x in seen] == [c[:2] for c in chunks] and sum(x[2] for x in seen) == len(bf[2])
    # one row that alone exceeds the chunk size cannot be cut
    with pytest.raises(L.OvtkError) as ei:
        fused.evaluate_chunked(inputs, chunk_chars=40)
    assert ei.value.code == L.E_CAPACITY


def test_detokenize_enqueue_finish(gpu_backend):
    """ovtk_detokenize_enqueue / ovtk_detokenize_finish: several calls in flight on two streams (per-call skip lists
    build their own tables in the call's workspace), each equals the oracle chain; a chars buffer that is too small
    is reported by finish()."""
    import torch
    backend = gpu_backend
    vocab, n_base = detok_vocab()
    V = len(vocab)
    vconst = list(pack_strings(vocab))
    dec = VocabDecoder(skip_tokens=[3, 17], lib=backend.lib)
    fused = FusedDetokenizer(dec, byte_fallback=True)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cases, tickets = [], []
    for k, (B, S, skip_in) in enumerate([(64, 300, None), (5, 0, None), (300, 2050, np.asarray([1, 2, n_base], np.int32)), (1, 1, None),
                                         (2000, 64, np.zeros(0, np.int32))]):
        ids = np.random.default_rng(70 + k).integers(-2, V + 2, (B, S)).astype(np.int32)
        eff = [3, 17] if skip_in is None else skip_in.tolist()
        r = O.vocab_decoder(ids, vocab, eff)
        bf = O.byte_fallback(*r[2:5])
        cases.append(list(O.fuze(r[0], r[1], bf[0], bf[1])) + [bf[2]])
        inputs = backend.data([ids]) + vconst + ([skip_in] if skip_in is not None else [])
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[k % 2]):
            tickets.append(fused.enqueue(inputs))
    for ref, ticket in zip(cases, tickets):
        assert_same(ref, ticket(), backend.host, "detokenize enqueue/finish")
    ids = np.random.default_rng(5).integers(0, n_base, (8, 100)).astype(np.int32)
    ticket = fused.enqueue(backend.data([ids]) + vconst, chars_capacity=10)
    with pytest.raises(L.OvtkError) as ei:
        ticket()
    assert ei.value.code == L.E_CAPACITY
    with pytest.raises(L.OvtkError):   # host arrays have no asynchronous form
        fused.enqueue([ids] + vconst)


def test_byte_fallback_layouts(backend):
    """Strings with gaps / reversed order, every byte value, lower-case hex (-> 0xFF quirk), 6-byte look-alikes."""
    toks = [b"<0x%02X>" % v for v in range(256)] + [b"<0x%02x>" % v for v in (10, 171, 255)] + \
           [b"plain", b"", b"<0xGG>", b"<12345", b"12345>", b"<1234>", b"<12<4>", b"<>>>>>", "<0xé>".encode(), b"<0x41>tail"]
    b, e, c = pack_strings(toks[::-1])
    b, e = b[::-1].copy(), e[::-1].copy()
    ref = O.byte_fallback(b, e, c)
    got = ByteFallback(lib=backend.lib).evaluate(backend.data([b, e, c]))
    assert_same(list(ref), got, backend.host, "ByteFallback")
    assert ref[2][:256].tolist() == list(range(256))


def test_fuze_and_errors(backend):
    rb = np.array([0, 2, 2, 3], np.int32)
    re_ = np.array([2, 2, 3, 5], np.int32)   # row 1 is empty: ends[ragged_ends[row]] (fuze.cpp:37)
    b = np.array([0, 4, 9, 9, 12], np.int32)
    e = np.array([4, 9, 9, 12, 20], np.int32)
    assert_same(list(O.fuze(rb, re_, b, e)), FuzeRagged(lib=backend.lib).evaluate(backend.data([rb, re_, b, e])), backend.host)
    with pytest.raises(L.OvtkError) as ei:  # index past the string tensor: the reference reads out of bounds
        FuzeRagged(lib=backend.lib).evaluate(backend.data([rb, re_ + 3, b, e]))
    assert ei.value.code == L.E_RANGE
    with pytest.raises(L.OvtkError) as ei:  # chars buffer too small
        vocab = [b"abc", b"de"]
        VocabDecoder(lib=backend.lib).evaluate(backend.data([np.zeros((4, 4), np.int32)]) + list(pack_strings(vocab)), chars_capacity=5)
    assert ei.value.code == L.E_CAPACITY
    with pytest.raises(L.OvtkError) as ei:
        VocabEncoder(lib=backend.lib).evaluate(list(pack_strings([b"a"])) + list(pack_strings([b"a"])) + [np.zeros(1, np.float32), np.zeros(1, np.float32)])
    assert ei.value.code == L.E_ARG


def test_fused_wordpiece_enqueue(gpu_backend):
    """ovtk_wordpiece_encode_enqueue / ovtk_encode_finish: two batches in flight, each equal to the blocking call."""
    backend = gpu_backend
    tok = load_tokenizer("bert_small")
    ws_pat = np.frombuffer(rb"\s+", np.uint8)
    from tools.harness import BERT_PUNCT
    pu_pat = np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    consts = list(O.pack_strings(tok["vocab"])) + [np.asarray(tok["unk_id"], np.int32)]
    fused = FusedSplitWordpiece(RegexSplit("remove", lib=backend.lib), RegexSplit("isolate", lib=backend.lib),
                                WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=backend.lib))
    batches = []
    for i, n in enumerate((3000, 800)):
        b, e, c = TextModel(40 + i, "zipf").batch(n, 200)
        c = np.frombuffer(c.tobytes().lower(), np.uint8).copy()
        rb_, re_ = ragged_rows(n)
        batches.append(backend.data([rb_, re_, b, e, c]))
    want = [fused.evaluate(d, ws_pat, pu_pat, consts) for d in batches]
    tickets = [fused.enqueue(d, ws_pat, pu_pat, consts) for d in batches]
    for w_, t in zip(want, tickets):
        got = t()
        assert all(np.array_equal(backend.host(x), backend.host(y)) for x, y in zip(w_, got))
