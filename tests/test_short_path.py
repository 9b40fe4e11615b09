"""The short path (round 6): span -> compact.  lookup_span_kernel looks the pieces the memo does not hold up in the piece store itself
and sums its rows' counts per tile; compact_kernel follows at once; lookup_kernel<kFused> / merge_kernel are launched only when a
wave reports that it could not.  The reference's piece cache (src/bpe_tokenizer.cpp:197-205, 331-338) is pure memoisation: results
never depend on the path taken."""
import ctypes as C

import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
from oracle import oracle as O
from tools.harness import BpeTok
from tools.workloads import TextModel, ragged_rows

from .util import assert_same


def _stats(lib):
    t, x = C.c_int64(), C.c_int64()
    L.check(lib, lib.ovtk_short_path_stats(C.byref(t), C.byref(x)))
    return int(t.value), int(x.value)


@pytest.fixture
def short_always(backend):
    L.check(backend.lib, backend.lib.ovtk_set_short_path(2))
    yield backend
    L.check(backend.lib, backend.lib.ovtk_set_short_path(1))


@pytest.mark.parametrize("name", ["gpt2", "llama3"])
def test_short_path_equals_oracle_while_learning_and_after(short_always, name):
    """Mode 2: every call tries.  First sight of a text: waves with pieces to merge report inexact, the other kernels follow, the
    result is the oracle's.  Repeats: the store holds what was merged, every wave is exact, ONE kernel -- the same result."""
    backend = short_always
    lib = backend.lib
    # (the emulator: the small tokenizers, a few waves in two blocks; the GPU: BASELINE's tokenizers -- their stores hold the text's
    # pieces --, a wave per row, 79 groups of 64 waves)
    tok = BpeTok.load(name + "_small" if backend.name == "emu" else name)
    bpe = BPETokenizer(**tok.attrs, lib=lib)
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), bpe)
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    n = 70 if backend.name == "emu" else 5000
    b, e, c = TextModel(91, "zipf").batch(n, 300 if backend.name == "emu" else 160)
    rb, re_ = ragged_rows(n)
    ref = orc(*rs(rb, re_, b, e, c)[:5])
    exact = []
    for rep in range(4):
        t0, x0 = _stats(lib)
        assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host, f"short path, call {rep}")
        t1, x1 = _stats(lib)
        assert t1 == t0 + 1, "the call was not launched as span -> compact"
        exact.append(x1 - x0)
    if name == "gpt2":   # (a Llama-3 id needs more than 16 bits: a store entry holds seven of them, and the text has words of more)
        assert exact[-1] == 1, f"after three sights of the text a call still needed the other kernels: {exact}"


def test_short_path_off_and_auto_give_the_same(backend):
    """Modes 0 / 1 / 2 on the same batches: identical outputs (mode 1 backs off while the tables learn and comes back)."""
    lib = backend.lib
    tok = BpeTok.load("gpt2_small")
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    n = 300 if backend.name == "emu" else 3000   # (more than 256 rows: not the one-launch small path)
    b, e, c = TextModel(92, "zipf").batch(n, 120)
    rb, re_ = ragged_rows(n)
    ref = orc(*rs(rb, re_, b, e, c)[:5])
    try:
        for mode in (0, 1, 2):
            L.check(lib, lib.ovtk_set_short_path(mode))
            bpe = BPETokenizer(**tok.attrs, lib=lib)
            fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), bpe)
            t0, x0 = _stats(lib)
            for rep in range(5 if mode == 1 else 2):
                assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host, f"mode {mode}, call {rep}")
            t1, x1 = _stats(lib)
            if mode == 0:
                assert t1 == t0
            if mode == 1:
                assert x1 > x0, "mode 1 never came back to the short path on a text it has seen five times"
    finally:
        L.check(lib, lib.ovtk_set_short_path(1))
    assert lib.ovtk_set_short_path(3) != 0


def test_short_path_rows_it_cannot_finish(short_always):
    """Empty rows, rows of several strings, skipped strings and pieces longer than the store's keys: the waves that hold them report
    inexact, the call takes the other kernels, the result is the oracle's."""
    backend = short_always
    lib = backend.lib
    tok = BpeTok.load("gpt2_small")
    bpe = BPETokenizer(**tok.attrs, lib=lib)
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), bpe)
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    rng = np.random.default_rng(5)
    texts = []
    for i in range(90):
        k = i % 6
        if k == 0:
            texts.append(b"")
        elif k == 1:
            texts.append(b"x" * 45 + b" and " + b"?" * 40)   # pieces of more than 31 bytes
        else:
            texts.append(bytes(rng.choice(list(b"abc de.,'s 12\n"), size=int(rng.integers(1, 200))).astype(np.uint8)))
    lens = np.array([len(t) for t in texts])
    e = np.cumsum(lens).astype(np.int32)
    b = (e - lens).astype(np.int32)
    c = np.frombuffer(b"".join(texts), dtype=np.uint8).copy()
    # rows: mostly one string each, some of two strings, one of none
    rb, re_, at = [], [], 0
    while at < len(texts):
        take = 2 if (at % 7 == 3 and at + 2 <= len(texts)) else 1
        rb.append(at)
        re_.append(at + take)
        at += take
    rb.append(len(texts))
    re_.append(len(texts))
    rb, re_ = np.array(rb, np.int32), np.array(re_, np.int32)
    skips = (np.arange(len(texts)) % 11 == 5)
    ref = orc(*rs(rb, re_, b, e, c, skips=skips)[:5])
    for rep in range(3):
        got = fused.evaluate(backend.data([rb, re_, b, e, c, skips.astype(np.uint8)]) + [pat], tok.consts)
        assert_same(ref, got, backend.host, f"call {rep}")


def test_short_path_wordpiece(short_always):
    """The fused WordPiece encode (the BERT words as lookup_span_kernel's scan) takes the short path too: the words the word memo does
    not hold are looked up in the word store by the span kernel, wordpiece_deferred_kernel is launched only while words are still to be
    walked through the tries.  src/wordpiece_tokenizer.cpp:96-130 per word; every call equals the oracle's chain."""
    from openvino_tokenizers_amd.ops import FusedSplitWordpiece, WordpieceTokenizer
    from tests.test_ops_parity import BERT_PUNCT, BERT_WS, bert_words, wp_consts
    from tools.harness import pack_strings
    from tools.make_tokenizers import load_tokenizer
    backend = short_always
    lib = backend.lib
    tok = load_tokenizer("bert_small" if backend.name == "emu" else "bert")
    n = 320 if backend.name == "emu" else 6000
    b, e, c = TextModel(93, "zipf").batch(n, 110)
    c = np.frombuffer(c.tobytes().lower(), np.uint8).copy()
    rb, re_ = ragged_rows(n)
    inputs = [rb, re_, b, e, c]
    ws_pat, pu_pat = np.frombuffer(BERT_WS.encode(), np.uint8), np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    ref = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*bert_words(inputs), tok["unk_id"])
    fused = FusedSplitWordpiece(RegexSplit("remove", lib=lib), RegexSplit("isolate", lib=lib),
                                WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=lib))
    exact = []
    for rep in range(4):
        t0, x0 = _stats(lib)
        assert_same(ref, fused.evaluate(backend.data(inputs), ws_pat, pu_pat, wp_consts(tok)), backend.host, f"WordPiece, call {rep}")
        t1, x1 = _stats(lib)
        assert t1 == t0 + 1, "the call was not launched as span -> compact"
        exact.append(x1 - x0)
    # (a word that comes out as unk is never filed -- input 8 may differ per call --, so a text with such words keeps its deferred kernel)
    assert exact[0] == 0 or exact[-1] == 1
