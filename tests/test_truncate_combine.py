"""Truncate (src/truncate.cpp) and CombineSegments (src/combine_segments.cpp): the oracle against HF's pair
truncation and the reference's known answers; the kernels against the oracle on random ragged batches."""
from pathlib import Path

import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import CombineSegments, RaggedToDense, Truncate
from oracle import oracle as O
from tests.golden.reference_kats import COMBINE_SEGMENTS_KATS
from tests.util import assert_same

G = Path(__file__).parent / "golden"
MODES = ["only_first", "only_second", "longest_first"]


def u8(s):
    return np.frombuffer(s.encode(), np.uint8)


def ragged(rng, n, max_len, gap=3):
    lens = rng.integers(0, max_len + 1, n)
    gaps = rng.integers(0, gap + 1, n)
    begins = np.cumsum(lens + gaps) - lens
    data = rng.integers(0, 50000, int(begins[-1] + lens[-1]) + 1 if n else 1).astype(np.int32)
    return begins.astype(np.int32), (begins + lens).astype(np.int32), data


def test_oracle_matches_hf_longest_first():
    """Kept lengths of HF tokenizers' pair truncation (tests/gen_golden.py main_truncate).  One documented difference:
    on a tie with an odd max_length the reference gives the odd token to the first sequence (`first_length >=
    second_length`, truncate.cpp:85), HF to the second."""
    rows = np.load(G / "golden_truncate_hf.npz")["rows"]
    ties = 0
    for mode, left, m, la, lb, ka, kb in rows.tolist():
        (b0, e0), (b1, e1) = O.truncate([([3], [3 + la]), ([7], [7 + lb])], m, "left" if left else "right", MODES[mode])
        got = (int(e0[0] - b0[0]), int(e1[0] - b1[0]))
        if m % 2 and la == lb and la + lb > m:
            assert got == (m // 2 + 1, m // 2) and (ka, kb) == (m // 2, m // 2 + 1)
            ties += 1
            continue
        assert got == (ka, kb), (mode, left, m, la, lb)
        assert (b0[0] == 3 and b1[0] == 7) if not left else (e0[0] == 3 + la and e1[0] == 7 + lb)
    assert ties > 0 and len(rows) > 1000


@pytest.mark.parametrize("side", ["right", "left"])
@pytest.mark.parametrize("max_length", [0, 1, 7, 16, 1 << 30])
def test_truncate_single(backend, side, max_length):
    rng = np.random.default_rng(max_length % 97 + len(side))
    n = 300 if backend.name == "emu" else 50000
    b, e, d = ragged(rng, n, 40)
    ref = O.truncate([(b, e)], max_length, side)
    got = Truncate(lib=backend.lib).evaluate(backend.data([b, e, d]) + [np.int32(max_length), u8(side), u8("longest_first")])
    assert_same(list(ref[0]), got[:2], backend.host, "Truncate")
    assert np.array_equal(backend.host(got[2]), d)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("side", ["right", "left"])
@pytest.mark.parametrize("max_length", [1, 8, 9, 31])
def test_truncate_pair(backend, mode, side, max_length):
    rng = np.random.default_rng(max_length * 3 + len(mode) + len(side))
    n = 300 if backend.name == "emu" else 50000
    b0, e0, d0 = ragged(rng, n, 24)
    b1, e1, d1 = ragged(rng, n, 24)
    ref = O.truncate([(b0, e0), (b1, e1)], max_length, side, mode)
    got = Truncate(lib=backend.lib).evaluate(backend.data([b0, e0, d0, b1, e1, d1]) + [np.int32(max_length), u8(side), u8(mode)])
    assert_same([ref[0][0], ref[0][1], ref[1][0], ref[1][1]], [got[0], got[1], got[3], got[4]], backend.host, "Truncate pair")
    if mode == "longest_first":
        kept = (backend.host(got[1]) - backend.host(got[0])) + (backend.host(got[4]) - backend.host(got[3]))
        assert kept.max() <= max(max_length, 0) or ((e0 - b0) + (e1 - b1)).max() <= max_length


def test_truncate_errors(backend):
    b, e, d = np.array([0], np.int32), np.array([3], np.int32), np.arange(3, dtype=np.int32)
    with pytest.raises(L.OvtkError, match="Unknown truncation side"):
        Truncate(lib=backend.lib).evaluate([b, e, d, np.int32(2), u8("up"), u8("longest_first")])
    with pytest.raises(L.OvtkError, match="Unknown truncation mode"):
        Truncate(lib=backend.lib).evaluate([b, e, d, b, e, d, np.int32(2), u8("left"), u8("shortest")])
    assert Truncate(lib=backend.lib).evaluate([b[:0], e[:0], d, np.int32(2), u8("left"), u8("")])[0].size == 0


@pytest.mark.parametrize("segments, expected", COMBINE_SEGMENTS_KATS)
def test_combine_segments_kats(backend, segments, expected):
    """tests/layer_tests.py:601-644."""
    inputs = []
    for s in segments:
        inputs += [np.array(s["begins"], np.int32), np.array(s["ends"], np.int32), np.array(s["data"], np.int32)]
    got = CombineSegments(lib=backend.lib).evaluate(backend.data(inputs) + [np.arange(len(segments), dtype=np.int32)])
    assert backend.host(got[0]).tolist() == expected["begins"] and backend.host(got[1]).tolist() == expected["ends"]
    assert backend.host(got[2]).tolist() == expected["data"]
    ids = backend.host(got[5]).tolist()
    want = []
    for i in range(len(expected["begins"])):
        for j, s in enumerate(segments):
            want += [j] * (s["ends"][i] - s["begins"][i])
    assert ids == want


@pytest.mark.parametrize("shape", ["bert_pair", "many", "all_broadcast", "empty_rows"])
def test_combine_segments_random(backend, shape):
    rng = np.random.default_rng(len(shape))
    n = 200 if backend.name == "emu" else 20000
    one = lambda v: (np.array([0], np.int32), np.array([1], np.int32), np.array([v], np.int32))  # noqa: E731
    if shape == "bert_pair":      # [CLS] a [SEP] b [SEP], tokenizer_pipeline.py CombineSegmentsStep
        segs, ids = [one(101), ragged(rng, n, 200), one(102), ragged(rng, n, 90), one(102)], [0, 0, 0, 1, 1]
    elif shape == "many":
        segs, ids = [ragged(rng, n, 9) for _ in range(16)], list(range(16))
    elif shape == "all_broadcast":
        segs, ids = [one(5), (np.array([1], np.int32), np.array([4], np.int32), np.arange(6, dtype=np.int32))], [3, -7]
    else:
        b, e, d = ragged(rng, n, 3)
        segs, ids = [(b, b.copy(), d), ragged(rng, n, 1), (np.array([2], np.int32), np.array([2], np.int32), d[:4])], [9, 8, 7]
    ref = O.combine_segments(segs, ids)
    inputs = [x for s in segs for x in s]
    got = CombineSegments(lib=backend.lib).evaluate(backend.data(inputs) + [np.array(ids, np.int32)])
    assert_same([ref[0], ref[1], ref[2], ref[0], ref[1], ref[3]], got, backend.host, "CombineSegments")


def test_combine_then_pad(backend):
    """The tail of every encode pipeline: Truncate -> CombineSegments -> RaggedToDense (tokenizer_pipeline.py:894-1100)."""
    rng = np.random.default_rng(5)
    n = 100 if backend.name == "emu" else 5000
    b, e, d = ragged(rng, n, 60)
    lib = backend.lib
    tb, te, _ = Truncate(lib=lib).evaluate(backend.data([b, e, d]) + [np.int32(30), u8("right"), u8("longest_first")])
    cls = (np.array([0], np.int32), np.array([1], np.int32), np.array([101], np.int32))
    sep = (np.array([0], np.int32), np.array([1], np.int32), np.array([102], np.int32))
    cb, ce, cd, _, _, ci = CombineSegments(lib=lib).evaluate(backend.data(list(cls)) + [tb, te] + backend.data([d]) + backend.data(list(sep))
                                                            + [np.array([0, 0, 0], np.int32)])
    rb, re_ = O.truncate([(b, e)], 30, "right")[0]
    ref = O.combine_segments([cls, (rb, re_, d), sep], [0, 0, 0])
    assert_same([ref[0], ref[1], ref[2]], [cb, ce, cd], backend.host, "chain")
    assert not backend.host(ci).any()
    dense, mask = RaggedToDense(lib=lib).evaluate([cb, ce, cd, np.int32(32), np.int32(0)])
    dense, mask = backend.host(dense), backend.host(mask)
    lens = np.minimum(e - b, 30) + 2
    assert dense.shape == (n, 32) and np.array_equal(mask.sum(1), lens) and (dense[:, 0] == 101).all()
    assert all(dense[i, lens[i] - 1] == 102 for i in range(n))


def test_combine_segments_errors(backend):
    b, e, d = np.array([0, 1], np.int32), np.array([1, 9], np.int32), np.arange(3, dtype=np.int32)
    with pytest.raises(L.OvtkError) as ei:
        CombineSegments(lib=backend.lib).evaluate([b, e, d, np.array([0], np.int32)])
    assert ei.value.code == L.E_RANGE
    ok = np.array([1, 3], np.int32)
    with pytest.raises(L.OvtkError) as ei:
        CombineSegments(lib=backend.lib).evaluate([b, ok, d, np.array([0], np.int32)], capacity=2)
    assert ei.value.code == L.E_CAPACITY
    with pytest.raises(L.OvtkError) as ei:
        CombineSegments(lib=backend.lib).evaluate([b, ok, d, b[:1], ok[:1], d, np.zeros(3, np.int32), np.zeros(3, np.int32), d,
                                                   np.array([0, 1, 2], np.int32)])
    assert ei.value.code == L.E_ARG


@pytest.mark.parametrize("shape", ["single", "pair_longest", "pair_left_only_first", "no_trunc_padleft", "fixed_width"])
def test_fused_encode_tail(backend, shape):
    """ovtk_encode_tail_run = Truncate -> CombineSegments -> RaggedToDense (ids, mask) + RaggedToDense (segment ids),
    compared with that chain of the oracle's restatements."""
    from openvino_tokenizers_amd.ops import FusedEncodeTail
    rng = np.random.default_rng(len(shape))
    n = 120 if backend.name == "emu" else 6000
    one = lambda v: (np.array([0], np.int32), np.array([1], np.int32), np.array([v], np.int32))  # noqa: E731
    a, b2 = ragged(rng, n, 70), ragged(rng, n, 50)
    cfg = dict(single=dict(segs=[one(101), a, one(102)], ids=[0, 0, 0], trunc=[1], m=40, side="right", mode="longest_first", pad_right=True, T=None),
               pair_longest=dict(segs=[one(101), a, one(102), b2, one(102)], ids=[0, 0, 0, 1, 1], trunc=[1, 3], m=61, side="right", mode="longest_first", pad_right=True, T=None),
               pair_left_only_first=dict(segs=[a, one(5), b2], ids=[0, 0, 1], trunc=[0, 2], m=30, side="left", mode="only_first", pad_right=True, T=None),
               no_trunc_padleft=dict(segs=[one(1), a], ids=[7, 3], trunc=[], m=2**31 - 1, side="right", mode="longest_first", pad_right=False, T=None),
               fixed_width=dict(segs=[one(101), a, one(102)], ids=[0, 0, 0], trunc=[1], m=64, side="right", mode="longest_first", pad_right=True, T=48))[shape]
    segs = [tuple(x.copy() for x in s) for s in cfg["segs"]]
    # the chain, with the oracle's restatements
    cut = [list(s) for s in segs]
    if cfg["trunc"]:
        res = O.truncate([(segs[j][0], segs[j][1]) for j in cfg["trunc"]], cfg["m"], cfg["side"], cfg["mode"])
        for j, (tb, te) in zip(cfg["trunc"], res):
            cut[j][0], cut[j][1] = tb, te
    cb, ce, cd, ci = O.combine_segments([tuple(s) for s in cut], cfg["ids"])
    width = int((ce - cb).max()) if cfg["T"] is None else cfg["T"]
    want_ids, want_mask = O.ragged_to_dense(cb, ce, cd, width, 9, cfg["pad_right"])
    want_types, _ = O.ragged_to_dense(cb, ce, ci, width, 4, cfg["pad_right"])
    op = FusedEncodeTail(cfg["m"], cfg["side"], cfg["mode"], cfg["pad_right"], lib=backend.lib)
    dsegs = [tuple(backend.data(list(s))) for s in segs]
    got = op.evaluate(dsegs, cfg["ids"], truncated=cfg["trunc"], pad_value=9, type_pad_value=4, target_dim=cfg["T"])
    assert_same([want_ids, want_mask.astype(bool), want_types], got, backend.host, "FusedEncodeTail")


@pytest.mark.parametrize("with_special", [False, True])
def test_encode_dense_equals_the_chain(backend, with_special):
    """ovtk_encode_dense_*: [SpecialTokensSplit ->] RegexSplit -> BPETokenizer -> Truncate -> CombineSegments (constant BOS / EOS) ->
    RaggedToDense x 2 in one call, the dense tensors written by the encode's last pass = the oracle's ragged ids through the oracle's
    truncate / combine_segments / ragged_to_dense (src/truncate.cpp:37-150, src/combine_segments.cpp:36-134,
    src/ragged_to_dense.cpp:70-174).  Both truncation sides, both padding sides, longest-row and fixed target_dim (also one below
    the longest row), batches of one launch and of the span kernel's size."""
    from openvino_tokenizers_amd.ops import BPETokenizer, FusedEncodeDense, RegexSplit, SpecialTokensSplit
    from tests.test_special_tokens import _texts_with_specials, u8
    from tests.util import BpeTok, one_string_per_row
    if backend.name == "hip-host":
        pytest.skip("device tensors only")
    tok = BpeTok.load("gpt2_small")
    pat = O.special_tokens_pattern([("<|endoftext|>", False, False)])
    rng = np.random.default_rng(3 + with_special)
    for n_rows in (24, 300):
        strings = _texts_with_specials(rng, n_rows) if with_special else [s.replace(b"<|endoftext|>", b"") for s in _texts_with_specials(rng, n_rows)]
        strings[2] = b""
        inputs = one_string_per_row(strings)
        if with_special:
            s_ref = O.SpecialTokensSplit(pat)(*inputs)
            rb, re_, ids = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(*s_ref[:5], skips=s_ref[5])[:5])
        else:
            rb, re_, ids = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(*inputs)[:5])
        for case, (max_length, side, pad_right, target, pre, suf) in enumerate([(1 << 20, "right", True, None, (), ()), (20, "right", True, None, (7,), (9, 11)),
                                                                                (33, "left", False, None, (5,), ()), (16, "right", False, 40, (), (3,)),
                                                                                (64, "left", True, 30, (1, 2), (3, 4))]):
            (tb, te), = O.truncate([(rb, re_)], max_length, side, "longest_first")
            segs = []
            if pre:
                segs.append((np.zeros(1, np.int32), np.full(1, len(pre), np.int32), np.asarray(pre, np.int32)))
            segs.append((tb, te, ids))
            if suf:
                segs.append((np.zeros(1, np.int32), np.full(1, len(suf), np.int32), np.asarray(suf, np.int32)))
            cb, ce, cdata = O.combine_segments(segs, list(range(len(segs))))[:3]
            width = int((ce - cb).max()) if target is None else target
            ref_ids, ref_mask = O.ragged_to_dense(cb, ce, cdata, width, 50256, pad_right=pad_right)
            op = FusedEncodeDense(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib),
                                  SpecialTokensSplit(lib=backend.lib) if with_special else None, max_length=max_length, trunc_side=side,
                                  pad_right=pad_right, pad_value=50256, prefix=pre, suffix=suf)
            # (a second call runs on what the memo learned: rows without unused staging entries; on the emulator for two of the cases)
            for call in range(2 if backend.name != "emu" or (n_rows == 300 and case in (1, 4)) else 1):
                got = op.evaluate(backend.data(inputs), tok.pattern_u8(), tok.consts, special_pattern=u8(pat) if with_special else None, target_dim=target,
                                  row_capacity=max(width, 1))
                what = f"dense: special {with_special}, {n_rows} rows, max_length {max_length} {side}, pad_right {pad_right}, target {target}, call {call}"
                assert backend.host(got[0]).shape == ref_ids.shape, what
                assert np.array_equal(backend.host(got[0]), ref_ids), what
                assert np.array_equal(backend.host(got[1]), ref_mask.astype(bool)), what
    # too little room: OVTK_E_CAPACITY
    from openvino_tokenizers_amd import _lib as L
    op = FusedEncodeDense(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    with pytest.raises(L.OvtkError) as ei:
        op.evaluate(backend.data(inputs), tok.pattern_u8(), tok.consts, row_capacity=3)
    assert ei.value.code == L.E_CAPACITY
