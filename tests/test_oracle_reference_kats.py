"""Pins the oracle against the reference's own known-answer tests (tests/layer_tests.py:331-389, 497-598)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.golden.reference_kats import (COMBINE_SEGMENTS_KATS, RAGGED_TO_DENSE_KATS, REGEX_SPLIT_KATS,
                                         SPECIAL_TOKENS_KATS, UTF8_VALIDATE_KATS)
from tests.util import one_string_per_row


@pytest.mark.parametrize("text, expected, layer", REGEX_SPLIT_KATS)
def test_regex_split_kat(text, expected, layer):
    pattern, behaviour, invert = layer
    out = O.RegexSplit(pattern, behaviour, invert)(*one_string_per_row([text]))
    # the reference test packs outputs 2..4 back into strings (StringTensorPack) and compares
    got = tuple(s.decode("utf-8") for s in O.unpack_strings(out[2], out[3], out[4]))
    assert got == expected


@pytest.mark.parametrize("inp, attr_pad_right, input_pad_right, expected", RAGGED_TO_DENSE_KATS)
def test_ragged_to_dense_kat(inp, attr_pad_right, input_pad_right, expected):
    pad_right = attr_pad_right if input_pad_right is None else input_pad_right  # input 5 overrides the attribute
    dense, mask = O.ragged_to_dense(inp["begins"], inp["ends"], np.asarray(inp["data"], np.int32),
                                    inp["padding_size"], inp["value"], pad_right=pad_right)
    assert np.array_equal(dense, np.asarray(expected, np.int32))
    lens = np.minimum(np.asarray(inp["ends"]) - np.asarray(inp["begins"]), inp["padding_size"])
    for r, n in enumerate(lens):
        row = mask[r]
        assert row.sum() == n and (row[:n].all() if pad_right else row[len(row) - n:].all())


@pytest.mark.parametrize("tokens, text, expected, expected_skips", SPECIAL_TOKENS_KATS)
def test_special_tokens_split_kat(tokens, text, expected, expected_skips):
    out = O.SpecialTokensSplit(O.special_tokens_pattern(tokens))(*one_string_per_row([text]))
    got = tuple(s.decode("utf-8") for s in O.unpack_strings(out[2], out[3], out[4]))
    assert got == expected and out[5].tolist() == expected_skips


@pytest.mark.parametrize("segments, expected", COMBINE_SEGMENTS_KATS)
def test_combine_segments_kat(segments, expected):
    ob, oe, od, oi = O.combine_segments([(s["begins"], s["ends"], s["data"]) for s in segments], np.arange(len(segments)))
    assert ob.tolist() == expected["begins"] and oe.tolist() == expected["ends"] and od.tolist() == expected["data"]


@pytest.mark.parametrize("mode", ["ignore", "replace"])
@pytest.mark.parametrize("raw", UTF8_VALIDATE_KATS)
def test_utf8_validate_kat(raw, mode):
    """tests/layer_tests.py:131-139: the op's output equals Python's bytes.decode(errors=mode)."""
    b, e, c = O.pack_strings([raw, b"", raw])
    ob, oe, oc = O.utf8_validate(b, e, c, mode == "replace")
    want = raw.decode(errors=mode).encode()
    assert O.unpack_strings(ob, oe, oc) == [want, b"", want]
