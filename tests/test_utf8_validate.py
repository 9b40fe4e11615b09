"""UTF8Validate (src/utf8_validate.cpp): the reference's known answers through the kernel (Python's bytes.decode is
the expected value there, tests/layer_tests.py:131-139) and the kernel against the oracle on random byte soup."""
import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import UTF8Validate
from oracle import oracle as O
from tests.golden.reference_kats import UTF8_VALIDATE_KATS
from tests.util import assert_same


@pytest.mark.parametrize("mode", ["ignore", "replace"])
def test_reference_kats(backend, mode):
    strings = list(UTF8_VALIDATE_KATS) + [b""]
    b, e, c = O.pack_strings(strings)
    got = UTF8Validate(replace_mode=mode == "replace", lib=backend.lib).evaluate(backend.data([b, e, c]))
    out = O.unpack_strings(*[backend.host(x) for x in got])
    assert out == [s.decode(errors=mode).encode() for s in strings]


@pytest.mark.parametrize("replace", [False, True])
@pytest.mark.parametrize("flavour", ["soup", "mostly_valid", "leads_only", "long"])
def test_random_bytes(backend, replace, flavour):
    rng = np.random.default_rng(len(flavour) + replace)
    n = 150 if backend.name == "emu" else 6000
    interesting = np.array([0x00, 0x41, 0x7F, 0x80, 0xBF, 0xC0, 0xC2, 0xDF, 0xE0, 0xED, 0xEF, 0xF0, 0xF4, 0xF5, 0xF7, 0xF8, 0xFF, 0xA0, 0x9F, 0x90],
                           np.uint8)
    strings = []
    for i in range(n):
        ln = int(rng.integers(0, 40))
        if flavour == "soup":
            s = bytes(rng.choice(interesting, ln))
        elif flavour == "mostly_valid":
            s = bytearray("".join(rng.choice(list("aé元😁 ߿ࠀ￿𐀀"), ln)).encode())
            for _ in range(int(rng.integers(0, 3))):
                if s:
                    s[int(rng.integers(0, len(s)))] = int(rng.choice(interesting))
            s = bytes(s)
        elif flavour == "leads_only":
            s = bytes(rng.choice(np.array([0xC3, 0xE2, 0xF0, 0x82], np.uint8), ln))
        else:
            s = bytes(rng.integers(0, 256, int(rng.integers(500, 3000)), dtype=np.uint8)) if i < 12 else b"x"
        strings.append(s)
    b, e, c = O.pack_strings(strings)
    ref = O.utf8_validate(b, e, c, replace)
    got = UTF8Validate(replace_mode=replace, lib=backend.lib).evaluate(backend.data([b, e, c]))
    assert_same(list(ref), got, backend.host, "UTF8Validate")
    if flavour == "soup":   # where the reference agrees with Python: no lead above 0xF4, no surrogates / E0,F0,F4 range cases
        for s, o in zip(strings, O.unpack_strings(*[backend.host(x) for x in got])):
            if not (set(s) & {0xE0, 0xED, 0xF0, 0xF4, 0xF5, 0xF7, 0xC0}):
                assert o == s.decode(errors="replace" if replace else "ignore").encode()


def test_offsets_start_at_first_begin(backend):
    """utf8_validate.cpp:46: out_idx starts at begins[0]; gaps between strings are dropped."""
    c = np.frombuffer(b"..ab\xff..c\xc3.", np.uint8)
    b, e = np.array([2, 7, 8], np.int32), np.array([5, 9, 8], np.int32)
    ref = O.utf8_validate(b, e, c, True)
    got = UTF8Validate(replace_mode=True, lib=backend.lib).evaluate(backend.data([b, e, c]))
    assert backend.host(got[0]).tolist() == ref[0].tolist() == [2, 7, 11] and backend.host(got[1]).tolist() == ref[1].tolist() == [7, 11, 11]
    assert np.array_equal(backend.host(got[2])[2:], ref[2][2:])


def test_errors(backend):
    c = np.frombuffer(b"abc", np.uint8)
    with pytest.raises(L.OvtkError) as ei:
        UTF8Validate(lib=backend.lib).evaluate([np.array([0], np.int32), np.array([9], np.int32), c])
    assert ei.value.code == L.E_RANGE
    out = UTF8Validate(lib=backend.lib).evaluate([np.zeros(0, np.int32), np.zeros(0, np.int32), c])
    assert out[0].size == 0 and out[2].size == 0


@pytest.mark.parametrize("replace", [False, True])
def test_symbols_across_the_64_byte_windows(backend, replace):
    """The kernel works a string through in windows of 64 bytes (ops_kernels.hpp utf8_windows): every kind of symbol -- valid, overlong,
    cut short, stray continuation bytes, leads behind leads -- at every offset around the window edges, and at the string's end."""
    seqs = [b"\xc3\xa9", b"\xe5\x85\x83", b"\xf0\x9f\x98\x81", b"\xc0\xaf", b"\xc1\xbf", b"\xe0\x80\xaf", b"\xe0\x9f\xbf", b"\xe0\xa0\x80",
            b"\xf0\x80\x80\xaf", b"\xf0\x8f\xbf\xbf", b"\xf0\x90\x80\x80", b"\xf7\xbf\xbf\xbf", b"\xc3", b"\xe5\x85", b"\xe5", b"\xf0\x9f\x98", b"\xf0\x9f",
            b"\xf0", b"\x80", b"\x80\x80\x80\x80", b"\xc3\xa9\x80", b"\xe5\x85\x83\xbf\xbf", b"\xf8\x80", b"\xff", b"\xc3\xc3\xa9", b"\xe5\xc3\xa9",
            b"\xf0\xe5\x85\x83", b"\xe5\x85\xc3", b"\xf0\x9f\x98\xf0\x9f\x98\x81", b"\xed\xa0\x80", b"\xf4\x90\x80\x80"]
    strings = []
    for k in list(range(57, 68)) + [0, 1, 124, 126, 127, 128, 191]:
        for q in seqs:
            strings.append(b"a" * k + q)                 # ... at the string's end
            strings.append(b"a" * k + q + b"z" * 5)      # ... inside
            strings.append(b"\xc3\xa9" * (k // 2) + b"b" * (k % 2) + q + b"\xe5\x85\x83")   # ... with carries on both sides
    b, e, c = O.pack_strings(strings)
    ref = O.utf8_validate(b, e, c, replace)
    got = UTF8Validate(replace_mode=replace, lib=backend.lib).evaluate(backend.data([b, e, c]))
    assert_same(list(ref), got, backend.host, "UTF8Validate at the window edges")
