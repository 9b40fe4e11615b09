"""StringTensorUnpack / Pack: the packed-u8 wire format of parse_packed_strings (src/utils.cpp:18-29)."""
import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import StringTensorPack, StringTensorUnpack
from oracle import oracle as O


def wire(strings):
    """The reference's serialisation: [i32 n][i32 begin_0][i32 end_i x n][bytes] (src/utils.cpp:18-29)."""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
    ends = np.cumsum([len(b) for b in bs]).astype(np.int32)
    head = np.concatenate([[len(bs), 0], ends]).astype(np.int32) if bs else np.array([0, 0], np.int32)
    return np.concatenate([head.view(np.uint8), np.frombuffer(b"".join(bs), np.uint8)])


STRINGS = ["Eng... test, string?!", "", "多语言 テスト 😁", "a", "\x00\x01 binary \xff".encode("latin-1"), "x" * 1000, ""]


@pytest.mark.parametrize("where", ["host", "device"])
def test_unpack_matches_reference_layout(backend, where):
    if where == "device" and backend.name != "hip-device":
        pytest.skip("device-resident packed buffer needs CUDA tensors")
    packed = wire(STRINGS)
    got = StringTensorUnpack(lib=backend.lib).evaluate(backend.data([packed]) if where == "device" else [packed])
    b, e, c = [backend.host(x) for x in got]
    rb, re_, rc = O.pack_strings(STRINGS)
    assert np.array_equal(b, rb) and np.array_equal(e, re_) and np.array_equal(c, rc)
    assert O.unpack_strings(b, e, c) == [s.encode() if isinstance(s, str) else s for s in STRINGS]


def test_unpack_format_errors(backend):
    op = StringTensorUnpack(lib=backend.lib)
    with pytest.raises(L.OvtkError, match="no batch size"):
        op.evaluate([np.zeros(3, np.uint8)])
    with pytest.raises(L.OvtkError, match="first string offset and end indices"):
        op.evaluate([np.array([5, 0, 1], np.int32).view(np.uint8)])
    with pytest.raises(L.OvtkError) as ei:
        op.evaluate([np.array([1, 0, 99], np.int32).view(np.uint8)])
    assert ei.value.code == L.E_RANGE
    empty = op.evaluate([np.array([0, 0], np.int32).view(np.uint8)])
    assert all(len(x) == 0 for x in empty)
    with pytest.raises(L.OvtkError):
        StringTensorUnpack(mode="chars", lib=backend.lib)


def test_pack_roundtrip_with_gaps_and_order(backend):
    rng = np.random.default_rng(3)
    n = 300 if backend.name == "emu" else 20000
    strings = [bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)) for _ in range(n)]
    b, e, c = O.pack_strings(strings)
    # scatter the strings: reversed order in memory with gaps, as ops are allowed to hand them over
    lens = e - b
    starts = (np.cumsum((lens + 3)[::-1])[::-1] - (lens + 3)).astype(np.int32)
    chars = np.zeros(int((lens + 3).sum()) + 1, np.uint8)
    for i in range(len(strings)):
        chars[starts[i]:starts[i] + lens[i]] = np.frombuffer(strings[i], np.uint8)
    packed = StringTensorPack(lib=backend.lib).evaluate(backend.data([starts, (starts + lens).astype(np.int32), chars]))[0]
    assert np.array_equal(np.asarray(packed), wire(strings))
    back = StringTensorUnpack(lib=backend.lib).evaluate([np.asarray(packed)])
    assert O.unpack_strings(*[backend.host(x) for x in back]) == strings
    with pytest.raises(L.OvtkError) as ei:
        StringTensorPack(lib=backend.lib).evaluate([np.array([0], np.int32), np.array([99999999], np.int32), chars[:4]])
    assert ei.value.code == L.E_RANGE


@pytest.mark.parametrize("pinned", [False, True])
def test_encode_from_the_packed_tensor(backend, pinned):
    """ovtk_encode_enqueue_packed = StringTensorUnpack -> RegexSplit -> BPETokenizer with ONE buffer over PCIe: the packed u8
    tensor of a batch against the oracle chain on the unpacked strings; empty strings, an empty batch, the all-empty batch
    quirk (regex_split.cpp:129-143), format errors."""
    from openvino_tokenizers_amd import _lib as L
    from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
    from oracle import oracle as O
    from tests.util import BpeTok, assert_same, one_string_per_row
    from tools.workloads import TextModel
    if backend.name == "hip-device":
        pytest.skip("the packed tensor is host memory by definition")
    if pinned and backend.name == "emu":
        pytest.skip("pinned memory needs the HIP runtime")
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    b, e, c = TextModel(61, "mixed").batch(40 if backend.name == "emu" else 700, 180)
    raw = c.tobytes()
    strings = [raw[x:y] for x, y in zip(b.tolist(), e.tolist())]
    strings[3] = b""
    strings[-1] = b""

    def host(a):
        if not pinned:
            return a
        import torch
        t = torch.empty(a.shape, dtype=getattr(torch, str(a.dtype)), pin_memory=True).numpy()
        t[...] = a
        return t
    for batch in (strings, strings[:1], [b"", b""], []):
        packed = host(wire(batch))
        n = len(batch)
        n_chars = sum(len(s) for s in batch)
        outs = tuple(host(np.full(m, -1, np.int32)) for m in (max(n, 1), max(n, 1), max(n_chars, 1)))
        got = fused.enqueue_packed(packed, tok.pattern_u8(), tok.consts, outs)()
        ref = orc(*rs(*one_string_per_row(batch))[:5]) if n_chars else None
        if ref is not None:
            assert_same(list(ref), got, backend.host, f"packed batch of {n}")
        else:   # no chars at all: RegexSplit's shape-{1} quirk (regex_split.cpp:129-143), one empty row whatever the batch held
            assert len(got[2]) == 0 and got[0].tolist() == [0] and got[1].tolist() == [0]
    with pytest.raises(L.OvtkError) as ei:
        fused.enqueue_packed(np.zeros(2, np.uint8), tok.pattern_u8(), tok.consts)
    assert ei.value.code == L.E_ARG
    bad = wire(strings[:4]).copy()
    bad[:4].view(np.int32)[0] = 1000   # more strings than the buffer has offsets for
    with pytest.raises(L.OvtkError) as ei:
        fused.enqueue_packed(bad, tok.pattern_u8(), tok.consts)
    assert ei.value.code == L.E_ARG
    bad = wire(strings[:4]).copy()
    bad[4 + 4 * 4: 8 + 4 * 4].view(np.int32)[0] = 10 ** 6   # the last end offset leaves the buffer
    with pytest.raises(L.OvtkError) as ei:
        fused.enqueue_packed(bad, tok.pattern_u8(), tok.consts)
    assert ei.value.code == L.E_RANGE
