"""RegexSplit / BPETokenizer / fused encode: oracle vs HF golden vectors, and the kernels vs the oracle.

`backend` runs every kernel test three ways: "emu" (CPU, the kernel sources under the SIMT emulator -- logic
only), "hip-host" and "hip-device" (gpu-marked: libovtk_amd.so on the MI355X with host resp. device buffers).
Bit-exact comparison throughout (integer work).
"""
import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
from oracle import oracle as O
from tests.util import BpeTok, assert_same, one_string_per_row
from tools.make_tokenizers import GPT2_PATTERN
from tools.workloads import TextModel, ragged_rows

GOLDEN = __import__("pathlib").Path(__file__).parent / "golden"
DIGITS_PATTERN = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"


# ------------------------------------------------------------------ oracle pinned by HF golden vectors (CPU)
def test_oracle_matches_hf_golden():
    z = np.load(GOLDEN / "golden_bpe_gpt2_small.npz")
    tok = BpeTok.load("gpt2_small")
    rb, re_ = ragged_rows(len(z["begins"]))
    sp = O.RegexSplit(tok.pattern, "isolate")(rb, re_, z["begins"], z["ends"], z["chars"])
    ob, oe, ids = tok.oracle()(*sp[:5])
    assert np.array_equal(ob, z["id_begins"]) and np.array_equal(oe, z["id_ends"])
    assert np.array_equal(ids, z["ids"])


def test_oracle_matches_hf_live():
    """Same check against a live HF tokenizer on fresh seeded text (skipped where `tokenizers` is absent)."""
    tokenizers = pytest.importorskip("tokenizers")
    hf = tokenizers.Tokenizer.from_file(str(GOLDEN / "tok_gpt2_small.hf.json"))
    tok = BpeTok.load("gpt2_small")
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    for kind in ("zipf", "mixed", "uniform"):
        b, e, c = TextModel(99, kind).batch(300, 150)
        rb, re_ = ragged_rows(len(b))
        ob, oe, ids = orc(*rs(rb, re_, b, e, c)[:5])
        raw = c.tobytes()
        for i in range(len(b)):
            assert hf.encode(raw[b[i]:e[i]].decode(), add_special_tokens=False).ids == ids[ob[i]:oe[i]].tolist()
    assert orc.tie_events == 0


def test_oracle_matches_hf_golden_llama3():
    z = np.load(GOLDEN / "golden_bpe_llama3_small.npz")
    tok = BpeTok.load("llama3_small")
    rb, re_ = ragged_rows(len(z["begins"]))
    sp = O.RegexSplit(tok.pattern, "isolate")(rb, re_, z["begins"], z["ends"], z["chars"])
    ob, oe, ids = tok.oracle()(*sp[:5])
    assert np.array_equal(ob, z["id_begins"]) and np.array_equal(oe, z["id_ends"]) and np.array_equal(ids, z["ids"])


def test_llama3_chain(backend):
    """Llama-3-shaped tokenizer: RegexSplit -> BPETokenizer on the device vs the oracle chain and the HF golden ids;
    the fused entry point gives the same ids."""
    z = np.load(GOLDEN / "golden_bpe_llama3_small.npz")
    tok = BpeTok.load("llama3_small")
    rb, re_ = ragged_rows(len(z["begins"]))
    inputs = [rb, re_, z["begins"], z["ends"], z["chars"]]
    pat = tok.pattern_u8()
    sp_ref = O.RegexSplit(tok.pattern, "isolate")(*inputs)
    split = RegexSplit("isolate", lib=backend.lib)
    sp = split.evaluate(backend.data(inputs) + [pat])
    assert_same(sp_ref[:4], sp[:4], backend.host, "RegexSplit llama3")
    bpe = BPETokenizer(**tok.attrs, lib=backend.lib)
    got = bpe.evaluate(list(sp[:5]) + tok.consts)
    assert np.array_equal(backend.host(got[2]), z["ids"]) and np.array_equal(backend.host(got[1]), z["id_ends"])
    fused = FusedSplitBPE(split, bpe).evaluate(backend.data(inputs) + [pat], tok.consts)
    assert np.array_equal(backend.host(fused[2]), z["ids"]) and np.array_equal(backend.host(fused[1]), z["id_ends"])


# ------------------------------------------------------------------ kernels vs oracle
def test_llama3_fused(backend):
    """The Llama-3 pattern through the fused RegexSplit + BPETokenizer kernel (bit-parallel scanner) = the oracle."""
    tok = BpeTok.load("llama3_small")
    n = 40 if backend.name == "emu" else 4000
    b, e, c = TextModel(5, "mixed").batch(n, 300)
    strings = ["it's IT'S don'T we'LL 12345 6,789.10\r\n\r\n  end  ", "a\tb \n c\n\n  d", "x\u00a0y !\n\nz", "", "1234567 " * 90, " " * 700 + "x"]
    b2, e2, c2 = O.pack_strings(strings)
    b = np.concatenate([b, b2 + len(c)]).astype(np.int32)
    e = np.concatenate([e, e2 + len(c)]).astype(np.int32)
    c = np.concatenate([c, c2])
    rb, re_ = ragged_rows(len(b))
    ref = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(rb, re_, b, e, c)[:5])
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [tok.pattern_u8()], tok.consts), backend.host, "fused llama3")


@pytest.mark.parametrize("kind", ["mixed", "zipf"])
def test_llama3_rows_around_the_window_sizes(backend, kind):
    """Rows of 513 .. 768 bytes through lookup_rows_kernel's Llama-3 instance (the three-dwords-per-lane form of the scanner),
    shorter rows (two dwords per lane) and longer ones (left to the generic kernel) in the same batch.  More than 256 rows:
    not the one-launch small-batch kernel."""
    tok = BpeTok.load("llama3_small")
    n = 300 if backend.name == "emu" else 6000
    b, e, c = TextModel(11, kind).batch(n, 640)
    lens = e - b
    assert ((lens > 512) & (lens <= 768)).sum() > n // 2 and (lens <= 512).any() and (lens > 768).any()
    rb, re_ = ragged_rows(len(b))
    ref = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(rb, re_, b, e, c)[:5])
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    for _ in range(2):   # (the second call: memo and store warm)
        assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [tok.pattern_u8()], tok.consts), backend.host, "fused llama3, rows around 512 / 768 bytes")


def run_all_paths(backend, tok, inputs, skips=None, pattern=None):
    """Oracle result + the three product paths (split op, BPE op on its pieces, fused) compared bit for bit."""
    pattern = pattern or tok.pattern
    pat = np.frombuffer(pattern.encode(), np.uint8)
    o_in = [np.asarray(x) for x in inputs]
    sp_ref = O.RegexSplit(pattern, "isolate")(*o_in, skips=skips)
    ref = tok.oracle()(*sp_ref[:5])
    data = backend.data(inputs)
    sk = backend.data([np.asarray(skips, np.uint8)]) if skips is not None else []
    split = RegexSplit("isolate", lib=backend.lib)
    sp = split.evaluate(data + sk + [pat])
    assert_same(sp_ref[:4], sp[:4], backend.host, "RegexSplit")
    if skips is not None:
        assert_same([sp_ref[5]], [sp[5]], backend.host, "RegexSplit skips")
    bpe = BPETokenizer(**tok.attrs, lib=backend.lib)  # one handle (one memo build) serves the op and the fused path
    got = bpe.evaluate(list(sp[:5]) + tok.consts)
    assert_same(ref, got, backend.host, "BPETokenizer")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), bpe)
    got2 = fused.evaluate(data + sk + [pat], tok.consts)
    assert_same(ref, got2, backend.host, "fused")
    return ref


def test_golden_strings(backend):
    z = np.load(GOLDEN / "golden_bpe_gpt2_small.npz")
    rb, re_ = ragged_rows(len(z["begins"]))
    ref = run_all_paths(backend, BpeTok.load("gpt2_small"), [rb, re_, z["begins"], z["ends"], z["chars"]])
    assert np.array_equal(ref[2], z["ids"])  # and therefore equal to HF


@pytest.mark.parametrize("kind,n,target", [("zipf", 32, 128), ("mixed", 24, 300), ("uniform", 16, 200)])
def test_config1_shapes(backend, kind, n, target):
    """BASELINE.json config 1 (32 x ~128-byte strings) plus mixed-Unicode / stress text."""
    b, e, c = TextModel(5, kind).batch(n, target)
    rb, re_ = ragged_rows(n)
    run_all_paths(backend, BpeTok.load("gpt2_small"), [rb, re_, b, e, c])


def test_big_vocab(backend):
    """GPT-2-shaped tokenizer (V = 50 257, 50 000 merges)."""
    b, e, c = TextModel(6, "zipf").batch(16, 256)
    rb, re_ = ragged_rows(16)
    tok = BpeTok.load("gpt2")
    if backend.name == "emu":  # building the memo = the device BPE over all 50 257 tokens: a minute and a half on the emulator;
        tok.attrs = dict(tok.attrs, cache_capacity=0)  # the memo is covered there by the small vocabularies, here on the GPU
    run_all_paths(backend, tok, [rb, re_, b, e, c])


def test_digits_pattern(backend):
    tok = BpeTok.load("gpt2_small")
    strings = ["hello 123", "If I have 100 million dollars?", "a1b2c3", "test 0987654321 end", " 1 2  3", "x٣٤y ½"]
    run_all_paths(backend, tok, one_string_per_row(strings), pattern=DIGITS_PATTERN)


def test_ragged_layouts(backend):
    """Empty strings, empty rows, several strings per row, rows sharing strings, gaps and reversed order in chars."""
    tok = BpeTok.load("gpt2_small")
    texts = [b"first string here", b"", b"second, with 'quotes' & 42 numbers", b"third\n\nline", b"  ", b"tail"]
    # lay the strings out back to front with gaps
    chars = bytearray(b"#" * 200)
    begins, ends, pos = [], [], 190
    for t in texts:
        pos -= len(t) + 3
        chars[pos:pos + len(t)] = t
        begins.append(pos)
        ends.append(pos + len(t))
    rb = np.array([0, 2, 2, 5, 1, 0], np.int32)   # row 2 is empty, row 4 re-uses strings 1..2, row 5 = row 0
    re_ = np.array([2, 2, 5, 6, 3, 2], np.int32)
    run_all_paths(backend, tok, [rb, re_, np.array(begins, np.int32), np.array(ends, np.int32),
                                 np.frombuffer(bytes(chars), np.uint8)])


def test_skips_pass_through(backend):
    tok = BpeTok.load("gpt2_small")
    strings = [b"some text", b"<|endoftext|>", b" more text here", b"<|endoftext|>", b"x"]
    b, e, c = O.pack_strings(strings)
    rb, re_ = np.array([0, 3], np.int32), np.array([3, 5], np.int32)
    ref = run_all_paths(backend, tok, [rb, re_, b, e, c], skips=[0, 1, 0, 1, 0])
    assert (ref[2] == tok.added[b"<|endoftext|>"]).sum() == 2  # the special token reached BPE whole


def test_all_empty_batch_quirk(backend):
    """regex_split.cpp:129-143: zero chars -> ragged dims of shape {1}, whatever the batch size."""
    tok = BpeTok.load("gpt2_small")
    inputs = one_string_per_row(["", "", ""])
    pat = tok.pattern_u8()
    sp = RegexSplit("isolate", lib=backend.lib).evaluate(backend.data(inputs) + [pat])
    ref = O.RegexSplit(tok.pattern, "isolate")(*inputs)
    assert_same(ref[:4], sp[:4], backend.host, "RegexSplit empty batch")
    got = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib)).evaluate(
        backend.data(inputs) + [pat], tok.consts)
    assert_same(tok.oracle()(*ref[:5]), got, backend.host, "fused empty batch")


def test_long_strings_and_pieces(backend):
    """Multi-chunk strings; pieces that need the wave path (> 24 symbols) and the exact path (> 512 bytes)."""
    tok = BpeTok.load("gpt2_small")
    m = TextModel(8, "zipf")
    _, _, c = m.batch(4, 3000)
    long_text = c.tobytes().decode()
    strings = [long_text[:2500], "a" * 300, " " * 256 + "x", "ab" * 400, "z" * 2000 + " end", "0" * 700,
               long_text[2500:4000] + "q" * 530 + long_text[4000:4400], "é" * 400, "short"]
    run_all_paths(backend, tok, one_string_per_row(strings))


def test_max_splits(backend):
    strings = ["one two three four five", "a b", "single"]
    inputs = one_string_per_row(strings)
    pat = np.frombuffer(GPT2_PATTERN.encode(), np.uint8)
    for ms in (1, 2, 7):
        ref = O.RegexSplit(GPT2_PATTERN, "isolate", False, ms)(*inputs)
        got = RegexSplit("isolate", max_splits=ms, lib=backend.lib).evaluate(backend.data(inputs) + [pat])
        assert_same(ref[:4], got[:4], backend.host, f"max_splits={ms}")


# ------------------------------------------------------------------ hand-made vocabularies
def pieces_inputs(rows):
    """rows: list of lists of pieces -> the five ragged-string inputs of the BPETokenizer op."""
    flat = [p for r in rows for p in r]
    b, e, c = O.pack_strings(flat)
    counts = np.array([len(r) for r in rows])
    re_ = np.cumsum(counts).astype(np.int32)
    return [(re_ - counts).astype(np.int32), re_, b, e, c]


def test_piece_lengths_around_path_limits(backend):
    """merge_kernel's paths by symbol count: F (<= 16, lane per piece), L (17..32, lane per piece, 32 at a time), W (a wave
    per piece).  Random letter strings miss the memo; whole batches of one length, and batches that mix them so that a
    64-piece batch holds more than 32 path-L pieces (two groups) next to F and W pieces."""
    tok = BpeTok.load("gpt2_small")
    rng = np.random.default_rng(77)
    letters = list(b"abcdefghijklmnopqrstuvwxyz")
    per = 40 if backend.name == "emu" else 300
    def word(n):
        return bytes(rng.choice(letters, size=n).tolist())
    rows = [[word(n) for _ in range(per)] for n in (15, 16, 17, 18, 24, 31, 32, 33, 40)]
    rows.append([word(int(rng.integers(1, 45))) for _ in range(per * 6)])
    rows.append([word(int(rng.integers(17, 33))) for _ in range(200)])
    rows.append([b" " + word(31), b" " + word(32), "é".encode() * 8 + word(15), "元".encode() * 10, "元".encode() * 11])
    inputs = pieces_inputs(rows)
    ref = tok.oracle()(*inputs)
    got = BPETokenizer(**tok.attrs, lib=backend.lib).evaluate(backend.data(inputs) + tok.consts)
    assert_same(ref, got, backend.host, "path limits")


def test_heap_tie_vocabulary(backend):
    """(rank, seq) ties in the merge queue: the two pairs pushed by ONE merge tie when the new
    symbol and both neighbours carry the same id.  Reachable when an added token's id collides with a merge
    result's id ("Q" -> id of "ab"): the text Q a b Q merges a+b between two symbols that already are "ab".
    Which of the tied pairs pops first is decided by libstdc++'s heap layout; the oracle runs the real
    std::priority_queue, the device replays such pieces on its exact heap path (bpe_exact_piece)."""
    vocab = [b"a", b"b", b"ab", b"abab", b"ba", b"aa", b"ababab", b"aab", b"bab"]
    merges = [(b"a", b"b"), (b"ab", b"ab"), (b"b", b"a"), (b"a", b"a"), (b"abab", b"ab"), (b"aa", b"b"), (b"b", b"ab")]
    tok = BpeTok(vocab, merges, {b"Q": 2, b"R": 5}, None)
    rng = np.random.default_rng(1)
    rows = []
    for _ in range(60):
        rows.append([bytes(rng.choice(list(b"abQR"), size=int(rng.integers(1, 70)), p=[0.35, 0.3, 0.25, 0.1]))
                     for _ in range(int(rng.integers(0, 8)))])
    rows.append([b"QabQ", b"QabQabQabQabQ" * 20, b"RaaR" * 150])  # the last one exceeds the LDS chunk as well
    inputs = pieces_inputs(rows)
    orc = tok.oracle()
    ref = orc(*inputs)
    assert orc.tie_events > 50, "the vocabulary is meant to produce (rank, seq) ties"
    got = BPETokenizer(**tok.attrs, lib=backend.lib).evaluate(backend.data(inputs) + tok.consts)
    assert_same(ref, got, backend.host, "tie vocabulary")


@pytest.mark.parametrize("n_vocab", [65535, 65536, 65537])
def test_highest_ids_around_the_u16_limits(backend, n_vocab):
    """The library narrows what it may: 16-bit staging entries while every id is below 65535 (0xFFFF marks an unused entry),
    16-bit ids in merge_kernel's LDS while every id is below 65536.  Vocabularies whose LAST tokens -- ids 65534, 65535,
    65536 -- are what the text merges to."""
    if backend.name == "emu":
        pytest.skip("a 65 536-token vocabulary: the emulator takes five minutes per handle to fill its memo (it passes there)")
    filler = [b"\x01f%05d" % i for i in range(n_vocab - 6)]
    vocab = [b"a", b"b", b"c"] + filler + [b"ab", b"abc", b"ca"]          # ids n-3, n-2, n-1
    merges = [(b"a", b"b"), (b"ab", b"c"), (b"c", b"a")]
    assert len(vocab) == n_vocab
    tok = BpeTok(vocab, merges, None, None)
    rows = [[b"abc", b"ab", b"ca", b"cab", b"abcabc" * 40], [b"c" * 3, b"ababab"], [], [b"caca" * 100]]
    rows += [[b"abcab" * int(k) for k in range(1, 40)] for _ in range(70)]   # (more than 256 rows in all: the ordinary launches too)
    rows += [[b"ab"]] * 300
    inputs = pieces_inputs(rows)
    ref = tok.oracle()(*inputs)
    assert int(np.max(ref[2])) == n_vocab - 1
    bpe = BPETokenizer(**tok.attrs, lib=backend.lib)
    for _ in range(2):
        assert_same(ref, bpe.evaluate(backend.data(inputs) + tok.consts), backend.host, f"V = {n_vocab}")


def test_unk_byte_fallback_suffix(backend):
    """Non-byte-level vocabulary: unknown bytes -> <0xHH> byte tokens, then unk, else dropped; end_suffix appended
    to every piece (also to empty ones); text-form merges ("a b" lines, 11-input form)."""
    base = [b"<unk>", b"a", b"b", b"c", b"d", b"</w>", b"<0x65>", b"<0x7A>", "é".encode(), b"ab", b"abc", b"c</w>",
            b"abc</w>", b"d</w>"]
    merges_text = [b"a b", b"ab c", b"c </w>", b"ab c</w>", b"d </w>"]
    for attrs in (dict(unk_token="<unk>", byte_fallback=True, end_suffix="</w>"),
                  dict(unk_token="<unk>", byte_fallback=False, end_suffix="</w>"),
                  dict(unk_token="", byte_fallback=True, end_suffix=""),
                  dict(unk_token="", byte_fallback=False, end_suffix="", fuse_unk=True)):
        tok = BpeTok(base, merges_text, None, None, **attrs)
        rows = [[b"abc", b"abcd", b"", b"zzz", b"e", "é".encode(), b"xyz"], [], [b"dabc" * 5, b"q"], [b""]]
        inputs = pieces_inputs(rows)
        cap = 256  # with an end_suffix an empty piece still yields tokens: size the ids buffer explicitly
        ref = tok.oracle()(*inputs, cap=cap)
        got = BPETokenizer(**tok.attrs, lib=backend.lib).evaluate(backend.data(inputs) + tok.consts, ids_capacity=cap)
        assert_same(ref, got, backend.host, f"attrs {attrs}")


def test_errors(backend):
    tok = BpeTok.load("gpt2_small")
    lib = backend.lib
    with pytest.raises(L.OvtkError) as ei:  # unknown behaviour: OPENVINO_ASSERT in the reference (regex_split.cpp:113)
        RegexSplit("sideways", lib=lib).evaluate(backend.data(one_string_per_row(["x"])) + [tok.pattern_u8()])
    assert ei.value.code == L.E_ARG
    with pytest.raises(L.OvtkError) as ei:
        RegexSplit("isolate", max_splits=0, lib=lib).evaluate(backend.data(one_string_per_row(["x"])) + [tok.pattern_u8()])
    assert ei.value.code == L.E_ARG
    with pytest.raises(L.OvtkError) as ei:  # a pattern outside the compiled subset must fail loudly, not fall back
        RegexSplit("isolate", lib=lib).evaluate(backend.data(one_string_per_row(["x"])) + [np.frombuffer(rb"([a-z])\1+", np.uint8)])
    assert ei.value.code == L.E_UNSUPPORTED
    bad = BpeTok([b"a", b"b"], [(b"a", b"c")], None, None)  # merge token missing: std::out_of_range in the reference
    with pytest.raises(L.OvtkError) as ei:
        BPETokenizer(lib=lib).evaluate(backend.data(one_string_per_row(["ab"])) + bad.consts)
    assert ei.value.code == L.E_VOCAB
    with pytest.raises(L.OvtkError) as ei:  # wrong input count (bpe_tokenizer.cpp:18-21)
        BPETokenizer(lib=lib).evaluate(one_string_per_row(["ab"]) + tok.consts[:4])
    assert ei.value.code == L.E_ARG
    rb, re_, b, e, c = one_string_per_row(["abc"])
    with pytest.raises(L.OvtkError) as ei:  # offsets leaving the chars tensor
        BPETokenizer(**tok.attrs, lib=lib).evaluate(backend.data([rb, re_, b, e + 100, c]) + tok.consts)
    assert ei.value.code == L.E_RANGE


def test_handle_reuse_and_determinism(backend):
    """One handle, many calls of different sizes (workspace growth), identical results every time."""
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    for n, target in [(4, 64), (40, 400), (2, 16), (40, 400)]:
        b, e, c = TextModel(n, "zipf").batch(n, target)
        rb, re_ = ragged_rows(n)
        ref = orc(*rs(rb, re_, b, e, c)[:5])
        for _ in range(2):
            assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host)


def test_enqueue_finish_overlapped(gpu_backend):
    """ovtk_encode_enqueue / ovtk_encode_finish: three batches launched back to back, finished afterwards in order --
    each equals the oracle, workspaces do not bleed into each other (every call in flight leases its own)."""
    backend = gpu_backend
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    batches, refs, tickets = [], [], []
    for i, (n, target, kind) in enumerate([(3000, 300, "zipf"), (500, 900, "mixed"), (3000, 120, "uniform")]):
        b, e, c = TextModel(n + i, kind).batch(n, target)
        rb, re_ = ragged_rows(n)
        refs.append(orc(*rs(rb, re_, b, e, c)[:5]))
        batches.append(backend.data([rb, re_, b, e, c]))
    for data in batches:
        tickets.append(fused.enqueue(data + [pat], tok.consts))
    for ref, ticket in zip(refs, tickets):
        assert_same(ref, ticket(), backend.host, "enqueue/finish")
    with pytest.raises(L.OvtkError):   # host arrays have no asynchronous form
        fused.enqueue([np.asarray(x.cpu()) for x in batches[0]] + [pat], tok.consts)


@pytest.mark.gpu
def test_enqueue_on_alternating_streams(gpu_backend):
    """Consecutive batches on two HIP streams (bench.py's host loop): their kernels share the CUs, every batch still
    equals the oracle."""
    import torch
    backend = gpu_backend
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    batches, refs, tickets = [], [], []
    for i, (n, target, kind) in enumerate([(20000, 300, "zipf"), (4000, 900, "mixed"), (20000, 120, "uniform"), (9000, 500, "zipf")]):
        b, e, c = TextModel(40 + i, kind).batch(n, target)
        rb, re_ = ragged_rows(n)
        refs.append(orc(*rs(rb, re_, b, e, c)[:5]))
        batches.append(backend.data([rb, re_, b, e, c]))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rounds in range(2):   # the second round runs on warm memo tables and leased workspaces
        tickets = []
        for k, data in enumerate(batches):
            with torch.cuda.stream(streams[k % 2]):
                tickets.append(fused.enqueue(data + [pat], tok.consts))
        for ref, ticket in zip(refs, tickets):
            assert_same(ref, ticket(), backend.host, "two streams")


@pytest.mark.parametrize("rows_per_ticket", [1, 2, 5])
def test_row_tickets_same_result(backend, rows_per_ticket):
    """ovtk_set_row_tickets: rows handed out dynamically (for a GPU shared with a collective) -- identical output."""
    tok = BpeTok.load("gpt2_small")
    n = 70 if backend.name == "emu" else 20000
    b, e, c = TextModel(99, "mixed").batch(n, 200)
    rb, re_ = ragged_rows(n)
    ref = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(rb, re_, b, e, c)[:5])
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    L.check(backend.lib, backend.lib.ovtk_set_row_tickets(rows_per_ticket))
    try:
        for _ in range(2):
            assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [tok.pattern_u8()], tok.consts), backend.host)
    finally:
        L.check(backend.lib, backend.lib.ovtk_set_row_tickets(0))
    with pytest.raises(L.OvtkError):
        L.check(backend.lib, backend.lib.ovtk_set_row_tickets(-1))


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_unaligned_chars_tensor(backend, shift):
    """The chars tensor itself starts at an address that is not a multiple of 4, and strings touch both of its ends:
    the window staging fetches whole dwords only inside the tensor."""
    tok = BpeTok.load("gpt2_small")
    strings = ["it's", "a b", "x" * 700 + " y's", "", "tail isn't"]
    b, e, c = O.pack_strings(strings)
    rb, re_ = ragged_rows(len(strings))
    ref = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(rb, re_, b, e, c)[:5])
    padded = np.concatenate([np.full(shift, 0x41, np.uint8), c])
    if backend.name == "hip-device":
        import torch
        chars = torch.as_tensor(padded, device="cuda")[shift:]
        assert chars.data_ptr() % 4 == shift
        data = backend.data([rb, re_, b, e]) + [chars]
    else:
        chars = padded[shift:]
        data = [rb, re_, b, e, chars]
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    assert_same(ref, fused.evaluate(data + [tok.pattern_u8()], tok.consts), backend.host, "unaligned tensor")


@pytest.mark.parametrize("mem", ["device", "host"])
def test_many_exact_pieces_leave_no_stray_writes(backend, mem):
    """More exact-path pieces (> 512 bytes) than merge_kernel's folded tail takes (256): that attempt sets
    kFlagTailPending and the host repeats it with the separate exact / count_scan launches.  compact_kernel of the FIRST
    attempt must not run on the stale tile offsets an earlier call left in the pooled workspace: a sentinel region behind
    the ids buffer (and behind begins/ends) stays untouched, and the result equals the oracle."""
    import ctypes as C
    if (mem == "device") != (backend.name != "hip-host"):
        pytest.skip("one memory kind per backend")
    tok = BpeTok.load("gpt2_small")
    bpe = BPETokenizer(**tok.attrs, lib=backend.lib)
    # call A: 3 tiles of long rows -> tile offsets beyond call B's whole ids buffer stay behind in the workspace
    ba, ea, ca = TextModel(3, "zipf").batch(130, 8000)
    rba, rea = ragged_rows(130)
    sp = O.RegexSplit(tok.pattern, "isolate")(rba, rea, ba, ea, ca)
    bpe.evaluate(backend.data(list(sp[:5])) + tok.consts)
    # call B: 320 rows, one 600-byte piece each
    strings = [chr(ord("a") + i % 26) * 600 for i in range(320)]
    rb, re_, b, e, c = one_string_per_row(strings)
    ref = tok.oracle()(rb, re_, b, e, c)
    cap, guard = len(c), 200000
    SENT = -7777777
    if backend.name == "hip-device":
        import torch
        mk = lambda a: torch.as_tensor(a, device="cuda")  # noqa: E731
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        host = lambda t: t.cpu().numpy()  # noqa: E731
    else:
        mk = lambda a: np.ascontiguousarray(a)  # noqa: E731
        ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        host = lambda a: a  # noqa: E731
    d = [mk(x) for x in (rb, re_, b, e, c)]
    ob = mk(np.full(len(rb) + guard, SENT, np.int32))
    oe = mk(np.full(len(rb) + guard, SENT, np.int32))
    ids = mk(np.full(cap + guard, SENT, np.int32))
    rs = L.RaggedStrings(ptr(d[0]), ptr(d[1]), len(rb), L.Strings(ptr(d[2]), ptr(d[3]), ptr(d[4]), len(b), len(c)))
    out = L.RaggedI32Out(ptr(ob), ptr(oe), ptr(ids), cap, 0, 0)
    memk = L.MEM_HOST if backend.name == "hip-host" else L.MEM_DEVICE
    L.check(backend.lib, backend.lib.ovtk_bpe_run(bpe._h, C.byref(rs), C.byref(out), memk, None))
    if backend.name == "hip-device":
        torch.cuda.synchronize()
    ob, oe, ids = host(ob), host(oe), host(ids)
    assert out.n_data == len(ref[2])
    assert np.array_equal(ob[: len(rb)], ref[0]) and np.array_equal(oe[: len(rb)], ref[1])
    assert np.array_equal(ids[: out.n_data], ref[2])
    assert (ids[out.n_data:] == SENT).all(), "ids written beyond the result"
    assert (ob[len(rb):] == SENT).all() and (oe[len(rb):] == SENT).all()


def test_enqueue_host_buffers(backend):
    """ovtk_encode_enqueue_host / ovtk_encode_finish: host buffers in, host buffers out, several batches in flight (on
    the GPU box: pinned buffers on two HIP streams, so that copies and kernels of neighbouring batches overlap)."""
    if backend.name == "hip-host":
        pytest.skip("covered by the other two backends")
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    gpu = backend.name != "emu"
    streams = [None, None]
    if gpu:
        import torch
        side = torch.cuda.Stream()
        streams = [torch.cuda.current_stream().cuda_stream, side.cuda_stream]

    def pinned(a):
        if not gpu:
            return a
        t = torch.empty(a.shape, dtype=getattr(torch, str(a.dtype)), pin_memory=True)
        v = t.numpy()
        v[...] = a
        return v

    refs, tickets = [], []
    for i, (n, target, kind) in enumerate([(40, 300, "zipf"), (12, 900, "mixed"), (60, 120, "uniform"), (3, 20, "zipf")]):
        if gpu:
            n *= 50
        b, e, c = TextModel(n + i, kind).batch(n, target)
        rb, re_ = ragged_rows(n)
        refs.append(orc(*rs(rb, re_, b, e, c)[:5]))
        outs = tuple(pinned(np.zeros(k, np.int32)) for k in (n, n, len(c)))
        tickets.append(fused.enqueue_host([pinned(x) for x in (rb, re_, b, e, c)] + [pat], tok.consts, outs, streams[i % 2]))
    for ref, ticket in zip(refs, tickets):
        assert_same(ref, ticket(), lambda x: x, "enqueue_host/finish")


@pytest.mark.parametrize("capacity", [20000, 7, 0])
def test_memo_learns_and_results_stay(backend, capacity):
    """The dynamic part of the piece memo (the reference's m_cache, bpe_tokenizer.cpp:197-205 / :331-338): pieces of
    several tokens are kept the first time merge_kernel computes them, up to cache_capacity; the results of every call --
    before, while and after learning, and with the table's room exhausted -- equal the oracle's."""
    import ctypes as C
    tok = BpeTok.load("gpt2_small")
    attrs = dict(tok.attrs, cache_capacity=capacity)
    bpe = BPETokenizer(**attrs, memo_learn=-1, lib=backend.lib)   # (memo_learn < 0: the first level learns cache_capacity pieces, the reference's count)
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), bpe)
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    lib = backend.lib
    learned = []
    n = 48 if backend.name == "emu" else 600
    for k in range(4):
        b, e, c = TextModel(70 + (k & 1), "zipf").batch(n, 300)   # batches 2, 3 repeat 0, 1: their pieces are known by then
        rb, re_ = ragged_rows(n)
        ref = orc(*rs(rb, re_, b, e, c)[:5])
        assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host)
        fixed, got = C.c_int64(), C.c_int64()
        L.check(lib, lib.ovtk_bpe_memo_entries(bpe._h, C.byref(fixed), C.byref(got)))
        learned.append(int(got.value))
        assert (fixed.value > 0) == (capacity != 0)
    assert all(0 <= x <= capacity for x in learned)
    if capacity == 0:
        assert learned == [0, 0, 0, 0]
    elif capacity == 7:
        assert learned[0] >= 5 and learned[-1] <= 7   # (a piece whose two slots are taken is not kept: 7 need not be reached)
    else:
        # the repeats add next to nothing (only pieces that lost a race for a slot the first time round, on the GPU)
        assert learned[0] > 20 and learned[1] > learned[0] and learned[1] <= learned[3] <= learned[1] * 1.15


@pytest.mark.parametrize("wide", [False, True])
def test_memo_learn_default_and_six_id_entries(backend, wide):
    """ovtk_bpe_params::memo_learn = 0 (the default): the first level learns up to max(cache_capacity, the store's capacity) pieces,
    and -- every id below 65 535 -- pieces of up to SIX ids (PieceTableDev::packed6, read by lookup_span_kernel); `wide` (an added
    token beyond 65 535): three i32 ids as ever.  cache_capacity = 7 here, so whatever is learned beyond 7 is the default's doing.
    Every call equals the oracle: first sight, and the repeats that run on what was learned -- words of four to six tokens among
    them (the vocabulary is small: most longer words are)."""
    import ctypes as C
    from tools.make_tokenizers import load_tokenizer
    lib = backend.lib
    t = load_tokenizer("gpt2_small")
    added = dict(t["added"])
    if wide:
        added[b"<|far|>"] = 70000
    tok = BpeTok(t["vocab"], t["merges"], added, t["pattern"], **dict(t["attrs"], cache_capacity=7))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    n = 300 if backend.name == "emu" else 900   # (more than 256 rows: lookup_span_kernel)
    b, e, c = TextModel(83, "zipf").batch(n, 200)
    rb, re_ = ragged_rows(n)
    ref = orc(*rs(rb, re_, b, e, c)[:5])
    bpe = BPETokenizer(**tok.attrs, lib=lib)
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), bpe)
    learned = []
    for rep in range(3):
        assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host, f"memo_learn default, call {rep}")
        fixed, got = C.c_int64(), C.c_int64()
        L.check(lib, lib.ovtk_bpe_memo_entries(bpe._h, C.byref(fixed), C.byref(got)))
        learned.append(int(got.value))
    assert learned[0] > 7 and learned[0] <= learned[1] <= learned[2] <= learned[0] * 1.15 + 2
    # an explicit count: memo_learn = 12 pieces, whatever cache_capacity and the store say
    bpe12 = BPETokenizer(**tok.attrs, memo_learn=12, lib=lib)
    fused12 = FusedSplitBPE(RegexSplit("isolate", lib=lib), bpe12)
    for rep in range(2):
        assert_same(ref, fused12.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host, f"memo_learn 12, call {rep}")
    fixed, got = C.c_int64(), C.c_int64()
    L.check(lib, lib.ovtk_bpe_memo_entries(bpe12._h, C.byref(fixed), C.byref(got)))
    assert 8 <= got.value <= 12


@pytest.mark.parametrize("wide", [False, True])
def test_piece_store_holds_what_was_merged(backend, wide):
    """The memo's second level (tables.hpp "piece store"): what merge_kernel had to merge -- pieces up to 31 bytes, up to 15 ids
    (7 when ids need more than 16 bits: `wide`, a vocabulary with an added token beyond 65535) -- is found there by the next
    call.  cache_capacity = 7 keeps the first level from learning, so the repeats live off the store; every call equals the
    oracle; with the store switched off (ovtk_set_memo_store(0)) a new handle works as before."""
    import ctypes as C
    from tools.make_tokenizers import load_tokenizer
    lib = backend.lib
    t = load_tokenizer("gpt2_small")
    added = dict(t["added"])
    if wide:
        added[b"<|far|>"] = 70000   # (an added token's id is just a number: bpe_tokenizer.cpp:110-114 -- the vocabulary stays small)
    tok = BpeTok(t["vocab"], t["merges"], added, t["pattern"], **dict(t["attrs"], cache_capacity=7))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    n = 24 if backend.name == "emu" else 800
    batches = []
    for k, kind in enumerate(("zipf", "mixed")):   # "mixed": non-Latin words of 16..31 bytes (the long-piece path)
        b, e, c = TextModel(80 + k, kind).batch(n, 260)
        rb, re_ = ragged_rows(n)
        batches.append(([rb, re_, b, e, c], orc(*rs(rb, re_, b, e, c)[:5])))
    for entries in (4096, 0):
        L.check(lib, lib.ovtk_set_memo_store(C.c_int64(entries)))
        try:
            bpe = BPETokenizer(**tok.attrs, memo_learn=-1, lib=lib)   # (the first level keeps to cache_capacity = 7)
            fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), bpe)
            counts = []
            for rep in range(3):
                for data, ref in batches:
                    assert_same(ref, fused.evaluate(backend.data(data) + [pat], tok.consts), backend.host, f"store {entries} rep {rep}")
                stored, cap = C.c_int64(), C.c_int64()
                L.check(lib, lib.ovtk_bpe_store_entries(bpe._h, C.byref(stored), C.byref(cap)))
                counts.append(int(stored.value))
            if entries:
                assert cap.value > 0 and counts[0] > 20 and counts[0] <= counts[1] <= counts[2] <= counts[0] * 1.2 + 2
            else:
                assert cap.value == 0 and counts == [0, 0, 0]
        finally:
            L.check(lib, lib.ovtk_set_memo_store(C.c_int64(1048576)))


@pytest.mark.gpu
def test_pinned_outputs_written_by_the_kernels(hip_lib):
    """Host-memory calls with PINNED output buffers take no D2H copy: compact_kernel (or encode_small_kernel) stores through
    the buffers' device-side addresses.  Same results as with pageable buffers (staged + copied), mixed cases too; a too
    small pinned ids buffer is OVTK_E_CAPACITY with nothing written past its end; a workspace that has to grow (the call is
    repeated inside finish) ends in the same place."""
    import torch
    tok = BpeTok.load("gpt2_small")
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()

    def pinned(a):
        t = torch.empty(a.shape, dtype=getattr(torch, str(a.dtype)), pin_memory=True)
        v = t.numpy()
        v[...] = a
        return v
    for n, target, kind in [(3000, 300, "zipf"), (20, 100, "zipf"), (2500, 200, "uniform")]:   # (20 rows: the one-launch path)
        fused = FusedSplitBPE(RegexSplit("isolate", lib=hip_lib), BPETokenizer(**tok.attrs, lib=hip_lib))   # fresh workspace sizes
        b, e, c = TextModel(n, kind).batch(n, target)
        rb, re_ = ragged_rows(n)
        ref = orc(*rs(rb, re_, b, e, c)[:5])
        ins = [rb, re_, b, e, c]
        for pin_out in ((True, True, True), (False, False, False), (True, False, True)):
            outs = tuple((pinned if p else (lambda a: a))(np.full(k, -7, np.int32)) for p, k in zip(pin_out, (n, n, len(c) + 64)))
            got = fused.enqueue_host([pinned(x) for x in ins] + [pat], tok.consts, outs)()
            assert_same(ref, got, lambda x: x, f"pinned outputs {pin_out}")
            assert np.all(outs[2][len(ref[2]):] == -7)   # nothing behind the ids
        short = tuple(pinned(np.full(k, -7, np.int32)) for k in (n, n, len(ref[2]) // 2 + 64))
        guard = short[2][len(ref[2]) // 2:]
        with pytest.raises(L.OvtkError) as ei:
            fused.enqueue_host([pinned(x) for x in ins] + [pat], tok.consts, (short[0], short[1], short[2][: len(ref[2]) // 2]))()
        assert ei.value.code == L.E_CAPACITY and np.all(guard == -7)


@pytest.mark.gpu
@pytest.mark.parametrize("vocab", ["gpt2_small", "llama3_small"])
def test_memo_learns_under_concurrent_lookups(gpu_backend, vocab):
    """One handle on three HIP streams, twelve different batches in flight three at a time: merge_kernel of one batch
    inserts memo entries while the lookup kernels of its neighbours probe the same table (tables.hpp, kPieceBusy / the
    payload tag).  Every batch equals the oracle, the first time and again with the memo as the race left it."""
    import ctypes as C
    import torch
    backend = gpu_backend
    tok = BpeTok.load(vocab)
    bpe = BPETokenizer(**dict(tok.attrs, cache_capacity=200000), lib=backend.lib)   # room for everything: inserts all the way
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), bpe)
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    batches, refs = [], []
    for i in range(12):
        kind = ("zipf", "mixed", "uniform")[i % 3]
        n = 3000 + 500 * (i % 4)
        b, e, c = TextModel(900 + i, kind).batch(n, 200 + 40 * (i % 5))
        rb, re_ = ragged_rows(n)
        refs.append(orc(*rs(rb, re_, b, e, c)[:5]))
        batches.append(backend.data([rb, re_, b, e, c]))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    learned = []
    for rounds in range(3):
        inflight = []
        for k, data in enumerate(batches):
            with torch.cuda.stream(streams[k % 3]):
                inflight.append((k, fused.enqueue(data + [pat], tok.consts)))
            if len(inflight) > 2:
                j, t = inflight.pop(0)
                assert_same(refs[j], t(), backend.host, f"round {rounds} batch {j}")
        for j, t in inflight:
            assert_same(refs[j], t(), backend.host, f"round {rounds} batch {j}")
        fixed, got = C.c_int64(), C.c_int64()
        L.check(backend.lib, backend.lib.ovtk_bpe_memo_entries(bpe._h, C.byref(fixed), C.byref(got)))
        learned.append(int(got.value))
    assert learned[0] > 1000 and learned[0] <= learned[1] <= learned[2] <= 200000


def test_fuzzed_tables(backend):
    """A fixed sample of tools/fuzz_bpe.py: random (untrained) vocabularies and merge tables -- duplicate strings, added
    tokens whose ids collide, missing bytes with and without unk / byte_fallback, end_suffix, text-form merges -- and pieces
    of every length class, twice per handle (the second call on what the memo learned)."""
    from tools import fuzz_bpe as F
    rng = np.random.default_rng(77)
    for k in range(25 if backend.name == "emu" else 120):
        tok, rows = F.case(rng)
        inputs = F.pieces_inputs(rows)
        cap = int(sum(len(p) + 8 for r in rows for p in r) * 2 + 64)
        ref = tok.oracle()(*inputs, cap=cap)
        op = BPETokenizer(**tok.attrs, lib=backend.lib)
        for rep in range(2):
            got = op.evaluate(backend.data(inputs) + tok.consts, ids_capacity=cap)
            assert_same(ref, got, backend.host, f"case {k} rep {rep} attrs {tok.attrs}")


def test_pending_rows_after_a_call_without_any(backend):
    """lookup_rows_kernel leaves the rows that are not one ASCII window to the generic kernel: an all-ASCII batch, then batches
    full of non-ASCII rows and multi-string rows on the SAME handle, then ASCII again -- every call equals the oracle."""
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    n = 40 if backend.name == "emu" else 2000
    for k, kind in enumerate(("zipf", "mixed", "mixed", "zipf", "mixed")):
        b, e, c = TextModel(90 + k, kind).batch(n, 200)
        rb, re_ = ragged_rows(n)
        if k == 2:   # rows of two strings, and one string longer than a scan window
            rb = np.arange(0, n, 2, dtype=np.int32)
            re_ = np.minimum(rb + 2, n).astype(np.int32)
        ref = orc(*rs(rb, re_, b, e, c)[:5])
        assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host, f"call {k} ({kind})")


def test_store_pauses_on_text_it_cannot_help(gpu_backend):
    """Uniform-random text: every piece is new, the piece store is probed and filled for nothing.  After four such calls in a
    row the handle leaves the store out for a while (api_encode.cpp store_pause); results never depend on that -- every call,
    before, during and after the pause, equals the oracle."""
    backend = gpu_backend
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    orc, rs = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    pat = tok.pattern_u8()
    n = 2500
    for k, kind in enumerate(["uniform"] * 7 + ["zipf", "uniform", "zipf"]):
        b, e, c = TextModel(300 + k, kind).batch(n, 220)
        rb, re_ = ragged_rows(n)
        ref = orc(*rs(rb, re_, b, e, c)[:5])
        assert_same(ref, fused.evaluate(backend.data([rb, re_, b, e, c]) + [pat], tok.consts), backend.host, f"call {k} ({kind})")
