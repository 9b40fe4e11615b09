"""BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle needs minutes for
these batches): lossless encode -> decode round trip, independence of how the rows are sharded ("linearity": the
result of a batch is the concatenation of the results of its halves), determinism, offsets that are ascending and
gap-free, agreement of the fused paths with the op-by-op chains, a byte checksum of the detokenizer output against the
per-token checksums -- plus the oracle on a prefix of every batch."""
import numpy as np
import pytest

from openvino_tokenizers_amd.ops import (BPETokenizer, ByteFallback, FusedDetokenizer, FusedSplitBPE, FusedSplitWordpiece,
                                         FuzeRagged, RaggedToDense, RegexSplit, VocabDecoder, WordpieceTokenizer)
from oracle import oracle as O
from tests.util import BpeTok
from tools.harness import pack_strings
from tools.make_tokenizers import load_tokenizer
from tools.workloads import TextModel, ragged_rows

pytestmark = pytest.mark.gpu


def dev(arrays):
    import torch
    return [torch.as_tensor(a, device="cuda") for a in arrays]


def check_offsets(begins, ends, n_ids):
    import torch
    assert int(begins[0]) == 0 and int(ends[-1]) == n_ids
    assert bool(torch.all(ends[:-1] == begins[1:])) and bool(torch.all(ends >= begins))


def halves_equal_whole(run, rb, re_, b, e, c, whole):
    """Rows [0, h) and [h, n) encoded on their own give the whole batch's ids back to back."""
    import torch
    n = len(rb)
    h = n // 2 + 17
    parts = []
    for lo, hi in ((0, h), (h, n)):
        rb2 = (rb[lo:hi] - rb[lo]).astype(np.int32)
        re2 = (re_[lo:hi] - rb[lo]).astype(np.int32)
        s0, s1 = int(rb[lo]), int(re_[hi - 1])
        parts.append(run(rb2, re2, b[s0:s1], e[s0:s1], c))
    ids = torch.cat([p[2] for p in parts])
    ends = torch.cat([parts[0][1], parts[1][1] + parts[0][2].numel()])
    assert torch.equal(ids, whole[2]) and torch.equal(ends, whole[1])


def decode_rows(tok_vocab, begins, ends, ids, pad, lib):
    """Ragged ids -> RaggedToDense (pad id, skipped) -> fused VocabDecoder + FuzeRagged: one string per row."""
    import torch
    assert not bool((ids == pad).any()), "the pad id occurs in the encoding"
    width = int((ends - begins).max())
    dense, _ = RaggedToDense(lib=lib).evaluate([begins, ends, ids, np.int32(width), np.int32(pad)])
    dec = VocabDecoder(skip_tokens=[pad], lib=lib)
    return FusedDetokenizer(dec).evaluate([dense] + list(pack_strings(tok_vocab)))


def same_text(out, b, e, c):
    import torch
    ob, oe, oc = out
    lens = torch.as_tensor(e - b, device="cuda")
    assert torch.equal(oe - ob, lens), "decoded row lengths differ from the input"
    assert int(b[0]) == 0 and np.array_equal(e[:-1], b[1:]), "test text is laid out back to back"
    assert torch.equal(oc, torch.as_tensor(c, device="cuda")), "decode(encode(text)) != text"


@pytest.mark.parametrize("name, kind, rows", [("gpt2", "zipf", 65536), ("llama3", "mixed", 131072)])
def test_bpe_full_size(hip_lib, name, kind, rows):
    """Config 2 (GPT-2-shaped) and one config-4 shard (Llama-3-shaped, V = 128 256): the fused kernels bench.py times
    (lookup_kernel<kFused> / <kFusedLlama3>), checked against the op-by-op chain, the halves, the decode round trip and
    the oracle."""
    import torch
    tok = BpeTok.load(name)
    b, e, c = TextModel(1234, kind).batch(rows, 512, seed=77)
    rb, re_ = ragged_rows(rows)
    pat = tok.pattern_u8()
    split = RegexSplit("isolate", lib=hip_lib)
    bpe = BPETokenizer(**tok.attrs, lib=hip_lib)
    fused = FusedSplitBPE(split, bpe)

    def chain(rb_, re2, b_, e_, c_):
        sp = split.evaluate(dev([rb_, re2, b_, e_]) + [c_, pat])
        return bpe.evaluate(list(sp[:5]) + tok.consts)

    def run(rb_, re2, b_, e_, c_):
        return fused.evaluate(dev([rb_, re2, b_, e_]) + [c_, pat], tok.consts)

    d_c = torch.as_tensor(c, device="cuda")
    whole = run(rb, re_, b, e, d_c)
    n_ids = whole[2].numel()
    check_offsets(whole[0], whole[1], n_ids)
    again = run(rb, re_, b, e, d_c)
    assert all(torch.equal(x, y) for x, y in zip(whole, again)), "two runs differ"
    ch = chain(rb, re_, b, e, d_c)
    assert all(torch.equal(x, y) for x, y in zip(whole, ch)), "fused encode differs from RegexSplit -> BPETokenizer"
    halves_equal_whole(run, rb, re_, b, e, d_c, whole)
    # byte-level BPE is lossless: the ids decode to exactly the input bytes
    pad = len(tok.vocab) - 1
    same_text(decode_rows(tok.vocab, whole[0], whole[1], whole[2], pad, hip_lib), b, e, c)
    # the oracle on the first rows
    k = 1500
    ref = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(rb[:k], re_[:k], b[:k], e[:k], c)[:5])
    assert np.array_equal(ref[1], whole[1][:k].cpu().numpy()) and np.array_equal(ref[2], whole[2][: int(ref[1][-1])].cpu().numpy())


def test_wordpiece_full_size(hip_lib):
    """Config 3: BERT-shaped WordPiece, 65 536 x ~256-byte rows."""
    import torch
    tok = load_tokenizer("bert")
    rows = 65536
    b, e, c = TextModel(1234, "zipf").batch(rows, 256, seed=78)
    c = np.frombuffer(c.tobytes().lower(), np.uint8).copy()
    rb, re_ = ragged_rows(rows)
    from tools.harness import BERT_PUNCT, BERT_WS
    ws = RegexSplit("remove", lib=hip_lib)
    pu = RegexSplit("isolate", lib=hip_lib)
    wp = WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=hip_lib)
    consts = list(pack_strings(tok["vocab"])) + [np.asarray(tok["unk_id"], np.int32)]
    fused = FusedSplitWordpiece(ws, pu, wp)
    u8 = lambda s: np.frombuffer(s.encode(), np.uint8)  # noqa: E731

    def run(rb_, re2, b_, e_, c_):
        return fused.evaluate(dev([rb_, re2, b_, e_]) + [c_], u8(BERT_WS), u8(BERT_PUNCT), consts)

    d_c = torch.as_tensor(c, device="cuda")
    whole = run(rb, re_, b, e, d_c)
    check_offsets(whole[0], whole[1], whole[2].numel())
    s1 = ws.evaluate(dev([rb, re_, b, e]) + [d_c, u8(BERT_WS)])
    s2 = pu.evaluate(list(s1[:5]) + [u8(BERT_PUNCT)])
    ch = wp.evaluate(list(s2[:5]) + consts)
    assert all(torch.equal(x, y) for x, y in zip(whole, ch)), "fused WordPiece differs from the three-op chain"
    halves_equal_whole(run, rb, re_, b, e, d_c, whole)
    k = 1500
    o1 = O.RegexSplit(BERT_WS, "remove")(rb[:k], re_[:k], b[:k], e[:k], c)
    o2 = O.RegexSplit(BERT_PUNCT, "isolate")(*o1[:5])
    ref = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*o2[:5], tok["unk_id"])
    assert np.array_equal(ref[1], whole[1][:k].cpu().numpy()) and np.array_equal(ref[2], whole[2][: int(ref[1][-1])].cpu().numpy())


def test_detokenize_full_size(hip_lib):
    """Config 5 chunk: 16 384 x 2 048 ids (< 2^31 output bytes per call)."""
    import torch
    tok = BpeTok.load("gpt2")
    rows, S, V = 16384, 2048, len(tok.vocab)
    rng = np.random.default_rng(5)
    ids = rng.integers(0, V - 1, size=(rows, S), dtype=np.int32)
    pad = V - 1
    ids[rng.random((rows, S)) < 0.01] = pad
    vconst = list(pack_strings(tok.vocab))
    d_ids = torch.as_tensor(ids, device="cuda")
    dec = VocabDecoder(skip_tokens=[pad], lib=hip_lib)
    fused = FusedDetokenizer(dec, byte_fallback=True).evaluate([d_ids] + vconst)
    r = dec.evaluate([d_ids] + vconst)
    bf = ByteFallback(lib=hip_lib).evaluate(list(r[2:5]))
    fz = FuzeRagged(lib=hip_lib).evaluate([r[0], r[1], bf[0], bf[1]])
    assert torch.equal(fused[0], fz[0]) and torch.equal(fused[1], fz[1]) and torch.equal(fused[2], bf[2]), "fused != chain"
    # length and byte checksum of every row = sums over its tokens (GPT-2-shaped vocabularies hold no <0xNN> tokens,
    # so ByteFallback changes nothing here)
    lens = (vconst[1] - vconst[0]).astype(np.int64)
    sums = np.add.reduceat(np.concatenate([vconst[2].astype(np.int64), [0]]), vconst[0].astype(np.int64)) * (lens > 0)
    lens[pad] = 0
    sums[pad] = 0
    want_len = lens[ids].sum(axis=1)
    got_len = (fused[1] - fused[0]).cpu().numpy().astype(np.int64)
    assert np.array_equal(got_len, want_len)
    csum = torch.cumsum(fused[2].to(torch.int64), 0)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), csum])
    got_sum = (csum[fused[1].long()] - csum[fused[0].long()]).cpu().numpy()
    assert np.array_equal(got_sum, sums[ids].sum(axis=1))
    k = 8
    o = O.vocab_decoder(ids[:k], tok.vocab, [pad])
    obf = O.byte_fallback(*o[2:5])
    ofz = O.fuze(o[0], o[1], obf[0], obf[1])
    assert np.array_equal(ofz[1], fused[1][:k].cpu().numpy()) and np.array_equal(obf[2][: int(ofz[1][-1])], fused[2][: int(ofz[1][-1])].cpu().numpy())


def test_detokenize_config5_as_stated(hip_lib):
    """BASELINE config 5 as stated: 1 048 576 x 2 048 token ids on one MI355X (8.6 GB of ids in, ~13 GB of text out), through
    FusedDetokenizer.evaluate_chunked(sink=...) -- row chunks below 2^31 output bytes each (the reference counts chars in
    int32, src/vocab_decoder.cpp:62-80), three HIP streams.  Per chunk: offsets ascending and gap-free from 0 to the chunk's
    byte count; EVERY row's length = the sum of its tokens' lengths; byte checksums of the chunk's first and last 4 096 rows =
    the sums over their tokens; the oracle chain bit for bit on the chunk's first 8 and last 2 rows."""
    import torch
    tok = BpeTok.load("gpt2")
    rows, S, V = 1048576, 2048, len(tok.vocab)
    pad = V - 1
    g = torch.Generator(device="cuda")
    g.manual_seed(55)
    ids = torch.empty((rows, S), dtype=torch.int32, device="cuda")
    for a in range(0, rows, 65536):   # (in slices: the int64 / float temporaries of the whole tensor would be 25 GB)
        part = torch.randint(0, V - 1, (65536, S), dtype=torch.int64, device="cuda", generator=g)
        part[torch.rand((65536, S), device="cuda", generator=g) < 0.01] = pad
        ids[a:a + 65536] = part.to(torch.int32)
        del part
    vconst = list(pack_strings(tok.vocab))
    lens = (vconst[1] - vconst[0]).astype(np.int64)
    sums = np.add.reduceat(np.concatenate([vconst[2].astype(np.int64), [0]]), vconst[0].astype(np.int64)) * (lens > 0)
    lens[pad] = 0
    sums[pad] = 0
    d_lens, d_sums = torch.as_tensor(lens, device="cuda"), torch.as_tensor(sums, device="cuda")
    fused = FusedDetokenizer(VocabDecoder(skip_tokens=[pad], lib=hip_lib), byte_fallback=True)
    seen = []

    def row_sums(table, a, b):
        out = torch.empty(b - a, dtype=torch.int64, device="cuda")
        for x in range(a, b, 16384):
            y = min(x + 16384, b)
            out[x - a:y - a] = table[ids[x:y].long()].sum(dim=1)
        return out

    def check_bytes(cb, ce, cc, a, lo, hi):   # rows [lo, hi) of the chunk that starts at batch row a
        c0, c1 = int(cb[lo]), int(ce[hi - 1])
        csum = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(cc[c0:c1].to(torch.int64), 0)])
        got = csum[(ce[lo:hi].long() - c0)] - csum[(cb[lo:hi].long() - c0)]
        assert torch.equal(got, row_sums(d_sums, a + lo, a + hi)), f"byte checksums differ in rows {a + lo}..{a + hi}"

    def oracle_rows(cb, ce, cc, a, lo, hi):
        o = O.vocab_decoder(ids[a + lo:a + hi].cpu().numpy(), tok.vocab, [pad])
        obf = O.byte_fallback(*o[2:5])
        ofz = O.fuze(o[0], o[1], obf[0], obf[1])
        c0 = int(cb[lo])
        assert np.array_equal(ofz[1] - ofz[0], (ce[lo:hi] - cb[lo:hi]).cpu().numpy())
        assert np.array_equal(obf[2][int(ofz[0][0]):int(ofz[1][-1])], cc[c0:int(ce[hi - 1])].cpu().numpy()), f"chunk at row {a}: oracle differs"

    def sink(a, b, cb, ce, cc):
        n = b - a
        assert int(cb[0]) == 0 and int(ce[-1]) == cc.numel() and cc.numel() <= (1 << 31) - 2
        assert torch.equal(cb[1:], ce[:-1]), "offsets are not gap-free"
        assert torch.equal((ce - cb).long(), row_sums(d_lens, a, b)), f"row lengths differ in chunk {a}..{b}"
        k = min(4096, n)
        check_bytes(cb, ce, cc, a, 0, k)
        check_bytes(cb, ce, cc, a, n - k, n)
        oracle_rows(cb, ce, cc, a, 0, min(8, n))
        oracle_rows(cb, ce, cc, a, max(n - 2, 0), n)
        seen.append((a, b, cc.numel()))

    n_chunks = fused.evaluate_chunked([ids] + vconst, sink=sink)
    assert n_chunks == len(seen) and n_chunks >= 6
    assert seen[0][0] == 0 and seen[-1][1] == rows and all(x[1] == y[0] for x, y in zip(seen, seen[1:])), "chunks do not tile the batch"
    assert sum(x[2] for x in seen) > 8 * (1 << 30)


@pytest.mark.parametrize("rows", [300000, 420000])
def test_many_short_rows(hip_lib, rows):
    """300 000 rows (more than the folded tail of merge_kernel takes: the separate exact / count_scan launches run; 49
    consecutive rows per wave of lookup_span_kernel) and 420 000 (more than 64 rows per wave on the persistent grid: the grid
    grows instead, 64 rows per wave) of ~24 bytes: halves back to back = the whole batch, offsets gap-free, the oracle on a prefix."""
    import torch
    tok = BpeTok.load("gpt2")
    b, e, c = TextModel(99, "zipf").batch(rows, 24, seed=5)
    rb, re_ = ragged_rows(rows)
    pat = tok.pattern_u8()
    fused = FusedSplitBPE(RegexSplit("isolate", lib=hip_lib), BPETokenizer(**tok.attrs, lib=hip_lib))

    def run(rb_, re2, b_, e_, c_):
        return fused.evaluate(dev([rb_, re2, b_, e_]) + [c_, pat], tok.consts)

    d_c = torch.as_tensor(c, device="cuda")
    whole = run(rb, re_, b, e, d_c)
    check_offsets(whole[0], whole[1], whole[2].numel())
    halves_equal_whole(run, rb, re_, b, e, d_c, whole)
    same_text(decode_rows(tok.vocab, whole[0], whole[1], whole[2], len(tok.vocab) - 1, hip_lib), b, e, c)
    k = 3000
    ref = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(rb[:k], re_[:k], b[:k], e[:k], c)[:5])
    assert np.array_equal(ref[1], whole[1][:k].cpu().numpy()) and np.array_equal(ref[2], whole[2][: int(ref[1][-1])].cpu().numpy())


def test_config4_as_stated_eight_shards_on_one_gpu(hip_lib):
    """BASELINE config 4 at its stated size -- 1 048 576 x ~512-byte mixed-script rows, Llama-3-shaped BPE (V = 128 256), sharded
    eight ways with a gather of the per-shard ragged ids -- on ONE GPU: the eight byte-balanced shards (shard_rows_by_bytes) are
    encoded straight to their wires one after the other (ovtk_encode_enqueue_wire: what each rank of an 8-GPU node does), one
    shard_unpack_kernel rebuilds the global tensor from the eight wires (what every rank does behind the all-gather), and the
    result equals the whole batch encoded in one ovtk_encode_run call -- begins, ends and ids bit for bit -- plus the oracle on a
    prefix of every shard.  Only the transport (RCCL over xGMI) is missing; `tests/test_distributed.py` covers it on gloo."""
    import ctypes as C

    import torch

    from openvino_tokenizers_amd import _lib as L
    from openvino_tokenizers_amd.distributed import shard_rows_by_bytes
    world, rows = 8, 1 << 20
    tok = BpeTok.load("llama3")
    b, e, c = TextModel(1234, "mixed").batch(rows, 512, seed=91)
    rb, re_ = ragged_rows(rows)
    pat = tok.pattern_u8()
    fused = FusedSplitBPE(RegexSplit("isolate", lib=hip_lib), BPETokenizer(**tok.attrs, lib=hip_lib))
    d_rb, d_re, d_b, d_e = dev([rb, re_, b, e])
    d_c = torch.as_tensor(c, device="cuda")
    whole = fused.evaluate([d_rb, d_re, d_b, d_e, d_c, pat], tok.consts)
    n_ids = whole[2].numel()
    check_offsets(whole[0], whole[1], n_ids)
    w_ends = whole[1].cpu().numpy()
    shards = shard_rows_by_bytes(b, e, world)
    assert shards[0][0] == 0 and shards[-1][1] == rows and all(shards[r][1] == shards[r + 1][0] for r in range(world - 1))
    text_per_shard = [int(e[hi - 1] - b[lo]) for lo, hi in shards]
    assert max(text_per_shard) - min(text_per_shard) <= 2 * 1024, "shards are not balanced by bytes"
    ids_per_shard = [int(w_ends[hi - 1]) - (int(w_ends[lo - 1]) if lo else 0) for lo, hi in shards]
    pad = (max(ids_per_shard) + 7) // 8 * 8
    h = C.c_void_p()
    L.check(hip_lib, hip_lib.ovtk_shard_exchange_create(world, C.c_int64(rows), 4, C.c_int64(max(hi - lo for lo, hi in shards)), 0, C.byref(h)))
    try:
        max_rows = int(hip_lib.ovtk_shard_max_rows(h))
        wire_bytes = int(hip_lib.ovtk_shard_wire_bytes(h, C.c_int64(pad)))
        wires = torch.zeros(world * wire_bytes, dtype=torch.uint8, device="cuda")
        for r, (lo, hi) in enumerate(shards):
            # the shard's rows keep their string indices: the strings tensor is the whole batch's, as on a rank that holds it
            rs = L.RaggedStrings(C.c_void_p(d_rb[lo:hi].data_ptr()), C.c_void_p(d_re[lo:hi].data_ptr()), hi - lo,
                                 L.Strings(C.c_void_p(d_b.data_ptr()), C.c_void_p(d_e.data_ptr()), C.c_void_p(d_c.data_ptr()), rows, len(c)))
            pending = C.c_void_p()
            L.check(hip_lib, hip_lib.ovtk_encode_enqueue_wire(fused.split._h, fused.bpe._h, C.byref(rs), None,
                                                              C.c_void_p(wires[r * wire_bytes:].data_ptr()), C.c_int64(max_rows), C.c_int64(pad), 4,
                                                              None, C.byref(pending)))
            out = L.RaggedI32Out(None, None, None, 0, 0, 0)
            L.check(hip_lib, hip_lib.ovtk_encode_finish(pending, C.byref(out)))
            assert out.n_data == ids_per_shard[r], f"shard {r}: {out.n_data} ids on the wire, {ids_per_shard[r]} in the whole batch"
        g_b = torch.empty(rows, dtype=torch.int32, device="cuda")
        g_e = torch.empty(rows, dtype=torch.int32, device="cuda")
        g_ids = torch.empty(n_ids, dtype=torch.int32, device="cuda")
        res = torch.zeros(4, dtype=torch.int64, device="cuda")
        L.check(hip_lib, hip_lib.ovtk_shard_unpack(h, C.c_void_p(wires.data_ptr()), C.c_int64(pad), C.c_void_p(g_b.data_ptr()), C.c_void_p(g_e.data_ptr()),
                                                   C.c_void_p(g_ids.data_ptr()), C.c_int64(n_ids), C.c_void_p(res.data_ptr()), L.MEM_DEVICE, None))
        torch.cuda.synchronize()
        r_ids, r_max, r_status, _ = (int(x) for x in res.cpu())
        assert r_status == 0 and r_ids == n_ids and r_max == max(ids_per_shard)
        assert torch.equal(g_b, whole[0]) and torch.equal(g_e, whole[1]) and torch.equal(g_ids, whole[2]), "the eight shards' wires do not give the whole batch"
    finally:
        hip_lib.ovtk_shard_exchange_destroy(h)
    # the oracle on the first rows of every shard
    orc, rs_o = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
    w_ids = whole[2]
    for lo, hi in shards:
        k = 250
        ref = orc(*rs_o(rb[:k], re_[:k], b[lo:lo + k], e[lo:lo + k], c)[:5])
        first = int(w_ends[lo - 1]) if lo else 0
        assert np.array_equal(ref[1] + first, w_ends[lo:lo + k]), f"rows {lo}..: ends differ from the oracle's"
        assert np.array_equal(ref[2], w_ids[first:first + int(ref[1][-1])].cpu().numpy()), f"rows {lo}..: ids differ from the oracle's"
