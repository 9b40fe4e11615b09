"""lookup_span_kernel (csrc/span_kernel.hpp): several rows per scan block, GPT-2 family.

Every case is the fused encode (RegexSplit `isolate` + BPETokenizer, `ovtk_encode_run`) against the oracle chain, on batches
large enough to leave the one-launch small-batch kernel (> 256 rows or > 64 KiB of text) and small enough for the emulator's
two blocks to own at most 64 rows per wave (<= 512 rows) -- the condition under which the launch code picks the span kernel.
What is probed: the block edges (rows of 1 .. 2 048 bytes, blocks that end exactly at 2 048), row starts inside a lane's 32
bytes, the rules' look-ahead / look-behind at row boundaries (white space, apostrophes, contractions cut by a row end), rows
the kernel must leave to the generic one (empty, longer than a block, non-ASCII, skipped, not contiguous), dense piece lists,
and miss-heavy text (the LDS miss list filling up and being written out mid-block).
Reference behaviour: src/regex_split.cpp:205-324 runs the pattern per string; src/bpe_tokenizer.cpp:47-164.
"""
import numpy as np
import pytest

from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
from oracle import oracle as O
from tests.util import BpeTok, assert_same
from tools.harness import pack_strings
from tools.workloads import TextModel, ragged_rows

DIGITS_PATTERN = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"


def fused_vs_oracle(backend, tok, inputs, skips=None, pattern=None, what="fused"):
    pattern = pattern or tok.pattern
    pat = np.frombuffer(pattern.encode(), np.uint8)
    o_in = [np.asarray(x) for x in inputs]
    sp_ref = O.RegexSplit(pattern, "isolate")(*o_in, skips=skips)
    ref = tok.oracle()(*sp_ref[:5])
    data = backend.data(inputs)
    sk = backend.data([np.asarray(skips, np.uint8)]) if skips is not None else []
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    got = fused.evaluate(data + sk + [pat], tok.consts)
    assert_same(ref, got, backend.host, what)
    return ref


def rows_of(strings):
    b, e, c = pack_strings(strings)
    rb, re_ = ragged_rows(len(b))
    return [rb, re_, b, e, c]


@pytest.mark.parametrize("kind,n,target", [("zipf", 400, 128), ("zipf", 288, 512), ("uniform", 320, 200), ("zipf", 300, 40),
                                           ("mixed", 300, 300), ("mixed", 280, 900)])
def test_text_models(backend, kind, n, target):
    b, e, c = TextModel(41, kind).batch(n, target)
    rb, re_ = ragged_rows(n)
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), [rb, re_, b, e, c], what=f"{kind} {n} x {target}")


def test_digits_variant(backend):
    b, e, c = TextModel(42, "zipf").batch(300, 160)
    rb, re_ = ragged_rows(300)
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), [rb, re_, b, e, c], pattern=DIGITS_PATTERN, what="individual digits")


def _filler(rng, n):
    """n bytes of word-like ASCII text."""
    words = [b"the", b"of", b"and", b"token", b"7", b"2024", b"don't", b"we'll", b"it's", b"I'm", b"x", b"  ", b"\n", b"\t", b"!?", b"'", b"''s",
             b"they've", b"you're", b"he'd", b"'t", b"a1b2", b"--", b"(", b")", b"e.g.", b" "]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(len(words)))]
        if rng.random() < 0.8:
            out += b" "
    return bytes(out[:n])


def test_row_lengths_around_the_block_edges(backend):
    """Rows of 1 .. 2 049 bytes in an order that makes blocks end exactly at, just before and just behind 2 048 bytes."""
    rng = np.random.default_rng(7)
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 1, 1, 1, 2047, 1, 2048, 2049, 1, 1024, 1024, 1023, 1025, 1000, 1048, 1, 2046, 2, 2045, 3, 1,
            700, 700, 648, 700, 700, 649, 512, 512, 512, 512, 512, 512, 512, 511, 1, 3000, 5, 4096, 7]
    lens = lens + [int(x) for x in rng.integers(1, 700, size=300 - len(lens))]
    strings = [_filler(rng, n) for n in lens]
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), rows_of(strings), what="block edges")


def test_rules_at_row_boundaries(backend):
    """What the rules look at across a byte must not be looked at across a row boundary: trailing / leading white space,
    an apostrophe at a row's end with contraction letters at the next row's start, contractions at a row's start, a space
    in front of a row that begins with a letter, digits runs cut by a row end."""
    ends = [b"abc ", b"abc  ", b"abc\n", b"abc \n ", b"abc'", b"abc 'l", b"abc'r", b"abc'", b"12", b"x 1", b"it'", b"!!", b"a\t", b" ", b"  ", b"'"]
    starts = [b"s next", b"ll be", b"e there", b"t", b" x", b"  x", b"'s", b"'ll go", b"34", b"d", b"m", b"ve", b"re", b"!", b"\nq", b" "]
    strings = []
    for a in ends:
        for s in starts:
            strings.append(a)
            strings.append(s)
    strings = strings[:500]
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), rows_of(strings), what="row boundaries")


def test_rows_left_to_the_generic_kernel(backend):
    """Empty rows, rows longer than a block, and skipped rows between ordinary ones; rows with non-ASCII text (those stay: the
    block takes the ballot form of the rules -- unless one of its rows is longer than 1 024 bytes)."""
    rng = np.random.default_rng(11)
    strings, skips = [], []
    for i in range(360):
        r = i % 12
        if r == 3:
            s = b""
        elif r == 5:
            s = _filler(rng, 2300 + i)
        elif r == 7:
            s = "naïve café über straße — ok".encode() + _filler(rng, 40 if i % 24 else 1100)
        elif r == 9:
            s = _filler(rng, 20) + "日本語のテキスト".encode() + _filler(rng, 30)
        else:
            s = _filler(rng, int(rng.integers(1, 400)))
        strings.append(s)
        skips.append(1 if r == 10 else 0)
    tok = BpeTok.load("gpt2_small")
    fused_vs_oracle(backend, tok, rows_of(strings), what="mixed rows")
    fused_vs_oracle(backend, tok, rows_of(strings), skips=np.asarray(skips, np.uint8), what="mixed rows with skips")


def _long_rows(rng):
    """Rows longer than a scan block: ASCII, with non-ASCII text, with pieces longer than a block or a ballot window."""
    jp = "日本語のテキストを分割する。".encode()
    mixed = lambda n: b"".join((_filler(rng, int(rng.integers(5, 200))) + ("naïve café — ".encode() if rng.random() < 0.5 else jp)) for _ in range(n // 100 + 1))[:n].decode(errors="ignore").encode()   # (cut at a character boundary)
    rows = [_filler(rng, 3000), _filler(rng, 5000), _filler(rng, 10000), _filler(rng, 2048 * 3), _filler(rng, 2048 * 2 + 8), _filler(rng, 2040 * 2),
            mixed(3000), mixed(5000), mixed(9000), mixed(1024), mixed(1025), mixed(2047), mixed(2049),
            b"a" * 3000, b" " * 5000, b"\n" * 2100 + b"x", "日".encode() * 1000, "日".encode() * 3000, b"7" * 4000, b"!" * 2048, b"!" * 2049,
            _filler(rng, 700) + b"b" * 2500 + _filler(rng, 700), _filler(rng, 2040) + b"c" * 40 + _filler(rng, 100),
            _filler(rng, 2030) + b" " * 30 + _filler(rng, 100), _filler(rng, 2041) + b"don't" + _filler(rng, 100),
            _filler(rng, 100) + "é".encode() * 1200 + _filler(rng, 100), jp * 40 + b"d" * 1100 + jp * 3,
            mixed(900) + b" " * 1500 + "ü".encode() + _filler(rng, 50), _filler(rng, 1500) + b"'" + b"l" * 3000, b"'" * 2500,
            _filler(rng, 2044) + b"'ll " + _filler(rng, 50), _filler(rng, 2046) + b"'re" + _filler(rng, 50), b" " * 2047 + b"x", b" " * 2048 + b"x",
            b"x" + b" " * 2047, b"x" * 2047 + b" ", mixed(2000) + "\U0001F600\U0001F601".encode() + _filler(rng, 300), "\U0001F600".encode() * 600]
    return rows


def test_rows_longer_than_a_block(backend):
    """Rows of 2 049 .. 10 000 bytes slide through the kernel's blocks: every block stops at the last piece start it can decide.
    Pieces longer than a block (3 000 letters, 5 000 blanks, 3 000 CJK characters, 4 000 digits) take the literal matcher and go
    to the deferred list directly; blocks with non-ASCII text take the ballot form window by window."""
    rng = np.random.default_rng(53)
    long_rows = _long_rows(rng)
    strings = []
    for i in range(300):
        strings.append(long_rows[(i // 3) % len(long_rows)] if i % 3 == 0 else _filler(rng, int(rng.integers(1, 300))))
    tok = BpeTok.load("gpt2_small")
    fused_vs_oracle(backend, tok, rows_of(strings), what="long rows")
    if backend.name == "emu":   # (the emulator takes ten seconds per leg: the other two run on the GPU tier)
        return
    fused_vs_oracle(backend, tok, rows_of(strings), pattern=DIGITS_PATTERN, what="long rows, digits variant")
    # nothing but long rows, back to back (chains of them)
    only = [long_rows[i % len(long_rows)] for i in range(290)]
    fused_vs_oracle(backend, tok, rows_of(only), what="only long rows")


def test_long_rows_share_a_work_item_of_compact_kernel(backend):
    """Rows of ~4 300 bytes, 260 of them: compact_kernel's work items (four rows, ~4 000 ids) are shared by two waves
    (EncodeWork::compact_split) -- where an item has no unused staging entry its copy is dealt out among them, an item with a deferred
    piece is squeezed by the first alone.  Twice: the first call defers what the memo has not seen, the second runs on what was learned."""
    rng = np.random.default_rng(54)
    strings = [_filler(rng, 4300 + int(rng.integers(0, 64))) for _ in range(260)]
    strings[17] = strings[17][:2000] + b"x" * 600 + strings[17][2600:]   # (a piece no memo entry holds: its item is squeezed on every call)
    tok = BpeTok.load("gpt2_small")
    pat = np.frombuffer(tok.pattern.encode(), np.uint8)
    inputs = rows_of(strings)
    ref = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(*[np.asarray(x) for x in inputs])[:5])
    fused = FusedSplitBPE(RegexSplit("isolate", lib=backend.lib), BPETokenizer(**tok.attrs, lib=backend.lib))
    for call in range(2):
        assert_same(ref, fused.evaluate(backend.data(inputs) + [pat], tok.consts), backend.host, f"long rows, call {call}")


def test_apostrophe_and_one_letter_at_the_end_of_a_full_block(backend):
    """a chain's LAST block of exactly 2 048 bytes that ends in an apostrophe and r / v / l -- the contraction rules
    ('re, 've, 'll) look one byte further, which is behind the block AND behind the text: whatever stands there (an `e` or an `l` left by
    the block before, in the packed-byte form of round 4) must not make the two bytes one piece.  Rows of exactly 2 048 bytes, chains of
    rows that add up to 2 048 and to 4 096, each followed by rows that begin with the letters that would complete the contraction."""
    rng = np.random.default_rng(55)
    strings = []
    for tail in (b"'r", b"'v", b"'l", b"'R", b"'L", b"x'"):
        body = _filler(rng, 2048 - len(tail) - 1) + b" "
        strings += [b"e" * 40 + b" l" * 30, body[:2048 - len(tail)] + tail, b"e", b"ll", b"le"]          # one row = one full block
        strings += [body[:1000], body[1000:2048 - len(tail)] + tail, b"e e e", b"l"]                      # two rows = one full block
        strings += [_filler(rng, 2048), body[:2048 - len(tail)] + tail, b"e"]                              # the chain's second block
    strings += [_filler(rng, int(rng.integers(1, 200))) for _ in range(300 - len(strings))]
    for s_ in strings:
        assert len(s_) >= 1
    tok = BpeTok.load("gpt2_small")
    fused_vs_oracle(backend, tok, rows_of(strings), what="apostrophe + letter at a full block's end")


def test_rows_that_are_not_contiguous(backend):
    """begins / ends that leave gaps, overlap, or run backwards through the chars tensor: no block may span such a seam."""
    rng = np.random.default_rng(13)
    chars = np.frombuffer(_filler(rng, 60000), np.uint8).copy()
    n = 300
    b = np.zeros(n, np.int32)
    e = np.zeros(n, np.int32)
    at = 0
    for i in range(n):
        ln = int(rng.integers(1, 300))
        kind = i % 5
        if kind == 1:
            at += int(rng.integers(1, 9))        # a gap
        elif kind == 2:
            at = max(0, at - int(rng.integers(1, 20)))   # overlaps the previous row
        elif kind == 3 and i % 10 == 3:
            at = int(rng.integers(0, 50000))     # somewhere else entirely
        if at + ln > len(chars):
            at = int(rng.integers(0, 1000))
        b[i], e[i] = at, at + ln
        at += ln
    rb, re_ = ragged_rows(n)
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), [rb, re_, b, e, chars], what="seams")
    # two strings per row here and there: never a span row
    rb2 = np.arange(0, n, 2, dtype=np.int32)
    re2 = rb2 + 2
    re2[::7] -= 1
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), [rb2, np.minimum(re2, n).astype(np.int32), b, e, chars], what="two strings per row")


def test_dense_piece_lists(backend):
    """Every byte (or every other byte) a piece: 2 048 resp. 1 024 pieces in one block."""
    strings = []
    for i in range(280):
        r = i % 4
        if r == 0:
            strings.append(b"a!" * 300)
        elif r == 1:
            strings.append(b" a" * 500)
        elif r == 2:
            strings.append(b"1,2;3.4" * 120)
        else:
            strings.append(b"!a" * 1024)
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), rows_of(strings), what="dense pieces")


def test_miss_heavy_text(backend):
    """Pieces the memo does not hold, many per round: the wave's LDS miss list is written out in the middle of a block;
    pieces of 16 bytes and more (no key) among them."""
    rng = np.random.default_rng(17)
    alphabet = np.frombuffer(b"qzxjkvwpy", np.uint8)
    strings = []
    for i in range(300):
        words = []
        for _ in range(int(rng.integers(5, 60))):
            words.append(bytes(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(3, 24)))]))
        strings.append(b" ".join(words))
    tok = BpeTok.load("gpt2_small")
    fused_vs_oracle(backend, tok, rows_of(strings), what="misses")
    tok.attrs = dict(tok.attrs, cache_capacity=0)   # no memo at all: every piece is a miss
    fused_vs_oracle(backend, tok, rows_of(strings[:280]), what="no memo")


def test_last_row_ends_the_chars_tensor(backend):
    """The last block's lanes may not read past the end of the chars tensor (byte-wise loads there)."""
    rng = np.random.default_rng(19)
    for tail in (1, 5, 17, 31, 32, 33):
        strings = [_filler(rng, int(rng.integers(1, 200))) for _ in range(299)] + [_filler(rng, tail)]
        fused_vs_oracle(backend, BpeTok.load("gpt2_small"), rows_of(strings), what=f"tail {tail}")


def test_contractions_at_every_lane_offset(backend):
    """An apostrophe at every offset of a lane's 32 bytes -- the letters behind it, and the piece start behind those, may
    belong to the next lane -- for every contraction, a near miss of each, and with the row ending inside the contraction."""
    tails = [b"'s x", b"'t", b"'m.", b"'d1", b"'re ", b"'ve!", b"'ll", b"'l", b"'r", b"'v ", b"'S", b"'lL x", b"''s", b" 's", b"!'t", b"7'd", b"\t'm", b"'", b"'s's'll're"]
    strings = []
    for off in range(0, 67):
        for i, t in enumerate(tails):
            if (off + i) % 3 == 0:
                strings.append(b"ab cd "[: off % 6] + b"x" * (off - off % 6) + t)
    strings = strings[:500]
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), rows_of(strings), what="contractions")
    fused_vs_oracle(backend, BpeTok.load("gpt2_small"), rows_of(strings), pattern=DIGITS_PATTERN, what="contractions, digits variant")


def test_bert_words_through_the_span_kernel(backend):
    """The fused WordPiece path on lookup_span_kernel<kSpanBertWords>: white space and delimiters at every row boundary and lane
    offset, rows of nothing but white space / delimiters, rows around the block edges, non-ASCII and over-long rows (left to the
    generic kernel), words longer than a memo key -- against the two RegexSplit ops + WordpieceTokenizer of the oracle, cold and on
    the word store's second call."""
    from openvino_tokenizers_amd.ops import FusedSplitWordpiece, WordpieceTokenizer
    from tests.test_ops_parity import BERT_PUNCT, BERT_WS, bert_words, wp_consts
    from tools.make_tokenizers import load_tokenizer
    tok = load_tokenizer("bert_small")
    rng = np.random.default_rng(37)
    frag = ["the", "token", "izer", "un", "affable", "hello", "world", "x", "a1", "2024", ",", ".", "!?", "(", ")", "--", " ", "  ", "\t", "\n", " , ",
            "word" * 5, "q" * 17, "don't", "e.g.", "[", "]", "{~}", "^_`"]
    strings = []
    for i in range(330):
        r = i % 15
        if r == 11:
            s = "   \t\n  " * int(rng.integers(1, 9))
        elif r == 12:
            s = ",.;:!?" * int(rng.integers(1, 30))
        elif r == 10:
            s = [" ", "\t", " ,", ", ", "x ", " x", "  "][i % 7]   # (a blank between two words gets no entry in the piece list)
        elif r == 13:
            s = "naïve café 元気 — " + "".join(rng.choice(frag, size=6))
        elif r == 14:
            s = " ".join(rng.choice(frag, size=700))[: 2100 + i]
        else:
            s = "".join(rng.choice(frag, size=int(rng.integers(1, 90)))) + (" " if i % 2 else "")
            if r == 3:
                s = s[: int(rng.choice([1, 31, 32, 33, 63, 64, 65]))]
        strings.append(s)
    inputs = rows_of([s.encode() for s in strings])
    ws_pat, pu_pat = np.frombuffer(BERT_WS.encode(), np.uint8), np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    ref = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*bert_words(inputs), tok["unk_id"])
    fused = FusedSplitWordpiece(RegexSplit("remove", lib=backend.lib), RegexSplit("isolate", lib=backend.lib),
                                WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=backend.lib))
    for call in range(2):
        got = fused.evaluate(backend.data(inputs), ws_pat, pu_pat, wp_consts(tok))
        assert_same(ref, got, backend.host, f"BERT words, call {call}")


def test_bert_long_rows(backend):
    """BERT words of the fused WordPiece path over rows longer than a block, words longer than a block, white space runs longer
    than a block (dropped), non-ASCII blocks."""
    from openvino_tokenizers_amd.ops import FusedSplitWordpiece, WordpieceTokenizer
    from tests.test_ops_parity import BERT_PUNCT, BERT_WS, bert_words, wp_consts
    from tools.make_tokenizers import load_tokenizer
    tok = load_tokenizer("bert_small")
    rng = np.random.default_rng(59)
    frag = ["the", "token", "izer", "un", "affable", "hello", "world", ",", ".", "!?", "(", " ", "  ", "\t", "\n", "don't", "e.g.", "naïve", "元気", "—", "straße"]
    ascii_frag = [f for f in frag if f.isascii()]
    text = lambda n, fr: " ".join(rng.choice(fr, size=n))[:n]
    long_rows = [text(3000, ascii_frag), text(9000, ascii_frag), text(3000, frag), text(7000, frag), "a" * 3000, " " * 5000, "x" + " " * 4100 + "y",
                 "," * 2500, "日" * 1500, text(2040, ascii_frag) + "w" * 50 + " z", text(2040, ascii_frag) + " " * 50 + "z", text(500, frag) + "é" * 1300 + " ok",
                 text(1000, frag) + "\u3000" * 900 + text(100, frag), "q" * 2047 + ",", "q" * 2048 + ",", " " * 2048 + "," + " " * 2048]
    strings = [long_rows[(i // 3) % len(long_rows)] if i % 3 == 0 else text(int(rng.integers(1, 300)), frag) for i in range(300)]
    inputs = rows_of([s.encode() for s in strings])
    ws_pat, pu_pat = np.frombuffer(BERT_WS.encode(), np.uint8), np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    ref = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*bert_words(inputs), tok["unk_id"])
    fused = FusedSplitWordpiece(RegexSplit("remove", lib=backend.lib), RegexSplit("isolate", lib=backend.lib),
                                WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=backend.lib))
    for call in range(2):
        got = fused.evaluate(backend.data(inputs), ws_pat, pu_pat, wp_consts(tok))
        assert_same(ref, got, backend.host, f"BERT long rows, call {call}")


def test_bert_row_whose_only_piece_in_a_block_is_an_omitted_blank(backend):
    """The piece list of the BERT words leaves out the one blank between two words.  Row B starts with such a blank, one byte in front
    of where block 0 stops (its last decided piece start), behind exactly 640 entries: B's first piece of that block is `entry 640`
    of 640 -- no lane of any lookup round holds it."""
    from openvino_tokenizers_amd.ops import FusedSplitWordpiece, WordpieceTokenizer
    from tests.test_ops_parity import BERT_PUNCT, BERT_WS, bert_words, wp_consts
    from tools.make_tokenizers import load_tokenizer
    tok = load_tokenizer("bert_small")
    rng = np.random.default_rng(61)
    for n_words in (640, 639, 641, 576):
        a = "ab " * (n_words - 1) + "q" * (2025 - 3 * (n_words - 1))
        b = " " + "c" * 30 + " the end"
        assert len(a) == 2025
        strings = [a, b] + [" ".join(rng.choice(["the", "token", "izer", ","], size=int(rng.integers(1, 40)))) for _ in range(300)]
        inputs = rows_of([x.encode() for x in strings])
        ws_pat, pu_pat = np.frombuffer(BERT_WS.encode(), np.uint8), np.frombuffer(BERT_PUNCT.encode(), np.uint8)
        ref = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])(*bert_words(inputs), tok["unk_id"])
        fused = FusedSplitWordpiece(RegexSplit("remove", lib=backend.lib), RegexSplit("isolate", lib=backend.lib),
                                    WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], lib=backend.lib))
        got = fused.evaluate(backend.data(inputs), ws_pat, pu_pat, wp_consts(tok))
        assert_same(ref, got, backend.host, f"{n_words} entries in front of the blank")


def _l3_rows(rng, n):
    """Rows for the Llama-3 family: digit runs of every length, line breaks inside white space, contractions in both cases,
    non-ASCII letters / digits / white space, U+017F, rows cut at any byte of those."""
    frag = ["the", "Token", "izer", " ", "  ", "\n", " \n", "\n\n", "\r\n", " \n  ", "\t", "1", "12", "123", "1234", "12345678", "1234567890123", "3.14", "don't", "DON'T", "we'LL",
            "I'm", "'s", "'", "''re", "!?", "--", "(x)", "e.g.", "naïve", "straße", "Ωμέγα", "привет", "日本語", "テキスト", "😀", "١٢٣", "ſ", "\u00a0", "\u3000", "a1b2", "x ", " y",
            "#include <stdio.h>\n", "    return 0;\n}\n", "http://a.b/c?d=1&e=2"]
    rows = []
    for i in range(n):
        k = int(rng.integers(1, 60)) if i % 7 else int(rng.integers(150, 500))
        t = "".join(rng.choice(frag, size=k))
        if i % 5 == 0:
            t = t[: int(rng.integers(1, len(t) + 1))]
        rows.append(t.encode())
    return rows


@pytest.mark.parametrize("name", ["llama3", "qwen2", "cl100k"])
def test_llama3_family_long_and_ragged_rows(backend, name):
    """The Llama-3 family through the fused encode (lookup_rows_kernel<kRowsLlama3>: llama3_packed_starts, the ballot form where that
    one declines -- long digit runs, many line breaks --, the literal matcher where neither decides -- non-ASCII digits, U+017F):
    the same rows as the span kernel's tests, rows of any length, pieces longer than a window.  (Round 4 ran these through
    lookup_span_kernel with the Llama-3 scanners window by window: bit-exact, but slower than a row at a time)"""
    from tools.workloads import MODEL_PATTERNS
    tok = BpeTok.load("llama3_small")
    pattern = MODEL_PATTERNS.get(name, tok.pattern)
    rng = np.random.default_rng(67)
    fused_vs_oracle(backend, tok, rows_of(_l3_rows(rng, 300)), pattern=pattern, what=f"{name}: fragments")
    if backend.name == "emu" and name != "llama3":   # (thirty seconds per pattern on the emulator: the other legs of the two variants run on the GPU tier)
        return
    b, e, c = TextModel(43, "mixed").batch(300, 700)
    rb, re_ = ragged_rows(300)
    fused_vs_oracle(backend, tok, [rb, re_, b, e, c], pattern=pattern, what=f"{name}: mixed text, rows of ~700 bytes")
    long_rows = _long_rows(rng) + [b"1" * 3000, b"\n" * 3000, b" \n" * 1200 + b"x", ("é" * 900 + " ").encode() * 3, b"9" * 600 + b" " + b"8" * 700]
    strings = [long_rows[(i // 3) % len(long_rows)] if i % 3 == 0 else _filler(rng, int(rng.integers(1, 300))) for i in range(300)]
    fused_vs_oracle(backend, tok, rows_of(strings), pattern=pattern, what=f"{name}: long rows")
