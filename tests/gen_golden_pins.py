"""Generates the round-3 pins of the oracle rows that had no external known answers (the reference pins
VocabDecoder / ByteFallback / FuzeRagged / WordPiece / BPE byte_fallback only through HuggingFace `decode` / `encode` on hub
models, tests/tokenizers_test.py:563,728,795; hub models are not reachable offline, HF `tokenizers` is):

  golden_detok_{gpt2_small,llama3_small}.npz   ids[B, S] (random ids, special ids included) -> HF Tokenizer.decode(ids,
        skip_special_tokens=True) as UTF-8 bytes per row.  Pins VocabDecoder (skip_tokens) -> FuzeRagged -> UTF8Validate
        (replace mode): the detokenizer chain the reference's converter builds for a byte-level BPE
        (python/openvino_tokenizers/tokenizer_pipeline.py, "VocabDecoderStep / FuseStep / UTF8ValidateStep").
  tok_spbpe_small.npz + tok_spbpe_small.hf.json   a SentencePiece-BPE-shaped tokenizer trained in-process: "<0xHH>" byte
        tokens, byte_fallback = true, "<unk>", Metaspace pieces -- the shape of Llama-2 / Mistral tokenizer.json files.
  golden_spbpe_small.npz   strings -> (pieces the Metaspace pre-tokenizer cuts, HF ids) and ids -> HF decode through
        decoders.Sequence([ByteFallback(), Fuse()]).  Pins BPETokenizer's byte_fallback / unk path on the encode side and
        VocabDecoder -> ByteFallback -> FuzeRagged on REAL fallback tokens on the decode side.
  golden_wordpiece_bert.npz   the full V = 30 522 `bert` tokenizer (BASELINE config 3) rebuilt as an HF WordPiece model from
        the committed vocabulary: strings -> ids.

  golden_bpe_{gpt2,llama3}.npz   (round 4) the HEADLINE tokenizers -- V = 50 257 of BASELINE config 2, V = 128 256 of config 4 --
        rebuilt as HF `models.BPE(vocab, merges)` + `Split(Regex(pattern), "isolated")` + ByteLevel from the committed
        tok_gpt2.npz / tok_llama3.npz: 2 100 rows each of zipf / mixed-script / uniform text at the configurations' ~512 bytes
        -> HF ids.  Pins the oracle (CPU tier) and the kernels (GPU tier) on the very tables bench.py measures with; the
        reference's counterpart is tests/tokenizers_test.py:563 (hub models against HF).

Run here (needs `tokenizers`); the .npz files are committed:    python -m tests.gen_golden_pins
"""
import json
from pathlib import Path

import numpy as np
from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

from tests.gen_golden import STRINGS, synthetic
from tools.make_tokenizers import load_tokenizer, pack, vocab_as_list
from tools.workloads import TextModel

G = Path(__file__).resolve().parent / "golden"


def _ragged(rows):
    lens = np.array([len(r) for r in rows], np.int64)
    ends = np.cumsum(lens).astype(np.int32)
    flat = np.concatenate([np.asarray(r) for r in rows if len(r)]) if lens.sum() else np.zeros(0)
    return (ends - lens).astype(np.int32), ends, flat


def _meta(**kw):
    kw["source"] = "tokenizers " + __import__("tokenizers").__version__
    return np.frombuffer(json.dumps(kw).encode(), np.uint8)


def main_detok():
    for name in ("gpt2_small", "llama3_small"):
        tok = Tokenizer.from_file(str(G / f"tok_{name}.hf.json"))
        tok.decoder = decoders.ByteLevel()
        t = load_tokenizer(name)
        V = len(t["vocab"])
        special = sorted(t["added"].values())
        rng = np.random.default_rng(2025)
        rows = []
        # (i) ids of real text: valid UTF-8 in, valid UTF-8 out
        texts = STRINGS + synthetic("mixed", 24, 120, 51)
        for s in texts:
            rows.append(tok.encode(s, add_special_tokens=False).ids)
        # (ii) uniform random ids (byte-level tokens glued at random: invalid UTF-8 the decoder must replace), special
        #      ids sprinkled in (skipped), and rows of nothing but special ids
        for k in range(40):
            n = int(rng.integers(1, 48))
            ids = rng.integers(0, V, n)
            ids[rng.random(n) < 0.08] = special[0]
            rows.append(ids.tolist())
        rows.append([special[0]] * 5)
        rows.append([])
        S = max(len(r) for r in rows)
        # VocabDecoder takes a dense [B, S] tensor: rows are padded with the special id (skipped on decode)
        ids = np.full((len(rows), S), special[0], np.int32)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
        out = [tok.decode(r.tolist(), skip_special_tokens=True).encode("utf-8") for r in ids]
        b, e, c = pack(out)
        np.savez_compressed(G / f"golden_detok_{name}.npz", ids=ids, skip_tokens=np.asarray(special, np.int32), out_begins=b,
                            out_ends=e, out_chars=c,
                            meta=_meta(tokenizer=f"tok_{name}.hf.json", what="Tokenizer.decode(ids, skip_special_tokens=True), "
                                       "decoder ByteLevel; expected = VocabDecoder(skip) -> FuzeRagged -> UTF8Validate(replace)"))
        print(name, ids.shape, "->", len(c), "bytes")


def main_spbpe():
    """A SentencePiece-BPE-shaped tokenizer (Llama-2 style): Metaspace pieces, byte_fallback, <unk>."""
    name = "spbpe_small"
    lines = TextModel(1234, "zipf").corpus_lines(2 << 20)
    byte_tokens = [f"<0x{b:02X}>" for b in range(256)]
    tok = Tokenizer(models.BPE(unk_token="<unk>", byte_fallback=True, fuse_unk=True))
    tok.pre_tokenizer = pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always", split=True)
    trainer = trainers.BpeTrainer(vocab_size=2500, special_tokens=["<unk>", "<s>", "</s>"] + byte_tokens, show_progress=False)
    tok.train_from_iterator(lines, trainer)
    tok.decoder = decoders.Sequence([decoders.ByteFallback(), decoders.Fuse()])
    (G / f"tok_{name}.hf.json").write_text(tok.to_str())
    tj = json.loads(tok.to_str())
    model = tj["model"]
    vocab = [t.encode("utf-8") for t in vocab_as_list(model["vocab"])]   # SentencePiece vocabularies are plain UTF-8
    merges = model["merges"]
    if merges and isinstance(merges[0], str):
        merges = [m.split(" ") for m in merges]
    merges = [(a.encode("utf-8"), b.encode("utf-8")) for a, b in merges]
    vb, ve, vc = pack(vocab)
    lb, le, lc = pack([m[0] for m in merges])
    rb, re_, rc = pack([m[1] for m in merges])
    ab, ae, ac = pack([])
    attrs = dict(unk_token="<unk>", fuse_unk=True, suffix_indicator="", end_suffix="", byte_fallback=True, cache_capacity=20000)
    meta = dict(kind="bpe", attrs=attrs, pattern="", behaviour="", trained_on="TextModel(1234,'zipf').corpus_lines(2 MiB)",
                hf="tokenizers", shape="SentencePiece BPE (Metaspace, <0xHH> byte tokens, byte_fallback)")
    np.savez_compressed(G / f"tok_{name}.npz", vocab_begins=vb, vocab_ends=ve, vocab_chars=vc, ml_begins=lb, ml_ends=le, ml_chars=lc,
                        mr_begins=rb, mr_ends=re_, mr_chars=rc, added_begins=ab, added_ends=ae, added_chars=ac,
                        added_ids=np.zeros(0, np.int32), meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
    # ---- encode side: the pieces Metaspace cuts (with the replacement character in place) -> ids
    texts = [s for s in STRINGS if s and "\x06" not in s] + synthetic("zipf", 32, 120, 61) + synthetic("mixed", 40, 160, 62)
    texts += ["café naïve über", "你好 世界", "emoji \U0001F600 end", "tab\there"]
    row_pieces, row_ids = [], []
    for s in texts:
        enc = tok.encode(s, add_special_tokens=False)
        pieces = [p for p, _ in tok.pre_tokenizer.pre_tokenize_str(s)]
        row_pieces.append([p.encode("utf-8") for p in pieces])
        row_ids.append(enc.ids)
    flat_pieces = [p for r in row_pieces for p in r]
    pb, pe, pc = pack(flat_pieces)
    n_p = np.array([len(r) for r in row_pieces], np.int64)
    rends = np.cumsum(n_p).astype(np.int32)
    ib, ie, ids = _ragged(row_ids)
    # ---- decode side: ids (with real <0xHH> runs in them) -> the decoder chain's text
    dec_rows = list(row_ids)
    rng = np.random.default_rng(77)
    V = len(vocab)
    for k in range(24):   # random ids: "<0xHH>" tokens in invalid orders, <unk>, <s>
        dec_rows.append(rng.integers(0, V, int(rng.integers(1, 40))).tolist())
    S = max(len(r) for r in dec_rows)
    pad = vocab.index(b"</s>")
    dids = np.full((len(dec_rows), S), pad, np.int32)
    for i, r in enumerate(dec_rows):
        dids[i, :len(r)] = r
    # HF's ByteFallback decoder turns a run of byte tokens that is not valid UTF-8 into one U+FFFD per token; the reference
    # leaves the raw bytes to UTF8Validate.  The pin is therefore on the BYTES before validation: tokens that are not
    # "<0xHH>" contribute their text, "<0xHH>" contributes byte HH -- which is HF's result wherever the bytes are valid
    # UTF-8; rows where they are not are kept out of the fixture's expected-text check (flagged).
    texts_out, exact = [], []
    for r in dids:
        s = tok.decode(r.tolist(), skip_special_tokens=False)
        raw = b"".join(bytes([int(vocab[i][3:5], 16)]) if (len(vocab[i]) == 6 and vocab[i].startswith(b"<0x")) else vocab[i] for i in r.tolist())
        ok = True
        try:
            ok = raw.decode("utf-8") == s
        except UnicodeDecodeError:
            ok = False
        texts_out.append(s.encode("utf-8"))
        exact.append(ok)
    ob, oe, oc = pack(texts_out)
    np.savez_compressed(G / f"golden_{name}.npz", piece_begins=pb, piece_ends=pe, piece_chars=pc, row_ends=rends,
                        id_begins=ib, id_ends=ie, ids=ids.astype(np.int32), dec_ids=dids, dec_begins=ob, dec_ends=oe, dec_chars=oc,
                        dec_exact=np.asarray(exact, np.bool_),
                        meta=_meta(tokenizer=f"tok_{name}.hf.json", what="encode: Metaspace pieces -> BPE ids (byte_fallback, unk); decode: "
                                   "decoders.Sequence([ByteFallback(), Fuse()]) = VocabDecoder -> ByteFallback -> FuzeRagged"))
    n_fb = int(sum(1 for i in ids.tolist() if 3 <= i < 259))
    print(name, "V", V, "M", len(merges), "strings", len(texts), "ids", len(ids), "byte-fallback ids", n_fb, "decode rows", len(dec_rows),
          "exact", int(sum(exact)))


def main_wordpiece_full():
    t = load_tokenizer("bert")
    vocab = {}
    for i, tokb in enumerate(t["vocab"]):
        vocab.setdefault(tokb.decode("utf-8"), i)
    tok = Tokenizer(models.WordPiece(vocab=vocab, unk_token="[UNK]", max_input_chars_per_word=int(t["max_bytes_per_word"])))
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    strings = [s.lower() for s in STRINGS if all(ord(ch) < 0x2E80 for ch in s) and "\x06" not in s]
    strings += [s.lower() for s in synthetic("zipf", 160, 256, 71)] + ["unaffable", "x" * 120, "a" * 100 + " b", "don't stop-me now!!!"]
    enc = [tok.encode(s, add_special_tokens=False).ids for s in strings]
    b, e, c = pack([s.encode("utf-8") for s in strings])
    ib, ie, ids = _ragged(enc)
    np.savez_compressed(G / "golden_wordpiece_bert.npz", begins=b, ends=e, chars=c, id_begins=ib, id_ends=ie, ids=ids.astype(np.int32),
                        meta=_meta(tokenizer="tok_bert.npz rebuilt as tokenizers.models.WordPiece + BertPreTokenizer"))
    print("bert", len(strings), "strings", len(ids), "ids")


def main_headline_bpe():
    from tokenizers import Regex
    from tools.make_tokenizers import gpt2_char_to_byte
    b2c = {b: c for c, b in gpt2_char_to_byte().items()}

    def chars(raw: bytes) -> str:
        return "".join(b2c[x] for x in raw)

    for name in ("gpt2", "llama3"):
        t = load_tokenizer(name)
        vocab = {}
        for i, tokb in enumerate(t["vocab"]):
            vocab.setdefault(chars(tokb), i)
        merges = [(chars(a), chars(b)) for a, b in t["merges"]]
        tok = Tokenizer(models.BPE(vocab=vocab, merges=merges))
        tok.pre_tokenizer = pre_tokenizers.Sequence([
            pre_tokenizers.Split(Regex(t["pattern"]), behavior="isolated", invert=False),
            pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
        strings = []
        for kind, seed in (("zipf", 81), ("mixed", 82), ("uniform", 83)):
            b, e, c = TextModel(seed, kind).batch(700, 512)
            raw = c.tobytes()
            strings += [raw[x:y] for x, y in zip(b.tolist(), e.tolist())]
        rows = [tok.encode(s.decode("utf-8"), add_special_tokens=False).ids for s in strings]
        sb, se, sc = pack(strings)
        ib, ie, ids = _ragged(rows)
        np.savez_compressed(G / f"golden_bpe_{name}.npz", begins=sb, ends=se, chars=sc, id_begins=ib, id_ends=ie, ids=ids.astype(np.int32),
                            meta=_meta(tokenizer=f"tok_{name}.npz", what="models.BPE(vocab, merges) + Split(Regex(pattern), isolated) + "
                                       "ByteLevel(use_regex=False); 700 rows each of TextModel(81,'zipf') / (82,'mixed') / (83,'uniform') at 512 bytes"))
        print(name, len(strings), "strings ->", len(ids), "ids")


if __name__ == "__main__":
    main_headline_bpe()
    if "--headline-only" in __import__("sys").argv:
        raise SystemExit(0)
    main_detok()
    main_spbpe()
    main_wordpiece_full()
