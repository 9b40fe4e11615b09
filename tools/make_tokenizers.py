"""Trains the *shaped* tokenizers used by tests and bench.py and stores them as fixtures.

Real GPT-2 / BERT / Llama-3 vocabulary files are not available offline, so the tokenizers are
trained in-process with HuggingFace `tokenizers` on the seeded synthetic corpus of
tools/workloads.py and converted exactly the way the reference's converter does before it
builds the BPETokenizer / WordpieceTokenizer ops:
  * byte-level BPE vocab + merges rewritten from GPT-2 "unicode chars" to RAW BYTES
    (python/openvino_tokenizers/utils.py:198-223, tokenizer_pipeline.py:674-694),
  * merges as (left, right) pairs -> the 14/18-input form of the op (tokenizer_pipeline.py:786-797),
  * added tokens (the special token) as their own string tensor + ids (tokenizer_pipeline.py:798-805).

Usage:  python -m tools.make_tokenizers [gpt2|gpt2_small|bert|llama3|all]
Output: tests/golden/tok_<name>.npz  (+ tok_<name>.hf.json for the small ones, so tests can
re-create the HF tokenizer; the large ones keep only the arrays the ops need).
"""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

from tools.workloads import TextModel

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"

GPT2_PATTERN = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"
LLAMA3_PATTERN = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*"
                  r"|\s*[\r\n]+|\s+(?!\S)|\s+")


def gpt2_char_to_byte():
    """transformers.models.gpt2 bytes_to_unicode, inverted (reference utils.py:198-212)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): b for c, b in zip(cs, bs)}


_C2B = gpt2_char_to_byte()


def chars_to_bytes(token: str, keep_corrupted: bool = False) -> bytes:
    """reference utils.py:215-225 (apply_unicode_to_bytes)."""
    try:
        return bytes(_C2B[ch] for ch in token)
    except KeyError:
        return token.encode() if keep_corrupted else b""


def pack(strings):
    lens = np.fromiter((len(s) for s in strings), dtype=np.int64, count=len(strings))
    ends = np.cumsum(lens).astype(np.int32)
    return (ends - lens).astype(np.int32), ends, np.frombuffer(b"".join(strings), dtype=np.uint8).copy()


def vocab_as_list(vocab: dict) -> list:
    """reference tokenizer_pipeline.py:517-530 (gaps filled with empty strings)."""
    out = []
    for tok, tid in sorted(vocab.items(), key=lambda x: x[1]):
        while len(out) < tid:
            out.append("")
        if len(out) == tid:
            out.append(tok)
        else:
            out[tid] = tok
    return out


def convert_byte_level_bpe(tok_json: dict):
    """HF tokenizer.json (ByteLevel BPE) -> the constant inputs + attributes of the BPETokenizer op."""
    model = tok_json["model"]
    vocab = [chars_to_bytes(t) for t in vocab_as_list(model["vocab"])]
    merges = model["merges"]
    if merges and isinstance(merges[0], str):
        merges = [m.split(" ") for m in merges]
    merges = [(chars_to_bytes(a), chars_to_bytes(b)) for a, b in merges]
    added = {t["content"]: t["id"] for t in tok_json["added_tokens"] if t["id"]}  # tokenizer_pipeline.py:712
    if added:
        grow = max(added.values()) - len(vocab) + 1
        vocab.extend(b"" for _ in range(max(grow, 0)))
    added_b = {}
    for t, i in added.items():
        tb = chars_to_bytes(t, keep_corrupted=True)
        vocab[i] = tb
        added_b[tb] = i
    attrs = dict(unk_token=model.get("unk_token") or "", fuse_unk=bool(model.get("fuse_unk")),
                 suffix_indicator=model.get("continuing_subword_prefix") or "",
                 end_suffix=model.get("end_of_word_suffix") or "", byte_fallback=bool(model.get("byte_fallback")),
                 cache_capacity=max(int(len(vocab) * 0.2), 20000))  # constants.py:35-36
    return vocab, merges, added_b, attrs


def save_bpe(name, vocab, merges, added, attrs, pattern, extra=None):
    vb, ve, vc = pack(vocab)
    lb, le, lc = pack([m[0] for m in merges])
    rb, re_, rc = pack([m[1] for m in merges])
    ab, ae, ac = pack(list(added.keys()))
    meta = dict(kind="bpe", attrs=attrs, pattern=pattern, behaviour="isolate", **(extra or {}))
    GOLDEN.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLDEN / f"tok_{name}.npz", vocab_begins=vb, vocab_ends=ve, vocab_chars=vc,
                        ml_begins=lb, ml_ends=le, ml_chars=lc, mr_begins=rb, mr_ends=re_, mr_chars=rc,
                        added_begins=ab, added_ends=ae, added_chars=ac,
                        added_ids=np.asarray(list(added.values()), dtype=np.int32),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def load_tokenizer(name):
    """-> dict(vocab=list[bytes], merges=list[(bytes,bytes)], added=dict, attrs=dict, pattern=str, ...)"""
    z = np.load(GOLDEN / f"tok_{name}.npz")
    meta = json.loads(bytes(z["meta"]).decode())

    def unpack(p):
        cb = bytes(z[p + "_chars"])
        return [cb[b:e] for b, e in zip(z[p + "_begins"].tolist(), z[p + "_ends"].tolist())]

    out = dict(meta)
    out["vocab"] = unpack("vocab")
    if meta["kind"] == "bpe":
        out["merges"] = list(zip(unpack("ml"), unpack("mr")))
        out["added"] = dict(zip(unpack("added"), z["added_ids"].tolist()))
    return out


def train_byte_level_bpe(name, vocab_size, corpus_bytes, pattern=GPT2_PATTERN, kind="zipf", keep_json=False,
                         special="<|endoftext|>"):
    from tokenizers import Regex, Tokenizer, models, pre_tokenizers, trainers

    t0 = time.time()
    lines = TextModel(1234, kind).corpus_lines(corpus_bytes)
    tok = Tokenizer(models.BPE())
    if pattern == GPT2_PATTERN:
        tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    else:
        tok.pre_tokenizer = pre_tokenizers.Sequence([
            pre_tokenizers.Split(Regex(pattern), behavior="isolated", invert=False),
            pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    trainer = trainers.BpeTrainer(vocab_size=vocab_size - 1, special_tokens=[], show_progress=False,
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(lines, trainer)
    tok.add_special_tokens([special])  # lands on the last id, as <|endoftext|> does in GPT-2
    tj = json.loads(tok.to_str())
    vocab, merges, added, attrs = convert_byte_level_bpe(tj)
    save_bpe(name, vocab, merges, added, attrs, pattern,
             extra=dict(trained_on=f"TextModel(1234,{kind!r}).corpus_lines({corpus_bytes})", hf="tokenizers"))
    if keep_json:
        (GOLDEN / f"tok_{name}.hf.json").write_text(tok.to_str())
    print(f"{name}: V={len(vocab)} M={len(merges)} added={added} in {time.time() - t0:.1f}s")
    return tok


def train_wordpiece(name, vocab_size, corpus_bytes, keep_json=False):
    from tokenizers import Tokenizer, models, pre_tokenizers, trainers

    t0 = time.time()
    lines = [l.lower() for l in TextModel(1234, "zipf").corpus_lines(corpus_bytes)]
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]", max_input_chars_per_word=100))
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    trainer = trainers.WordPieceTrainer(vocab_size=vocab_size, show_progress=False,
                                        special_tokens=["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    tok.train_from_iterator(lines, trainer)
    tj = json.loads(tok.to_str())
    vocab = [t.encode("utf-8") for t in vocab_as_list(tj["model"]["vocab"])]
    vb, ve, vc = pack(vocab)
    meta = dict(kind="wordpiece", suffix_indicator=tj["model"]["continuing_subword_prefix"],
                max_bytes_per_word=tj["model"]["max_input_chars_per_word"], unk_id=vocab.index(b"[UNK]"),
                trained_on=f"lower(TextModel(1234,'zipf').corpus_lines({corpus_bytes}))")
    np.savez_compressed(GOLDEN / f"tok_{name}.npz", vocab_begins=vb, vocab_ends=ve, vocab_chars=vc,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    if keep_json:
        (GOLDEN / f"tok_{name}.hf.json").write_text(tok.to_str())
    print(f"{name}: V={len(vocab)} in {time.time() - t0:.1f}s")
    return tok


TARGETS = {
    "gpt2": lambda: train_byte_level_bpe("gpt2", 50257, 48 << 20),
    "gpt2_small": lambda: train_byte_level_bpe("gpt2_small", 3000, 2 << 20, keep_json=True),
    "llama3_small": lambda: train_byte_level_bpe("llama3_small", 4000, 2 << 20, pattern=LLAMA3_PATTERN, kind="mixed",
                                                 keep_json=True, special="<|begin_of_text|>"),
    "llama3": lambda: train_byte_level_bpe("llama3", 128256, 96 << 20, pattern=LLAMA3_PATTERN, kind="mixed",
                                           special="<|begin_of_text|>"),
    "bert": lambda: train_wordpiece("bert", 30522, 32 << 20),
    "bert_small": lambda: train_wordpiece("bert_small", 2000, 2 << 20, keep_json=True),
}

if __name__ == "__main__":
    which = sys.argv[1:] or ["gpt2_small"]
    if which == ["all"]:
        which = list(TARGETS)
    for w in which:
        TARGETS[w]()
