"""Differential fuzzer for WordpieceTokenizer and the fused BERT chain: random vocabularies (prefix and ## continuation
tokens, duplicates, tokens longer than 15 bytes, non-ASCII), random words / sentences, max_bytes_per_word around the word
lengths; emulator build of the kernels against the oracle.    python tools/fuzz_wordpiece.py [seed] [n_cases]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.ops import FusedSplitWordpiece, RegexSplit, WordpieceTokenizer  # noqa: E402
from oracle import oracle as O  # noqa: E402  (the checker)
from tools.harness import one_string_per_row, pack_strings  # noqa: E402

BERT_WS = r"\s+"


def case(rng):
    letters = [bytes([x]) for x in rng.choice(list(b"abcdexyz"), size=int(rng.integers(2, 6)), replace=False)]
    if rng.random() < 0.3:
        letters += ["é".encode(), "元".encode()]
    si = b"##" if rng.random() < 0.8 else b"@"
    vocab = [b"[UNK]"]
    for _ in range(int(rng.integers(3, 60))):
        n = int(rng.choice([1, 1, 2, 2, 3, 4, 6, 9, 16, 22]))
        w = b"".join(letters[i] for i in rng.integers(len(letters), size=n))
        vocab.append((si if rng.random() < 0.5 else b"") + w)
    if rng.random() < 0.3:
        vocab += [b",", b"!"]
    words = []
    for _ in range(int(rng.integers(1, 60))):
        r = rng.random()
        if r < 0.5 and len(vocab) > 3:   # glue vocabulary strings together: words that tokenise
            parts = [vocab[i] for i in rng.integers(1, len(vocab), size=int(rng.integers(1, 5)))]
            w = b"".join(p[len(si):] if p.startswith(si) else p for p in parts)
        else:
            w = b"".join(letters[i] for i in rng.integers(len(letters), size=int(rng.integers(1, 30))))
        words.append(w)
    max_bytes = int(rng.choice([100, 100, 40, 10, 3, 1]))
    return vocab, si, words, max_bytes


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng(seed)
    lib = L.load(ROOT / "tests" / "emu" / "build" / "libovtk_emu.so")
    ws_pat = np.frombuffer(BERT_WS.encode(), np.uint8)
    from tools.harness import BERT_PUNCT
    pu_pat = np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    bad = 0
    for k in range(n):
        vocab, si, words, max_bytes = case(rng)
        # (a) the op on pre-split words, rows of random sizes
        cuts = np.sort(rng.integers(0, len(words) + 1, size=int(rng.integers(0, 4))))
        bounds = [0] + cuts.tolist() + [len(words)]
        b, e, c = pack_strings(words)
        rb = np.array(bounds[:-1], np.int32)
        re_ = np.array(bounds[1:], np.int32)
        inputs = [rb, re_, b, e, c]
        consts = list(pack_strings(vocab)) + [np.asarray(0, np.int32)]
        ref = O.WordpieceTokenizer(vocab, si.decode(), max_bytes)(*inputs, 0)
        got = WordpieceTokenizer(si.decode(), max_bytes, lib=lib).evaluate(inputs + consts)
        if not all(np.array_equal(np.asarray(a), np.asarray(g)) for a, g in zip(ref, got)):
            print("MISMATCH (op) case", k, vocab[:10], words[:6], max_bytes)
            bad += 1
            continue
        # (b) the fused chain on sentences made of the same words
        sents = [b" ".join(words[i:j]).decode("utf-8", "replace") + ("," if rng.random() < 0.3 else "") for i, j in zip(bounds[:-1], bounds[1:])]
        sin = one_string_per_row(sents)
        s1 = O.RegexSplit(BERT_WS, "remove")(*sin)
        s2 = O.RegexSplit(BERT_PUNCT, "isolate")(*s1[:5])
        ref2 = O.WordpieceTokenizer(vocab, si.decode(), max_bytes)(*s2[:5], 0)
        try:
            fused = FusedSplitWordpiece(RegexSplit("remove", lib=lib), RegexSplit("isolate", lib=lib), WordpieceTokenizer(si.decode(), max_bytes, lib=lib))
            got2 = fused.evaluate(sin, ws_pat, pu_pat, consts)
        except L.OvtkError as err:
            print("ERROR (fused) case", k, err)
            bad += 1
            continue
        if not all(np.array_equal(np.asarray(a), np.asarray(g)) for a, g in zip(ref2, got2)):
            print("MISMATCH (fused) case", k, vocab[:10], sents[:3], max_bytes)
            bad += 1
    print(f"seed {seed}: {n} cases, {bad} BAD")


if __name__ == "__main__":
    main()
