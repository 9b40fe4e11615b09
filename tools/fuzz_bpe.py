"""Differential fuzzer for BPETokenizer: random vocabularies and merge tables (not trained ones: duplicate strings, merges
whose result collides with other tokens, equal-rank ties through added tokens, missing bytes with and without unk /
byte_fallback, end_suffix), random pieces of every length class, the emulator build of the kernels against the oracle (the
reference's algorithm with the real std::priority_queue).
    python tools/fuzz_bpe.py [seed] [n_cases]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.ops import BPETokenizer  # noqa: E402
from oracle import oracle as O  # noqa: E402  (the checker)
from tools.harness import BpeTok  # noqa: E402


def pieces_inputs(rows):
    flat = [p for r in rows for p in r]
    b, e, c = O.pack_strings(flat)
    counts = np.array([len(r) for r in rows])
    re_ = np.cumsum(counts).astype(np.int32)
    return [(re_ - counts).astype(np.int32), re_, b, e, c]


def case(rng):
    alphabet = [bytes([x]) for x in rng.choice(list(b"abcdefgh \n"), size=int(rng.integers(2, 7)), replace=False)]
    if rng.random() < 0.3:
        alphabet.append("é".encode())
    suffix = b"</w>" if rng.random() < 0.25 else b""
    vocab = list(alphabet)
    if suffix:
        vocab.append(suffix)
    merges = []
    for _ in range(int(rng.integers(1, 40))):
        l, r = vocab[rng.integers(len(vocab))], vocab[rng.integers(len(vocab))]
        if len(l) + len(r) > 12:
            continue
        merges.append((l, r))
        if l + r not in vocab or rng.random() < 0.1:   # (sometimes a duplicate string: the later id wins)
            vocab.append(l + r)
    extra = []
    if rng.random() < 0.4:
        extra += [b"<unk>"]
    if rng.random() < 0.3:
        extra += [b"<0x%02X>" % x for x in rng.choice(256, size=20, replace=False)] + [b"<0x7A>", b"<0x71>"]
    order = rng.permutation(len(vocab))
    vocab = [vocab[i] for i in order] + extra
    attrs = dict(unk_token="<unk>" if b"<unk>" in extra and rng.random() < 0.8 else "", byte_fallback=bool(rng.random() < 0.5),
                 end_suffix=suffix.decode(), fuse_unk=bool(rng.random() < 0.3))
    if rng.random() < 0.3:
        attrs["cache_capacity"] = int(rng.choice([0, 1, 5, 20000]))
    added = None
    if rng.random() < 0.3:
        added = {b"Q": int(rng.integers(len(vocab))), b"RR": int(rng.integers(len(vocab)))}
    text_form = rng.random() < 0.3 and all(b" " not in m[0] + m[1] and b"\n" not in m[0] + m[1] for m in merges)
    tok = BpeTok(vocab, [m[0] + b" " + m[1] for m in merges] if text_form else merges, added, None, **attrs)
    letters = alphabet + [b"z", b"q"] + ([b"Q", b"RR"] if added else [])
    rows = []
    for _ in range(int(rng.integers(1, 12))):
        row = []
        for _ in range(int(rng.integers(0, 6))):
            n = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 15, 16, 17, 24, 32, 33, 60, 200, 600], p=[.04, .1, .1, .1, .1, .1, .1, .06, .05, .05, .05, .04, .03, .04, .03, .01]))
            row.append(b"".join(letters[i] for i in rng.integers(len(letters), size=n)))
        rows.append(row)
    return tok, rows


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng(seed)
    lib = L.load(ROOT / "tests" / "emu" / "build" / "libovtk_emu.so")
    bad = skipped = 0
    for k in range(n):
        tok, rows = case(rng)
        inputs = pieces_inputs(rows)
        cap = int(sum(len(p) + 8 for r in rows for p in r) * 2 + 64)
        try:
            ref = tok.oracle()(*inputs, cap=cap)
        except Exception as e:   # the reference rejects the tables (a merge token missing, ...)
            try:
                BPETokenizer(**tok.attrs, lib=lib).evaluate(inputs + tok.consts, ids_capacity=cap)
                print("ORACLE REFUSED, DEVICE DID NOT:", k, str(e)[:100])
                bad += 1
            except L.OvtkError:
                skipped += 1
            continue
        op = BPETokenizer(**tok.attrs, lib=lib)
        for rep in range(2):   # twice: the second call runs on what the memo learned
            got = op.evaluate(inputs + tok.consts, ids_capacity=cap)
            if not all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(ref, got)):
                print("MISMATCH case", k, "rep", rep, "attrs", tok.attrs, "vocab", tok.vocab[:12], "... merges", tok.merges[:8], "added", tok.added)
                bad += 1
                break
    print(f"seed {seed}: {n} cases, {skipped} refused by both, {bad} BAD")


if __name__ == "__main__":
    main()
