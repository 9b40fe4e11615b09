"""Generates tests/golden/golden_*.npz: input strings + the ids HuggingFace `tokenizers` produces for them
with the committed *_small tokenizers (tests/golden/tok_*_small.hf.json).  HF is the differential oracle of
the reference's own end-to-end tests (tests/tokenizers_test.py:479-536); the strings follow that file's
categories (English with whitespace/digit runs, multilingual, emoji, whitespace-only, control chars) plus
seeded synthetic text.  Run here (needs `tokenizers`); the .npz files are committed.

    python -m tests.gen_golden
"""
import json
from pathlib import Path

import numpy as np
from tokenizers import Tokenizer

from tools.workloads import TextModel

G = Path(__file__).resolve().parent / "golden"

STRINGS = [
    "Eng... test, string?!",
    "Multiline\nstring!\nWow!",
    "A lot\t w!",
    "A lot\t\tof whitespaces!",
    "\n\n\n\t\t   A    lot\t\tof\twhitespaces\n!\n\n\n\t\n\n",
    "Eng, but with d1gits: 123; 0987654321, stop.0987654321 - eng, but with d1gits: 123",
    "USER: <image>\nWhat is in the image? ASSISTANT:",
    "What is OpenVINO?",
    "it's don't we'll they've I'm he'd you're 'tis 'sup x's's 'S 'RE",
    "If I have 100 million dollars, what kinds of projects should I invest to maximize my benefits?",
    "Тестовая строка!",
    "Testzeichenfolge?",
    "Tester, la chaîne...",
    "測試字符串",
    "سلسلة الاختبار",
    "מחרוזת בדיקה",
    "Сынақ жолы á",
    "介绍下清华大学",
    "若我有一亿美元，在人工智能盛行的今天，我怎样投资才能收益最大化？",
    "😀",
    "😁😁",
    "🤣🤣🤣😁😁😁😁",
    "🤷‍♂️",
    "🤦🏼‍♂️",
    "",
    "\x06",
    " ",
    " " * 10,
    " " * 256,
    "\n",
    " \t\n",
    "a" * 40,
    "ab" * 150,
    # (strings containing the special token are left out: HF cuts them out before pre-tokenisation, which in the
    #  reference is SpecialTokensSplit's job -- the op before this path; see test_skips_pass_through)
    "tab\tseparated\tvalues  and   runs    of     spaces      end ",
    "x y　z w",
]


def synthetic(kind, n, target, seed):
    b, e, c = TextModel(4321, kind).batch(n, target, seed=seed)
    raw = c.tobytes()
    return [raw[x:y].decode("utf-8") for x, y in zip(b.tolist(), e.tolist())]


def main():
    strings = STRINGS + synthetic("zipf", 48, 160, 11) + synthetic("mixed", 32, 200, 12) + synthetic("uniform", 16, 96, 13)
    for name in ("gpt2_small",):
        tok = Tokenizer.from_file(str(G / f"tok_{name}.hf.json"))
        enc = [tok.encode(s, add_special_tokens=False).ids for s in strings]
        raw = [s.encode("utf-8") for s in strings]
        lens = np.array([len(r) for r in raw], np.int64)
        ends = np.cumsum(lens).astype(np.int32)
        tl = np.array([len(x) for x in enc], np.int64)
        tends = np.cumsum(tl).astype(np.int32)
        np.savez_compressed(G / f"golden_bpe_{name}.npz", begins=(ends - lens).astype(np.int32), ends=ends,
                            chars=np.frombuffer(b"".join(raw), np.uint8), id_begins=(tends - tl).astype(np.int32),
                            id_ends=tends, ids=np.concatenate([np.asarray(x, np.int32) for x in enc if len(x)]),
                            meta=np.frombuffer(json.dumps(dict(source="tokenizers " + __import__("tokenizers").__version__,
                                                               tokenizer=f"tok_{name}.hf.json")).encode(), np.uint8))
        print(name, len(strings), "strings", int(tl.sum()), "ids")


def main_wordpiece():
    """WordPiece (bert_small): HF BertPreTokenizer + WordPiece, no normalizer -- lower-cased ASCII/Latin text only
    (HF handles CJK in its normalizer, the reference in the split pattern; both agree on everything else)."""
    strings = [s.lower() for s in STRINGS if all(ord(ch) < 0x2E80 for ch in s) and "\x06" not in s]
    strings += [s.lower() for s in synthetic("zipf", 64, 160, 21)] + ["unaffable", "x" * 120, "a" * 100 + " b", "don't stop-me now!!!"]
    name = "bert_small"
    tok = Tokenizer.from_file(str(G / f"tok_{name}.hf.json"))
    enc = [tok.encode(s, add_special_tokens=False).ids for s in strings]
    raw = [s.encode("utf-8") for s in strings]
    lens = np.array([len(r) for r in raw], np.int64)
    ends = np.cumsum(lens).astype(np.int32)
    tl = np.array([len(x) for x in enc], np.int64)
    tends = np.cumsum(tl).astype(np.int32)
    np.savez_compressed(G / f"golden_wordpiece_{name}.npz", begins=(ends - lens).astype(np.int32), ends=ends,
                        chars=np.frombuffer(b"".join(raw), np.uint8), id_begins=(tends - tl).astype(np.int32),
                        id_ends=tends, ids=np.concatenate([np.asarray(x, np.int32) for x in enc if len(x)]),
                        meta=np.frombuffer(json.dumps(dict(source="tokenizers " + __import__("tokenizers").__version__,
                                                           tokenizer=f"tok_{name}.hf.json")).encode(), np.uint8))
    print(name, len(strings), "strings", int(tl.sum()), "ids")


def main_llama3():
    """Llama-3-shaped byte-level BPE (tiktoken-style split pattern, trained in-process): HF ids for mixed-script text."""
    strings = STRINGS + synthetic("zipf", 32, 160, 31) + synthetic("mixed", 48, 200, 32) + synthetic("uniform", 8, 96, 33)
    strings += ["it's IT'S don'T we'LL 'ſ 12345 6,789.10\r\n\r\n  end  ", "a\tb \n c\n\n  d", "x\u00a0y !\n\nz"]
    name = "llama3_small"
    tok = Tokenizer.from_file(str(G / f"tok_{name}.hf.json"))
    enc = [tok.encode(s, add_special_tokens=False).ids for s in strings]
    raw = [s.encode("utf-8") for s in strings]
    lens = np.array([len(r) for r in raw], np.int64)
    ends = np.cumsum(lens).astype(np.int32)
    tl = np.array([len(x) for x in enc], np.int64)
    tends = np.cumsum(tl).astype(np.int32)
    np.savez_compressed(G / f"golden_bpe_{name}.npz", begins=(ends - lens).astype(np.int32), ends=ends,
                        chars=np.frombuffer(b"".join(raw), np.uint8), id_begins=(tends - tl).astype(np.int32),
                        id_ends=tends, ids=np.concatenate([np.asarray(x, np.int32) for x in enc if len(x)]),
                        meta=np.frombuffer(json.dumps(dict(source="tokenizers " + __import__("tokenizers").__version__,
                                                           tokenizer=f"tok_{name}.hf.json")).encode(), np.uint8))
    print(name, len(strings), "strings", int(tl.sum()), "ids")


def main_truncate():
    """Pair truncation known answers from HF tokenizers (WordLevel vocabulary, no post-processor so max_length
    counts only the two sequences): kept lengths for every (len_a, len_b, max_length, side), longest_first -- the mode
    tokenizer_pipeline.py:955 always builds."""
    from tokenizers import Tokenizer as T
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import WhitespaceSplit
    tok = T(WordLevel({"a": 0, "b": 1, "[UNK]": 2}, unk_token="[UNK]"))
    tok.pre_tokenizer = WhitespaceSplit()
    rows = []
    for strategy in ("longest_first",):  # the reference's only_first / only_second deliberately differ from HF (truncate.cpp:92,97)
        for side in ("right", "left"):
            for max_length in (1, 2, 5, 8, 9):
                tok.enable_truncation(max_length, strategy=strategy, direction=side)
                for la in range(0, 13):
                    for lb in range(0, 13):
                        try:
                            enc = tok.encode(" ".join(["a"] * la), " ".join(["b"] * lb))
                        except Exception:  # HF refuses when the sequence to truncate cannot absorb the excess
                            continue
                        ka = sum(1 for s in enc.sequence_ids if s == 0)
                        kb = sum(1 for s in enc.sequence_ids if s == 1)
                        rows.append((["only_first", "only_second", "longest_first"].index(strategy), side == "left",
                                     max_length, la, lb, ka, kb))
    np.savez_compressed(G / "golden_truncate_hf.npz", rows=np.array(rows, np.int32))
    print("truncate rows", len(rows))


if __name__ == "__main__":
    main_truncate()
    main()
    main_wordpiece()
    main_llama3()
