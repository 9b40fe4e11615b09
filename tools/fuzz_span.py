"""GPU fuzz of lookup_span_kernel against the oracle: random batches of rows built from fragments that stress the block logic (rows of
every length from 1 byte to several blocks, blanks and delimiters at row edges, contractions, non-ASCII text, giant pieces), through
the fused GPT-2 / individual-digits / BERT-words paths and (round 5) the Llama-3 family's three patterns (Llama-3, Qwen2, tiktoken
cl100k: lookup_span_kernel<kSpanLlama3>, csrc/span_l3.hpp), two calls per batch (cold tables, then what they learned).
    python tools/fuzz_span.py [first_seed] [last_seed]
Prints one line per seed; exits 1 at the first difference."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, FusedSplitWordpiece, RegexSplit, WordpieceTokenizer  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.test_ops_parity import BERT_PUNCT, BERT_WS, bert_words, wp_consts  # noqa: E402
from tests.test_span_kernel import DIGITS_PATTERN, rows_of  # noqa: E402
from tests.util import BpeTok  # noqa: E402
from tools.make_tokenizers import load_tokenizer  # noqa: E402

L3_FRAG = ["\n", "\n\n", "\r\n", " \n", "\n    ", "\n\t", "123", "1234567", "12345678901234567890123456789012345", "'S", "'LL", "I'M", "they'Re", "\u00a0", "\u3000", "\u2028",
           "x²", "١٢٣", "ſ", "'ſ", "!\n\n", "...\n", "}\n\n", " \n \n  \n"]
# what DeepSeek-V3's pattern and o200k_base tell apart (span_fam.hpp): case, marks, slashes behind line breaks, control characters
FAM_FRAG = ["A", "AB", "HTTP", "Camel", "camelCase", "XMLHttpRequest", "aB", "ABc", "\u0301", "e\u0301", "!\u0301", "!!\u0301", " \u0301", "\u20dd", "\u01c5", "\u02b0",
            "日A", "A日B", "ПР", "пР", "A's", "a'T", "B'Re", "it's's", "\n/", "!\n/", "*/\n/*", "!\n/!\n/a", "\x01", "\x7f", "\u00ad", "\u200b", "1a", "12ab", "!ab", "!abé",
            "#tag", "@user", "_id", ".com", " !a", "!!a", "A" * 40, "ABCDEFGH" * 300]
FRAG = ["the", "token", "izer", " ", " ", " ", "  ", "\n", "\t", "a", "x", ",", ".", "!?", "don't", "we'll", "'", "'s", "I'm", "12", "2024", "1", "a1b2", "--", "(", ")",
        "naïve", "straße", "日本語", "Ωμέγα", "😀", "hello", "world", "un", "affable", "e.g.", " , ", "q" * 17, "word" * 5]


def rows(rng, n, frag=None):
    frag = frag or FRAG
    out = []
    for _ in range(n):
        k = int(rng.integers(0, 9))
        if k == 0:
            s = rng.choice([" ", "", "\n", ",", "a", " a", "a ", "  "])
        elif k == 1:
            s = "".join(rng.choice(frag, size=int(rng.integers(1, 6))))
        elif k == 2:
            s = " ".join(rng.choice(FRAG[:20], size=int(rng.integers(1, 80))))
        elif k == 3:
            s = "".join(rng.choice(frag, size=int(rng.integers(200, 900))))
        elif k == 4:
            s = rng.choice(["a", " ", "7", "日", ",", "\n"]) * int(rng.integers(1, 4200))
        elif k == 5:
            s = "ab " * int(rng.integers(1, 800))
        else:
            s = "".join(rng.choice(frag, size=int(rng.integers(1, 120))))
        out.append(s.encode())
    return out


def check(ref, got, what):
    for i, (r, g) in enumerate(zip(ref, got)):
        g = g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g)
        if r.shape != g.shape or not np.array_equal(r, g):
            print(f"DIFFERENCE: {what}, output {i}: shapes {r.shape} / {g.shape}")
            sys.exit(1)


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 20)
    lib = L.load()
    from tools.workloads import MODEL_PATTERNS
    gpt2, bert, llama3 = BpeTok.load("gpt2_small"), load_tokenizer("bert_small"), BpeTok.load("llama3_small")
    ws_pat, pu_pat = np.frombuffer(BERT_WS.encode(), np.uint8), np.frombuffer(BERT_PUNCT.encode(), np.uint8)
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        inputs = rows_of(rows(rng, int(rng.integers(300, 700))))
        dev = [torch.as_tensor(a, device="cuda") for a in inputs]
        for pattern in (gpt2.pattern, DIGITS_PATTERN):
            ref = gpt2.oracle()(*O.RegexSplit(pattern, "isolate")(*inputs)[:5])
            fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**gpt2.attrs, lib=lib))
            for call in range(2):
                check(ref, fused.evaluate(dev + [np.frombuffer(pattern.encode(), np.uint8)], gpt2.consts), f"seed {seed} BPE call {call}")
        ref = O.WordpieceTokenizer(bert["vocab"], bert["suffix_indicator"], bert["max_bytes_per_word"])(*bert_words(inputs), bert["unk_id"])
        fw = FusedSplitWordpiece(RegexSplit("remove", lib=lib), RegexSplit("isolate", lib=lib),
                                 WordpieceTokenizer(bert["suffix_indicator"], bert["max_bytes_per_word"], lib=lib))
        for call in range(3):
            check(ref, fw.evaluate(dev, ws_pat, pu_pat, wp_consts(bert)), f"seed {seed} BERT call {call}")
        inputs3 = rows_of(rows(rng, int(rng.integers(300, 700)), FRAG + L3_FRAG))
        dev3 = [torch.as_tensor(a, device="cuda") for a in inputs3]
        for name in ("llama3", "qwen2", "cl100k"):
            pattern = MODEL_PATTERNS.get(name, llama3.pattern)
            ref = llama3.oracle()(*O.RegexSplit(pattern, "isolate")(*inputs3)[:5])
            fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**llama3.attrs, lib=lib))
            for call in range(2):
                check(ref, fused.evaluate(dev3 + [np.frombuffer(pattern.encode(), np.uint8)], llama3.consts), f"seed {seed} {name} call {call}")
        inputs4 = rows_of(rows(rng, int(rng.integers(300, 700)), FRAG + L3_FRAG + FAM_FRAG))
        dev4 = [torch.as_tensor(a, device="cuda") for a in inputs4]
        for name in ("deepseek-v3", "o200k"):
            pattern = MODEL_PATTERNS[name]
            ref = llama3.oracle()(*O.RegexSplit(pattern, "isolate")(*inputs4)[:5])
            fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**llama3.attrs, lib=lib))
            for call in range(2):
                check(ref, fused.evaluate(dev4 + [np.frombuffer(pattern.encode(), np.uint8)], llama3.consts), f"seed {seed} {name} call {call}")
        print(f"seed {seed}: {len(inputs[0])} + {len(inputs3[0])} rows, {len(inputs[4])} + {len(inputs3[4])} bytes ok", flush=True)


if __name__ == "__main__":
    main()
