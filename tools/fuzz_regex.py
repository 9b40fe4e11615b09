"""Differential fuzzer for the compiled-pattern RegexSplit: random patterns from a small grammar, every string over a small
alphabet up to length 4 plus random longer ones, the emulator build of the device matcher against PCRE2 (the oracle).
Patterns either side refuses are skipped; a mismatch prints the pattern and the first offending string.
    python tools/fuzz_regex.py [seed] [n_patterns]"""
import itertools
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvino_tokenizers_amd import _lib as L  # noqa: E402
from tests.conftest import Backend  # noqa: E402
from tests.test_split_rules import check  # noqa: E402

ATOMS = ["a", "b", "c", " ", r"\n", ".", r"\s", r"\S", r"\d", r"\w", r"\W", "[ab]", "[^a]", "[a-c]", r"[^\s]", r"\p{L}", r"\P{L}", r"\p{N}",
         "1", "é", "[é1]", r"[\s\d]", r"\b", r"\B", "^", "$", r"\z", r"\A"]
if "wide" in sys.argv[3:]:
    ATOMS = ATOMS[:18] + ["A", r"\h", r"\v", r"\N", "[[:alpha:]]", r"\p{Lu}", r"\p{Ll}", "元", r"(?s:.)", "_", r"[^\r\n]", r"\r", r"\t", r"[\p{L}\p{N}]",
                         r"[^\s\p{L}\p{N}]", "'", r"\b", r"\B", "^", "$", r"\z", r"\Z", r"\A"]
if "scripts" in sys.argv[3:]:
    # script properties (round 5).  The oracle's PCRE2 10.39 reads a bare name as the Script property, the product (10.46's meaning)
    # as Script_Extensions: the alphabet holds only characters on which the two agree (letters of ONE script)
    ATOMS = ATOMS[:18] + [r"\p{Greek}", r"\p{Cyrillic}", r"\P{Greek}", r"[\p{Greek}\p{Han}]", r"\p{Han}", r"[^\p{Cyrillic}a]", r"\b", r"\B", "^", "$", r"\z", r"\A"]
QUANT = ["", "", "", "*", "+", "?", "{1,2}", "{2}", "{0,2}", "*?", "+?", "??", "*+", "++", "?+", "{1,2}?", "{1,2}+"]
ALPHA = ["a", "b", "c", " ", "\n", "1", "é", "A"] + (["元", "\r", "\t", "_", "'"] if "wide" in sys.argv[3:] else []) + \
        (["α", "я", "元"] if "scripts" in sys.argv[3:] else [])


def gen(rng, depth=0):
    r = rng.random()
    if depth >= 3 or r < 0.45:
        a = ATOMS[rng.integers(len(ATOMS))]
        if a in (r"\b", r"\B", "^", "$", r"\z", r"\A"):
            return a
        return a + QUANT[rng.integers(len(QUANT))]
    if r < 0.65:
        return gen(rng, depth + 1) + gen(rng, depth + 1)
    if r < 0.80:
        return "(?:" + gen(rng, depth + 1) + "|" + gen(rng, depth + 1) + ")" + QUANT[rng.integers(len(QUANT))]
    if r < 0.88:
        return "(" + gen(rng, depth + 1) + ")" + QUANT[rng.integers(len(QUANT))]
    if r < 0.91:
        kind = ["(?=", "(?!", "(?<=", "(?<!"][rng.integers(4)]
        inner = ATOMS[rng.integers(18)]  # single-character atoms only (fixed-width look-behind)
        return kind + inner + ")"
    if r < 0.94:   # round 6: look-ahead over anything, look-behind over alternatives of fixed sequences, atomic groups
        k = rng.integers(3)
        if k == 0:
            return ["(?=", "(?!"][rng.integers(2)] + gen(rng, depth + 1) + ")"
        if k == 1:
            alts = "|".join("".join(ATOMS[rng.integers(18)] for _ in range(int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 3))))
            return ["(?<=", "(?<!"][rng.integers(2)] + alts + ")"
        return "(?>" + gen(rng, depth + 1) + ")" + QUANT[rng.integers(len(QUANT))]
    return "(?i:" + gen(rng, depth + 1) + ")"


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n_pat = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(seed)
    lib = L.load(ROOT / "tests" / "emu" / "build" / "libovtk_emu.so")
    backend = Backend("emu", lib)
    strings = ["".join(t) for k in range(1, 4 if len(ALPHA) <= 8 else 3) for t in itertools.product(ALPHA, repeat=k)]
    strings += ["".join(rng.choice(ALPHA, size=3)) for _ in range(600)] if len(ALPHA) > 8 else []
    strings += ["".join(rng.choice(ALPHA, size=int(k))) for k in rng.integers(4, 12, size=400)] + [""]
    done = refused = bad = 0
    behaviours = ["isolate", "remove", "mergedwithprevious", "mergedwithnext", "contiguous"]
    for i in range(n_pat):
        pat = gen(rng)
        beh = behaviours[rng.integers(len(behaviours))]
        inv = bool(rng.integers(2)) and beh in ("isolate", "remove")
        try:
            check(backend, pat, strings, beh, inv)
            done += 1
        except L.OvtkError as e:
            if e.code != L.E_UNSUPPORTED:
                print("ERROR", repr(pat), e)
                bad += 1
            refused += 1
        except AssertionError as e:
            print("MISMATCH", repr(pat), beh, inv, str(e)[:300])
            bad += 1
        except Exception as e:  # the oracle's PCRE2 refused the pattern
            refused += 1
            if "compil" not in str(e).lower() and "pcre" not in str(e).lower():
                print("??", repr(pat), type(e).__name__, str(e)[:200])
    print(f"seed {seed}: {done} patterns compared, {refused} refused by one side, {bad} BAD")


if __name__ == "__main__":
    main()
