"""Differential fuzzer for regex_compile.cpp on the host: random patterns rich in look-around over several characters, atomic groups and
possessive groups (round 6), every string over a small alphabet up to length 4 plus random longer ones, the compiled tables
(tools/regex_host_check.cpp) against the system's PCRE2.  Build the checker first (see its header).
    python tools/fuzz_regex_host.py SEED N_PATTERNS"""
import itertools, subprocess, sys
import numpy as np
from pathlib import Path
CHECK = Path(__file__).resolve().parent / "build" / "regex_host_check"
ATOMS = ["a", "b", "c", " ", r"\n", ".", r"\s", r"\S", r"\d", r"\w", r"\W", "[ab]", "[^a]", "[a-c]", r"[^\s]", r"\p{L}", r"\P{L}", r"\p{N}",
         "1", "é", "[é1]", r"[\s\d]", r"\b", r"\B", "^", "$", r"\z", r"\A"]
QUANT = ["", "", "", "*", "+", "?", "{1,2}", "{2}", "{0,2}", "*?", "+?", "??", "*+", "++", "?+", "{1,2}?", "{1,2}+"]
ALPHA = ["a", "b", "c", " ", "\n", "1", "é", "A"]
def fixed(rng, n=None):
    n = n or int(rng.integers(1, 4))
    return "".join(ATOMS[rng.integers(18)] for _ in range(n))
def gen(rng, depth=0):
    r = rng.random()
    if depth >= 3 or r < 0.40:
        a = ATOMS[rng.integers(len(ATOMS))]
        if a in (r"\b", r"\B", "^", "$", r"\z", r"\A"):
            return a
        return a + QUANT[rng.integers(len(QUANT))]
    if r < 0.58:
        return gen(rng, depth + 1) + gen(rng, depth + 1)
    if r < 0.70:
        return "(?:" + gen(rng, depth + 1) + "|" + gen(rng, depth + 1) + ")" + QUANT[rng.integers(len(QUANT))]
    if r < 0.76:
        return "(" + gen(rng, depth + 1) + ")" + QUANT[rng.integers(len(QUANT))]
    if r < 0.84:   # look-ahead over anything
        return ["(?=", "(?!"][rng.integers(2)] + gen(rng, depth + 1) + ")"
    if r < 0.90:   # look-behind over fixed sequences
        alts = "|".join(fixed(rng) for _ in range(int(rng.integers(1, 3))))
        return ["(?<=", "(?<!"][rng.integers(2)] + alts + ")"
    if r < 0.97:   # atomic
        return "(?>" + gen(rng, depth + 1) + ")" + QUANT[rng.integers(len(QUANT))]
    return "(?i:" + gen(rng, depth + 1) + ")"
seed = int(sys.argv[1]); n = int(sys.argv[2])
rng = np.random.default_rng(seed)
strings = ["".join(t) for k in range(1, 5) for t in itertools.product(ALPHA, repeat=k)]
strings += ["".join(rng.choice(ALPHA, size=int(k))) for k in rng.integers(5, 14, size=600)]
cnt = {0: 0, 1: 0, 2: 0, 3: 0}
why = {}
for i in range(n):
    pat = gen(rng)
    r = subprocess.run([str(CHECK), pat] + strings, capture_output=True, text=True, errors="replace")
    cnt[r.returncode] = cnt.get(r.returncode, 0) + 1
    if r.returncode == 1 or r.returncode not in (0, 2, 3):
        print("BAD", repr(pat), r.returncode, r.stdout[:300])
    if r.returncode == 2:
        k = r.stdout.split("(")[1].split(")")[0] if "(" in r.stdout else r.stdout
        why[k] = why.get(k, 0) + 1
print("seed", seed, "same", cnt[0], "BAD", cnt[1], "unsupported", cnt[2], "pcre2-invalid both", cnt[3], "other", {k: v for k, v in cnt.items() if k not in (0, 1, 2, 3)})
for k, v in sorted(why.items(), key=lambda x: -x[1]): print("   ", v, k)
