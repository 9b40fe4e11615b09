"""Seeded synthetic text batches for the parity tests and bench.py.

Harness code (not product, not oracle).  Everything is numpy-vectorised so that the 32 MiB
config-2 batch (65 536 rows x ~512 B) is generated in a few seconds on the GPU box.

Text model ("zipf"): words drawn Zipf(s=1.1) from a seeded 50 000-word lexicon (lengths 1..14,
log-normal, English-like letter frequencies), ~10 % capitalised, separated by one space, with
2 % irregular whitespace (double space / newline / tab / space+newline), 8 % digit runs and
10 % punctuation (70 % of it glued to the previous word), a few contractions ('s 't 're ...).
Stress model ("uniform"): independent uniformly random printable ASCII bytes.
Mixed-Unicode model ("mixed"): as "zipf" but 30 % of the words come from Latin-1 / Greek /
Cyrillic / CJK / Hiragana / emoji lexicons built only from code points whose General_Category
has been stable since Unicode 13.
Row r is `round(N(target, 0.1*target))` bytes, cut from the stream at a UTF-8 boundary.
"""
from __future__ import annotations

import numpy as np

_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_P = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0,
                      1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.10, 0.07])
_LETTER_P = _LETTER_P / _LETTER_P.sum()

_PUNCT = [b",", b".", b"!", b"?", b";", b":", b"-", b"(", b")", b'"', b"'", b"...", b"--", b"/", b"%", b"$", b"&",
          b"*", b"@", b"#"]
_PUNCT_P = np.array([30, 30, 4, 4, 3, 3, 5, 3, 3, 4, 3, 1, 1, 1, 1, 1, 0.5, 0.5, 0.5, 0.5])
_PUNCT_P = _PUNCT_P / _PUNCT_P.sum()
_CONTRACTIONS = [b"'s", b"'t", b"'re", b"'ve", b"'m", b"'ll", b"'d"]
_ODD_WS = [b"  ", b"\n", b"\t", b" \n", b"\n\n", b"   ", b" \t"]

_UNI_RANGES = {  # inclusive code point ranges, General_Category stable across Unicode 13..16
    "latin1": [(0x00C0, 0x00D6), (0x00D8, 0x00F6), (0x00F8, 0x00FF)],
    "greek": [(0x0391, 0x03A1), (0x03A3, 0x03A9), (0x03B1, 0x03C9)],
    "cyrillic": [(0x0410, 0x044F)],
    "cjk": [(0x4E00, 0x9FA5)],
    "hiragana": [(0x3041, 0x3096)],
    "emoji": [(0x1F600, 0x1F64F)],
}


class AtomTable:
    """Flat byte pool + offsets for a list of byte strings."""

    def __init__(self, items):
        lens = np.fromiter((len(x) for x in items), dtype=np.int64, count=len(items))
        self.off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        self.len = lens
        self.pool = np.frombuffer(b"".join(items), dtype=np.uint8).copy()
        self.n = len(items)


def make_lexicon(rng: np.random.Generator, n_words: int = 50_000):
    """Word r (0 = most frequent) has a log-normal length whose median grows with log(rank):
    ~2.5 letters for the top ranks, ~8 for the tail, as in natural language."""
    rank = np.arange(1, n_words + 1, dtype=np.float64)
    median = 2.4 + 0.60 * np.log(rank)
    lens = np.clip(np.round(rng.lognormal(mean=np.log(median), sigma=0.35)), 1, 14).astype(np.int64)
    letters = rng.choice(_LETTERS, size=int(lens.sum()), p=_LETTER_P)
    off = np.concatenate([[0], np.cumsum(lens)])
    return [letters[off[i]:off[i + 1]].tobytes() for i in range(n_words)]


def _unicode_lexicon(rng, script, n_words):
    cps = np.concatenate([np.arange(a, b + 1) for a, b in _UNI_RANGES[script]])
    max_len = {"cjk": 4, "emoji": 2, "hiragana": 6}.get(script, 10)
    lens = np.clip(np.round(rng.lognormal(mean=1.2, sigma=0.45, size=n_words)), 1, max_len).astype(np.int64)
    return ["".join(map(chr, rng.choice(cps, size=int(l)))).encode("utf-8") for l in lens]


def _zipf_p(n, s=1.1):
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** s
    return p / p.sum()


class TextModel:
    def __init__(self, seed: int = 1234, kind: str = "zipf", n_words: int = 50_000):
        assert kind in ("zipf", "mixed", "uniform")
        self.kind = kind
        self.seed = seed
        rng = np.random.default_rng(np.random.PCG64(seed))
        if kind == "uniform":
            return
        words = make_lexicon(rng, n_words)
        caps = [w[:1].upper() + w[1:] for w in words[:5000]]
        digits = [b"".join(bytes([48 + int(d)]) for d in rng.integers(0, 10, size=int(l)))
                  for l in np.clip(rng.geometric(0.45, size=2000), 1, 8)]
        groups = {"word": words, "cap": caps, "digit": digits, "punct": list(_PUNCT), "contr": list(_CONTRACTIONS),
                  "sp": [b" "], "oddws": list(_ODD_WS), "none": [b""]}
        if kind == "mixed":
            for script in _UNI_RANGES:
                groups[script] = _unicode_lexicon(rng, script, 4000 if script != "emoji" else 80)
        self.base = {}
        items = []
        for g, lst in groups.items():
            self.base[g] = len(items)
            items.extend(lst)
        self.gsize = {g: len(lst) for g, lst in groups.items()}
        self.atoms = AtomTable(items)
        self.p_word = _zipf_p(len(words))
        self.p_cap = _zipf_p(len(caps))

    # ------------------------------------------------------------------ stream generation
    def _slots(self, rng, n_slots):
        """Returns (sep_item, atom_item) index arrays of length n_slots."""
        b, gs = self.base, self.gsize
        u = rng.random(n_slots)
        atom = np.empty(n_slots, np.int64)
        sep = np.full(n_slots, b["sp"], np.int64)
        is_punct = u < 0.10
        is_digit = (u >= 0.10) & (u < 0.18)
        is_contr = (u >= 0.18) & (u < 0.19)
        is_cap = (u >= 0.19) & (u < 0.28)
        is_word = u >= 0.28
        atom[is_punct] = b["punct"] + rng.choice(gs["punct"], size=int(is_punct.sum()), p=_PUNCT_P)
        atom[is_digit] = b["digit"] + rng.integers(0, gs["digit"], size=int(is_digit.sum()))
        atom[is_contr] = b["contr"] + rng.integers(0, gs["contr"], size=int(is_contr.sum()))
        atom[is_cap] = b["cap"] + rng.choice(gs["cap"], size=int(is_cap.sum()), p=self.p_cap)
        nw = int(is_word.sum())
        widx = b["word"] + rng.choice(gs["word"], size=nw, p=self.p_word)
        if self.kind == "mixed":
            v = rng.random(nw)
            scripts = list(_UNI_RANGES)
            share = 0.30 / len(scripts)
            for k, script in enumerate(scripts):
                m = (v >= k * share) & (v < (k + 1) * share)
                widx[m] = b[script] + rng.integers(0, gs[script], size=int(m.sum()))
        atom[is_word] = widx
        # separators: glued punctuation / contractions, 2 % odd whitespace
        glue = (is_punct & (rng.random(n_slots) < 0.70)) | is_contr
        sep[glue] = b["none"]
        odd = (~glue) & (rng.random(n_slots) < 0.02)
        sep[odd] = b["oddws"] + rng.integers(0, gs["oddws"], size=int(odd.sum()))
        return sep, atom

    def stream(self, rng, n_bytes: int) -> np.ndarray:
        """At least n_bytes of text as a uint8 array."""
        if self.kind == "uniform":
            return rng.integers(32, 127, size=n_bytes, dtype=np.uint8)
        out, have = [], 0
        while have < n_bytes:
            n_slots = max(1024, int((n_bytes - have) / 5.0) + 1024)
            n_slots = min(n_slots, 2_000_000)
            sep, atom = self._slots(rng, n_slots)
            items = np.empty(2 * n_slots, np.int64)
            items[0::2] = sep
            items[1::2] = atom
            lens = self.atoms.len[items]
            total = int(lens.sum())
            starts = np.cumsum(lens) - lens
            owner = np.repeat(np.arange(len(items), dtype=np.int64), lens)
            pos = np.arange(total, dtype=np.int64) - starts[owner]
            buf = self.atoms.pool[self.atoms.off[items][owner] + pos]
            out.append(buf)
            have += total
        return np.concatenate(out) if len(out) > 1 else out[0]

    # ------------------------------------------------------------------ batches
    def batch(self, n_rows: int, target_len: int, seed: int | None = None):
        """Returns (begins i32[n_rows], ends i32[n_rows], chars u8[total]) with contiguous rows."""
        rng = np.random.default_rng(np.random.PCG64(self.seed * 7919 + 17 if seed is None else seed))
        want = np.maximum(1, np.round(rng.normal(target_len, 0.1 * target_len, size=n_rows))).astype(np.int64)
        total = int(want.sum())
        text = self.stream(rng, total + 8 * n_rows + 64)
        cuts = np.concatenate([[0], np.cumsum(want)])
        if self.kind == "mixed":
            # move every cut forward to the next UTF-8 lead byte so that rows are valid UTF-8
            is_lead = (text & 0xC0) != 0x80
            lead_pos = np.flatnonzero(is_lead)
            cuts = lead_pos[np.searchsorted(lead_pos, cuts, side="left")]
            cuts[0] = 0
        chars = text[: int(cuts[-1])].copy()
        begins = cuts[:-1].astype(np.int32)
        ends = cuts[1:].astype(np.int32)
        assert ends[-1] == len(chars) and len(chars) < 2**31
        return begins, ends, chars

    def corpus_lines(self, n_bytes: int, line_len: int = 2048, seed: int = 99):
        """Training corpus for the in-process tokenizers: list[str] lines."""
        b, e, c = self.batch(max(1, n_bytes // line_len), line_len, seed=seed)
        raw = c.tobytes()
        return [raw[x:y].decode("utf-8") for x, y in zip(b.tolist(), e.tolist())]


def ragged_rows(n_rows: int):
    """ragged_begins/ends for one string per row (tokenizer_pipeline.py:1668-1676)."""
    r = np.arange(n_rows + 1, dtype=np.int32)
    return r[:-1].copy(), r[1:].copy()


# Split patterns of current tokenizer.json files (the tables of a shaped tokenizer can be run behind any of them: bench.py --pattern).
MODEL_PATTERNS = {
    "cl100k": (r"'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}++|\p{N}{1,3}+| ?[^\s\p{L}\p{N}]++[\r\n]*+|\s++$|\s*[\r\n]|"
               r"\s+(?!\S)|\s+"),
    "qwen2": r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+",
    "o200k": (r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?|"
              r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?|\p{N}{1,3}|"
              r" ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+"),
    "deepseek-v3": (r"""[!"#$%&'()*+,\-./:;<=>?@\[\\\]^_`{|}~][A-Za-z]+|[^\r\n\p{L}\p{P}\p{S}]?[\p{L}\p{M}]+| ?[\p{P}\p{S}]+[\r\n]*|"""
                    r"\s*[\r\n]+|\s+(?!\S)|\s+"),
}
