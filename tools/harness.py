"""Harness-side helpers shared by tests, bench.py and __graft_entry__ (not product code)."""
from __future__ import annotations

import numpy as np

from tools.make_tokenizers import load_tokenizer
from tools.workloads import ragged_rows

# The two RegexSplit patterns of a converted BERT pipeline (python/openvino_tokenizers/tokenizer_pipeline.py:392-435):
# whitespace is removed, punctuation / CJK characters are isolated.
BERT_WS = r"\s+"
BERT_PUNCT = "|".join([r"[!-/]", r"[:-@]", r"[\[-`]", r"[{-~]", r"[\p{P}]", r"[\x{4E00}-\x{9FFF}]", r"[\x{3400}-\x{4DBF}]",
                       r"[\x{20000}-\x{2A6DF}]", r"[\x{2A700}-\x{2B73F}]", r"[\x{2B740}-\x{2B81F}]",
                       r"[\x{2B820}-\x{2CEAF}]", r"[\x{F900}-\x{FAFF}]", r"[\x{2F800}-\x{2FA1F}]"])

def pack_strings(strings):
    """list of bytes/str -> (begins, ends, chars): python/openvino_tokenizers/utils.py:436-458."""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
    lens = np.fromiter((len(b) for b in bs), dtype=np.int64, count=len(bs))
    ends = np.cumsum(lens).astype(np.int32)
    return (ends - lens).astype(np.int32), ends, np.frombuffer(b"".join(bs), dtype=np.uint8).copy()


class BpeTok:
    """A BPE tokenizer in the form the ops receive it: constant inputs 5.. of BPETokenizer + attributes."""

    def __init__(self, vocab, merges, added=None, pattern=None, **attrs):
        self.vocab, self.merges, self.added, self.pattern, self.attrs = vocab, merges, added or {}, pattern, attrs
        consts = list(pack_strings(vocab))
        if merges and isinstance(merges[0], (tuple, list)):
            consts += list(pack_strings([m[0] for m in merges])) + list(pack_strings([m[1] for m in merges]))
        else:
            consts += list(pack_strings(merges))
        if self.added:
            consts += list(pack_strings(list(self.added.keys()))) + [np.asarray(list(self.added.values()), np.int32)]
        self.consts = consts

    @classmethod
    def load(cls, name):
        t = load_tokenizer(name)
        return cls(t["vocab"], t["merges"], t["added"], t["pattern"], **t["attrs"])

    def oracle(self):
        from oracle import oracle as O  # the checker; never reached from product code
        return O.BPETokenizer(self.vocab, self.merges, self.added, **self.attrs)

    def pattern_u8(self):
        return np.frombuffer(self.pattern.encode(), np.uint8)


def one_string_per_row(strings):
    b, e, c = pack_strings(strings)
    rb, re_ = ragged_rows(len(b))
    return [rb, re_, b, e, c]
