"""Helpers shared by the parity tests."""
from __future__ import annotations

import numpy as np

from oracle import oracle as O
from tools.make_tokenizers import load_tokenizer
from tools.workloads import ragged_rows


class BpeTok:
    """A BPE tokenizer in the form the ops receive it (constant inputs + attributes)."""

    def __init__(self, vocab, merges, added=None, pattern=None, **attrs):
        self.vocab, self.merges, self.added, self.pattern, self.attrs = vocab, merges, added or {}, pattern, attrs
        pk = O.pack_strings
        consts = list(pk(vocab))
        if merges and isinstance(merges[0], (tuple, list)):
            consts += list(pk([m[0] for m in merges])) + list(pk([m[1] for m in merges]))
        else:
            consts += list(pk(merges))
        if self.added:
            consts += list(pk(list(self.added.keys()))) + [np.asarray(list(self.added.values()), np.int32)]
        self.consts = consts

    @classmethod
    def load(cls, name):
        t = load_tokenizer(name)
        return cls(t["vocab"], t["merges"], t["added"], t["pattern"], **t["attrs"])

    def oracle(self):
        return O.BPETokenizer(self.vocab, self.merges, self.added, **self.attrs)

    def pattern_u8(self):
        return np.frombuffer(self.pattern.encode(), np.uint8)


def one_string_per_row(strings):
    b, e, c = O.pack_strings(strings)
    rb, re_ = ragged_rows(len(b))
    return [rb, re_, b, e, c]


def assert_same(ref, got, host, what=""):
    assert len(ref) == len(got), what
    for i, (r, g) in enumerate(zip(ref, got)):
        g = host(g)
        assert r.shape == g.shape, f"{what} output {i}: shape {g.shape} != {r.shape}"
        if not np.array_equal(r, g):
            bad = int(np.flatnonzero(r != g)[0])
            raise AssertionError(f"{what} output {i}: first difference at {bad}: ref {r[bad:bad+8]} got {g[bad:bad+8]}")
