"""Helpers shared by the parity tests."""
from __future__ import annotations

import numpy as np

from tools.harness import BpeTok, one_string_per_row, pack_strings  # noqa: F401


def assert_same(ref, got, host, what=""):
    assert len(ref) == len(got), what
    for i, (r, g) in enumerate(zip(ref, got)):
        g = host(g)
        assert r.shape == g.shape, f"{what} output {i}: shape {g.shape} != {r.shape}"
        if not np.array_equal(r, g):
            bad = int(np.flatnonzero(r != g)[0])
            raise AssertionError(f"{what} output {i}: first difference at {bad}: ref {r[bad:bad+8]} got {g[bad:bad+8]}")
