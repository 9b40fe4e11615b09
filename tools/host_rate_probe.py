"""PCIe-inclusive rate of the fused encode: the same config-2 batch handed over as HOST buffers (numpy), i.e. what the
OpenVINO CPU-plugin adapter of INTEGRATION.md pays per evaluate() call.  Reported in DESIGN.md, never as bench `value`."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401  (shares the HIP runtime)

from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit  # noqa: E402
from tools.harness import BpeTok  # noqa: E402
from tools.workloads import TextModel, ragged_rows  # noqa: E402


def main():
    tok = BpeTok.load("gpt2")
    rows = 65536
    b, e, c = TextModel(1234, "zipf").batch(rows, 512, seed=1000)
    rb, re_ = ragged_rows(rows)
    fused = FusedSplitBPE(RegexSplit("isolate"), BPETokenizer(**tok.attrs))
    pat = tok.pattern_u8()
    for _ in range(2):
        fused.evaluate([rb, re_, b, e, c, pat], tok.consts)
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        out = fused.evaluate([rb, re_, b, e, c, pat], tok.consts)
    dt = (time.perf_counter() - t0) / n
    print(f"host buffers (pageable numpy, H2D + encode + D2H per call): {dt * 1e3:.2f} ms per batch of {len(c)} bytes "
          f"= {len(c) / dt / 1e6:.0f} MB/s, {len(out[2])} ids")


if __name__ == "__main__":
    main()
