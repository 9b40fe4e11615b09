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
    print(f"host buffers (pageable numpy, fresh output arrays, H2D + encode + D2H per call): {dt * 1e3:.2f} ms per batch of "
          f"{len(c)} bytes = {len(c) / dt / 1e6:.0f} MB/s, {len(out[2])} ids")

    # The same call through the C ABI with PINNED host buffers that the caller reuses (what an adapter with a pinned
    # ov::Allocator, or a serving loop with its own staging buffers, hands over).
    import ctypes as C
    from openvino_tokenizers_amd import _lib as L
    lib = fused.bpe._lib
    pin = [torch.as_tensor(x).pin_memory() for x in (rb, re_, b, e, c)]
    o_b = torch.empty(rows, dtype=torch.int32).pin_memory()
    o_e = torch.empty(rows, dtype=torch.int32).pin_memory()
    o_i = torch.empty(len(c), dtype=torch.int32).pin_memory()
    rs = L.RaggedStrings(pin[0].data_ptr(), pin[1].data_ptr(), rows,
                         L.Strings(pin[2].data_ptr(), pin[3].data_ptr(), pin[4].data_ptr(), rows, len(c)))
    o = L.RaggedI32Out(o_b.data_ptr(), o_e.data_ptr(), o_i.data_ptr(), len(c), 0, 0)
    for _ in range(2):
        L.check(lib, lib.ovtk_encode_run(fused.split._h, fused.bpe._h, C.byref(rs), None, C.byref(o), L.MEM_HOST, None))
    t0 = time.perf_counter()
    for _ in range(n):
        L.check(lib, lib.ovtk_encode_run(fused.split._h, fused.bpe._h, C.byref(rs), None, C.byref(o), L.MEM_HOST, None))
    dt = (time.perf_counter() - t0) / n
    assert o.n_data == len(out[2]) and (o_i[: o.n_data].numpy() == out[2]).all()
    print(f"host buffers (pinned, reused, H2D + encode + D2H per call): {dt * 1e3:.2f} ms per batch = {len(c) / dt / 1e6:.0f} MB/s "
          f"({(len(c) + 8 * rows + 4 * o.n_data + 8 * rows) / dt / 1e9:.1f} GB/s over PCIe both ways)")


if __name__ == "__main__":
    main()
