"""GPU probe (round 5): the fused Llama-3 encode on texts of different make-up, per-kernel times from the library's own event brackets.
    python tools/l3_text_probe.py [rows] [bytes]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit  # noqa: E402
from tests.util import BpeTok  # noqa: E402
from tools.workloads import TextModel, ragged_rows  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    name = sys.argv[3] if len(sys.argv) > 3 else "llama3"
    lib = L.load()
    tok = BpeTok.load(name)
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**tok.attrs, lib=lib))
    for kind in ("zipf", "mixed", "uniform"):
        b, e, c = TextModel(7, kind).batch(rows, nbytes)
        rb, re_ = ragged_rows(rows)
        dev = [torch.as_tensor(a, device="cuda") for a in (rb, re_, b, e, c)]
        for _ in range(3):
            fused.evaluate(dev + [tok.pattern_u8()], tok.consts)
        torch.cuda.synchronize()
        lib.ovtk_profile_reset()
        lib.ovtk_profile_enable(1)
        for _ in range(5):
            fused.evaluate(dev + [tok.pattern_u8()], tok.consts)
        torch.cuda.synchronize()
        lib.ovtk_profile_enable(0)
        buf = C.create_string_buffer(16384)
        lib.ovtk_profile_dump(buf, 16384)
        t = {ln.split()[0]: float(ln.split()[1]) / max(1, int(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.strip()}
        print(kind, len(c), {k: round(v, 4) for k, v in t.items()}, flush=True)


if __name__ == "__main__":
    main()
