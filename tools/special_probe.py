"""GPU probe (round 5): the special-tokens stage of ovtk_encode_special_run on text with / without special tokens, per-kernel times from
the library's own event brackets.   python tools/special_probe.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.ops import BPETokenizer, FusedSpecialSplitBPE, RegexSplit, SpecialTokensSplit  # noqa: E402
from tests.util import BpeTok  # noqa: E402
from tools.workloads import TextModel, ragged_rows  # noqa: E402


def main():
    rows, nbytes = 65536, 512
    lib = L.load()
    tok = BpeTok.load("gpt2")
    pat = np.frombuffer(r"(\<\|endoftext\|\>)".encode(), np.uint8)
    fused = FusedSpecialSplitBPE(SpecialTokensSplit(lib=lib), RegexSplit("isolate", lib=lib), BPETokenizer(**tok.attrs, lib=lib))
    b, e, c = TextModel(7, "zipf").batch(rows, nbytes)
    rb, re_ = ragged_rows(rows)
    sp = np.frombuffer(b"<|endoftext|>", np.uint8)
    for frac in (0.0, 0.01, 0.1, 1.0):
        cc = c.copy()
        rng = np.random.default_rng(3)
        for i in np.flatnonzero(rng.random(rows) < frac):
            at = int(b[i]) + 100
            cc[at:at + len(sp)] = sp
        dev = [torch.as_tensor(a, device="cuda") for a in (rb, re_, b, e, cc)]
        for _ in range(3):
            fused.evaluate(dev + [pat], tok.pattern_u8(), tok.consts)
        torch.cuda.synchronize()
        lib.ovtk_profile_reset()
        lib.ovtk_profile_enable(1)
        for _ in range(5):
            fused.evaluate(dev + [pat], tok.pattern_u8(), tok.consts)
        torch.cuda.synchronize()
        lib.ovtk_profile_enable(0)
        buf = C.create_string_buffer(16384)
        lib.ovtk_profile_dump(buf, 16384)
        t = {ln.split()[0]: float(ln.split()[1]) / max(1, int(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.strip()}
        print(frac, {k: round(v, 4) for k, v in t.items()}, flush=True)


if __name__ == "__main__":
    main()
