"""Temporary: where merge_kernel's time goes (wall_clock64 probes; needs the -DOVTK_PROBE build in tools/build)."""
import ctypes as C, sys, argparse
from pathlib import Path
from types import SimpleNamespace
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvino_tokenizers_amd import _lib as L
import bench
lib = L.load(ROOT / "tools" / "build" / "libovtk_probe.so")
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="2"); ap.add_argument("--no-memo", action="store_true")
a = ap.parse_args()
args = SimpleNamespace(config=a.config, tokenizer="gpt2", text="zipf", rows=65536, bytes=512, batches=4, no_memo=a.no_memo)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = bench.make_workload(args, lib, dev, 0)
for i in range(6):
    wl.step(i)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
for i in range(6):
    lib.ovtk_debug_probe(None, 1)
    wl.step(i)
    torch.cuda.synchronize()
    lib.ovtk_debug_probe(out, 0)
    t0 = out[0]
    us = lambda k: (out[k] - t0) / 100.0
    print(f"batch {i}: fold done(max) {us(1):.1f}  first block out of batches {us(2):.1f}  last block out {us(3):.1f}  tail start {us(4):.1f} "
          f"exact done {us(5):.1f}  scan done {us(6):.1f}  n_exact {out[7]}")
