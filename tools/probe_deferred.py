"""Diagnostic: where wordpiece_deferred_kernel's time goes (config 3).  Needs the -DOVTK_PROBE build of the library (tools/probe_merge.py
says how).  Per wave: 0 start, 1 trie roots in LDS, 2 tile sums folded, 3 its (last) batch's entries loaded, 4 the store has answered,
5 out of the batch loop, 6 ticket drawn, 7 (last block only) the tile scan done.  wall_clock64 ticks are 10 ns."""
import ctypes as C, sys
from pathlib import Path
from types import SimpleNamespace
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvino_tokenizers_amd import _lib as L
import bench
lib = L.load(ROOT / "tools" / "build" / "libovtk_probe.so")
args = SimpleNamespace(config="3", rows=65536, bytes=512, batches=4)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = bench.WordpieceEncode(args, lib, dev, 0)
for i in range(8):
    wl.step(i)
torch.cuda.synchronize()
out = np.zeros((8192, 12), np.uint64)
names = ["start", "roots in LDS", "folded", "entries loaded", "store answered", "out of the loop", "ticket drawn", "tail done", "lookup loads back"]
for i in range(4):
    lib.ovtk_debug_probe_ops(None, 1)
    wl.step(i)
    torch.cuda.synchronize()
    lib.ovtk_debug_probe_ops(out.ctypes.data_as(C.POINTER(C.c_ulonglong)), 0)
    ts = out.astype(np.int64)
    live = ts[:, 0] > 0
    t0 = ts[live, 0].min()
    print(f"batch {i}: waves {live.sum()}")
    for k, name in enumerate(names):
        m = live & (ts[:, k] > 0)
        if not m.any():
            continue
        v = (ts[m, k] - t0) / 100.0
        print(f"  {name:16s} n={m.sum():5d}  min {v.min():6.1f}  p50 {np.median(v):6.1f}  p90 {np.percentile(v, 90):6.1f}  max {v.max():6.1f} us")
