"""Diagnostic: where merge_kernel's time goes.  Needs the -DOVTK_PROBE build of the library:
  cd openvino_tokenizers_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -w -DOVTK_PROBE -shared \\
      -o ../../tools/build/libovtk_probe.so api_encode.cpp api_ops.cpp tables.cpp runtime.cpp regex_compile.cpp
Per wave: 0 start, 1 tile sums folded, 2 symbols of its (last) batch in LDS, 3 merges of that batch done, 4 out of the batch
loop, 5/6 (last block only) tail start / scan done.  wall_clock64 ticks are 10 ns."""
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
args = SimpleNamespace(config=a.config, tokenizer="gpt2", text="zipf", rows=65536 if a.config != "4" else 131072, bytes=512, batches=4, no_memo=a.no_memo, pattern=None, cache_capacity=None)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = bench.make_workload(args, lib, dev, 0)
for i in range(8):
    wl.step(i)
torch.cuda.synchronize()
out = np.zeros((8192, 12), np.uint64)
for i in range(4):
    lib.ovtk_debug_probe(None, 1)
    wl.step(i)
    torch.cuda.synchronize()
    lib.ovtk_debug_probe(out.ctypes.data_as(C.POINTER(C.c_ulonglong)), 0)
    ts = out.astype(np.int64)
    live = ts[:, 0] > 0
    t0 = ts[live, 0].min()
    def stat(k):
        m = live & (ts[:, k] > 0)
        v = (ts[m, k] - t0) / 100.0
        return f"n={m.sum()} min {v.min():.1f} p50 {np.median(v):.1f} p90 {np.percentile(v, 90):.1f} max {v.max():.1f}" if m.any() else "-"

    def phase(k):   # accumulated over the wave's batches
        v = ts[live, k] / 100.0
        return f"p50 {np.median(v):.1f} p90 {np.percentile(v, 90):.1f} max {v.max():.1f} us"
    m = live & (ts[:, 8] > 0)
    print(f"batch {i}: waves {live.sum()}; start {stat(0)}; folded {stat(1)}; out of the batch loop {stat(4)}\n"
          f"  per wave, summed over its batches: entries + symbols (F) {phase(2)}; merges (F) {phase(3)}; path L {phase(10)}; path W {phase(11)}\n"
          + (f"  largest symbol count per wave p50 {np.median(ts[m, 8])}; merge steps per wave (last batch) p50 {np.median(ts[m, 9])} max {ts[m, 9].max()}"
             if m.any() else "  no wave had anything to merge: every deferred piece was in the store")
          + f"\n  tail (last block): start {stat(5)}; scan done {stat(6)}")
