"""Diagnostic: per-step time of the pinned-host two-half calls as the streams age (fresh streams, 240 steps)."""
import ctypes as C, sys, time, os
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import argparse
import bench
from openvino_tokenizers_amd import _lib as L
dev = torch.device("cuda", 0)
if "bind" in sys.argv[1:]:
    print(bench.bind_to_gpu_numa_node(dev)[1])
lib = L.load()
args = argparse.Namespace(batches=4, tokenizer="gpt2", text="zipf", rows=65536, bytes=512, no_memo=False)
wl = bench.BpeEncode(args, lib, dev, 0, "gpt2", "zipf", 65536, 512, 1000, n_batches=4)
tb = wl.batches
def pin(t):
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True); p.copy_(t); return p
hin = [[pin(x) for x in tb.d[k]] for k in range(4)]
rs = [L.RaggedStrings(h[0].data_ptr(), h[1].data_ptr(), tb.rows, L.Strings(h[2].data_ptr(), h[3].data_ptr(), h[4].data_ptr(), tb.rows, h[4].numel())) for h in hin]
outs = []
NOUT = int(os.environ.get("NOUT", 6))
for _ in range(NOUT):
    b = torch.empty(tb.rows, dtype=torch.int32, pin_memory=True); e = torch.empty(tb.rows, dtype=torch.int32, pin_memory=True)
    ids = torch.empty(tb.cap, dtype=torch.int32, pin_memory=True)
    outs.append((b, e, ids, L.RaggedI32Out(b.data_ptr(), e.data_ptr(), ids.data_ptr(), tb.cap, 0, 0)))
mode = sys.argv[1] if len(sys.argv) > 1 else ""
for rep in range(int(os.environ.get("AGE_REPS", 2))):
    streams = [torch.cuda.Stream(dev) for _ in range(4)]
    ptrs = [C.c_void_p(s.cuda_stream) for s in streams]
    inflight, stamps = [], []
    for i in range(int(os.environ.get("AGE_STEPS", 240))):
        o = outs[i % NOUT]
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_enqueue_host(wl.split._h, wl.bpe._h, C.byref(rs[i % 4]), None, C.byref(o[3]), ptrs[i % 4], C.byref(pending)))
        inflight.append((pending, o))
        if len(inflight) > 3:
            p, oo = inflight.pop(0)
            L.check(lib, lib.ovtk_encode_finish(p, C.byref(oo[3])))
            stamps.append(time.perf_counter())
        if mode == "sync" and i % 40 == 39:
            torch.cuda.synchronize()
    for p, oo in inflight:
        L.check(lib, lib.ovtk_encode_finish(p, C.byref(oo[3])))
    d = np.diff(np.array(stamps)) * 1e3
    print(f"rep {rep}: ms/step by 20-step window:", " ".join(f"{d[k:k+20].mean():.2f}" for k in range(0, len(d) - 19, 20)))
