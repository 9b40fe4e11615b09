"""Probe of the pinned-host two-half calls: where does a step's time go (enqueue vs finish), by streams / depth."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import argparse  # noqa: E402

import bench  # noqa: E402
from openvino_tokenizers_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
lib = L.load()
args = argparse.Namespace(batches=4, tokenizer="gpt2", text="zipf", rows=65536, bytes=512, no_memo=False)
wl = bench.BpeEncode(args, lib, dev, 0, "gpt2", "zipf", 65536, 512, 1000, n_batches=4)
tb = wl.batches


def pin(t):
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    return p


hin = [[pin(x) for x in tb.d[k]] for k in range(4)]
rs = [L.RaggedStrings(h[0].data_ptr(), h[1].data_ptr(), tb.rows, L.Strings(h[2].data_ptr(), h[3].data_ptr(), h[4].data_ptr(), tb.rows, h[4].numel()))
      for h in hin]
outs = []
for _ in range(6):
    b = torch.empty(tb.rows, dtype=torch.int32, pin_memory=True)
    e = torch.empty(tb.rows, dtype=torch.int32, pin_memory=True)
    ids = torch.empty(tb.cap, dtype=torch.int32, pin_memory=True)
    outs.append((b, e, ids, L.RaggedI32Out(b.data_ptr(), e.data_ptr(), ids.data_ptr(), tb.cap, 0, 0)))
for n_streams, depth in [(1, 0), (2, 1), (3, 2), (4, 3), (3, 1)]:
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    ptrs = [C.c_void_p(s.cuda_stream) for s in streams]
    t_enq, t_fin = [], []

    def loop(n):
        inflight = []
        for i in range(n):
            o = outs[i % len(outs)]
            pending = C.c_void_p()
            t0 = time.perf_counter()
            L.check(lib, lib.ovtk_encode_enqueue_host(wl.split._h, wl.bpe._h, C.byref(rs[i % 4]), None, C.byref(o[3]), ptrs[i % n_streams], C.byref(pending)))
            t_enq.append(time.perf_counter() - t0)
            inflight.append((pending, o))
            if len(inflight) > depth:
                p, oo = inflight.pop(0)
                t0 = time.perf_counter()
                L.check(lib, lib.ovtk_encode_finish(p, C.byref(oo[3])))
                t_fin.append(time.perf_counter() - t0)
        for p, oo in inflight:
            L.check(lib, lib.ovtk_encode_finish(p, C.byref(oo[3])))
    loop(8)
    t_enq.clear()
    t_fin.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(24)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 24
    print(f"streams {n_streams} depth {depth}: {dt * 1e3:.3f} ms/step = {tb.n_chars[0] / dt / 1e9:.1f} GB/s; enqueue {np.mean(t_enq) * 1e3:.3f} ms, "
          f"finish {np.mean(t_fin) * 1e3:.3f} ms")
