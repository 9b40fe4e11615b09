"""Host-side cost of every piece of ShardExchange.submit() in a one-rank RCCL group (debug tool)."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.distributed import ShardExchange  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    lib = L.load()
    rows, per = 65536, 112
    lens = np.full(rows, per, np.int64)
    ends = torch.as_tensor(np.cumsum(lens).astype(np.int32), device=dev)
    begins = ends - per
    ids = torch.randint(0, 50000, (rows * per,), dtype=torch.int32, device=dev)
    ex = ShardExchange(rows, 50257, dev, lib=lib)
    for it in range(3):
        ex.submit(begins, ends, ids)
    ex.flush()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    host = 0.0
    for it in range(n):
        t1 = time.perf_counter()
        ex.submit(begins, ends, ids)
        host += time.perf_counter() - t1
    ex.flush()
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"exchange alone: {tot / n * 1e6:.1f} us per batch; submit() on the caller's thread {host / n * 1e6:.1f} us")
    # finer: raw pieces
    send, recv = ex._buffers(0)
    def rep(name, fn, n=200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"  {name}: host {((t1 - t0) / n) * 1e6:.1f} us, with drain {((time.perf_counter() - t0) / n) * 1e6:.1f} us")
    rep("all_gather_into_tensor async", lambda: dist.all_gather_into_tensor(recv, send, async_op=True))
    rep("all_gather + wait", lambda: dist.all_gather_into_tensor(recv, send, async_op=True).wait())
    rep("torch.empty x3", lambda: (torch.empty(rows, dtype=torch.int32, device=dev), torch.empty(rows, dtype=torch.int32, device=dev),
                                   torch.empty(rows * per, dtype=torch.int32, device=dev)))
    rep("pinned empty", lambda: torch.empty(4, dtype=torch.int64, pin_memory=True))
    rep("torch.zeros(4) dev", lambda: torch.zeros(4, dtype=torch.int64, device=dev))
    ev = torch.cuda.Event()
    rep("event record+sync", lambda: (ev.record(), ev.synchronize()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
