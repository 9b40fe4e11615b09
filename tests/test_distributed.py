"""N > 1 path on CPU: gloo, world_size 2 -- the row-shard + all-gather exchange of openvino_tokenizers_amd.distributed."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openvino_tokenizers_amd.distributed import all_gather_ragged, shard_rows


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lens_all, ids_all, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_rows(len(lens_all), rank, world)
        ends_all = np.cumsum(lens_all)
        begins_all = ends_all - lens_all
        t0 = int(begins_all[lo]) if lo < len(lens_all) else int(ends_all[-1])
        t1 = int(ends_all[hi - 1]) if hi > lo else t0
        # what a rank's local encode returns: offsets relative to its own ids
        b = torch.as_tensor((begins_all[lo:hi] - t0).astype(np.int32))
        e = torch.as_tensor((ends_all[lo:hi] - t0).astype(np.int32))
        ids = torch.as_tensor(ids_all[t0:t1].astype(np.int32))
        gb, ge, gi = all_gather_ragged(b, e, ids)
        q.put((rank, gb.numpy(), ge.numpy(), gi.numpy()))
    finally:
        dist.destroy_process_group()


def test_shard_rows_partition():
    for n in (0, 1, 7, 8, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_ragged_gloo_world2():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 9, size=37).astype(np.int64)  # uneven shards, some empty rows
    ids = rng.integers(0, 50000, size=int(lens.sum())).astype(np.int64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lens, ids, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ends = np.cumsum(lens).astype(np.int32)
    for _, gb, ge, gi in results:
        assert np.array_equal(ge, ends) and np.array_equal(gb, ends - lens.astype(np.int32))
        assert np.array_equal(gi, ids.astype(np.int32))
