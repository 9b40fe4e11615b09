"""N > 1 path: the row-shard exchange of openvino_tokenizers_amd.distributed.
CPU: gloo, world_size 2 and 3, pack/unpack kernels through the emulator build.  GPU: the same exchange over RCCL
("nccl") in a one-rank group on the box's single MI355X -- streams, async work handles and the HIP kernels for real."""
import os
import socket
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openvino_tokenizers_amd.distributed import ShardExchange, all_gather_ragged, shard_rows, shard_rows_by_bytes

EMU = Path(__file__).parent / "emu" / "build" / "libovtk_emu.so"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_shard(lens_all, ids_all, rank, world, device="cpu", span=None):
    """What a rank's local encode returns: offsets relative to its own ids."""
    lo, hi = span if span is not None else shard_rows(len(lens_all), rank, world)
    ends_all = np.cumsum(lens_all)
    begins_all = ends_all - lens_all
    t0 = int(begins_all[lo]) if lo < len(lens_all) else int(ends_all[-1] if len(ends_all) else 0)
    t1 = int(ends_all[hi - 1]) if hi > lo else t0
    b = torch.as_tensor((begins_all[lo:hi] - t0).astype(np.int32), device=device)
    e = torch.as_tensor((ends_all[lo:hi] - t0).astype(np.int32), device=device)
    ids = torch.as_tensor(ids_all[t0:t1].astype(np.int32), device=device)
    return b, e, ids


def _batches(seed, n_rows, vocab, n_batches=4):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_batches):
        lens = rng.integers(0, 9 + 6 * k, size=n_rows).astype(np.int64)   # later batches outgrow the first pad
        out.append((lens, rng.integers(0, vocab, size=int(lens.sum())).astype(np.int64)))
    return out


def _spans(lens, world, by_bytes):
    """The partition every rank computes for a batch: by row count, or by bytes (the id counts stand in for the text
    lengths of the rows here)."""
    if not by_bytes:
        return [shard_rows(len(lens), r, world) for r in range(world)]
    ends = np.cumsum(lens)
    return shard_rows_by_bytes(ends - lens, ends, world)


def _worker(rank, world, port, n_rows, vocab, q, by_bytes=False, transport="allgather"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openvino_tokenizers_amd import _lib as L
        lib = L.load(EMU)
        batches = _batches(7, n_rows, vocab)
        ex = ShardExchange(n_rows, vocab, "cpu", lib=lib, max_shard_rows=n_rows if by_bytes else 0, transport=transport)
        got = []
        for lens, ids in batches:
            done = ex.submit(*_local_shard(lens, ids, rank, world, span=_spans(lens, world, by_bytes)[rank]))
            if done is not None:
                got.append([t.numpy().copy() for t in done])
        got += [[t.numpy().copy() for t in b] for b in ex.flush()]
        regathers = ex.regathers
        ex.close()
        one = all_gather_ragged(*_local_shard(*batches[0], rank, world, span=_spans(batches[0][0], world, by_bytes)[rank]), n_rows=n_rows,
                                vocab_size=vocab, lib=lib, max_shard_rows=n_rows if by_bytes else 0)
        q.put((rank, got, [t.numpy().copy() for t in one], regathers, ex.id_bytes))
    except BaseException as exc:  # the parent fails on this instead of waiting for its timeout
        q.put((rank, repr(exc), None, 0, 0))
        raise
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


def test_shard_rows_by_bytes():
    """contiguous row ranges balanced by bytes.  A partition; no shard above its fair share by more than its
    largest row; skewed batches (a few huge rows) stay balanced where the count-balanced split is not."""
    rng = np.random.default_rng(3)
    for n, w in [(0, 2), (1, 3), (5, 8), (1000, 2), (1000, 8), (65536, 8)]:
        lens = rng.integers(0, 600, size=n).astype(np.int64)
        if n >= 1000:
            lens[rng.integers(0, n, size=5)] = 200000   # a few very long rows
        ends = np.cumsum(lens)
        spans = shard_rows_by_bytes(ends - lens, ends, w)
        assert len(spans) == w and spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1)) and all(a <= b for a, b in spans)
        if n >= 1000:
            share = [int(lens[a:b].sum()) for a, b in spans]
            fair = lens.sum() / w
            assert max(share) <= fair + lens.max()
    lens = np.array([1000] * 100 + [1] * 900, np.int64)   # long rows first: halves by count are 100 400 vs 500 bytes
    ends = np.cumsum(lens)
    share = [int(lens[a:b].sum()) for a, b in shard_rows_by_bytes(ends - lens, ends, 2)]
    assert abs(share[0] - share[1]) <= 1000 and sum(share) == int(lens.sum())
    # several strings per row: the row's bytes are the sum of its strings'
    b = np.array([0, 10, 10, 500, 520], np.int32)
    e = np.array([10, 10, 500, 520, 1000], np.int32)
    rb, re_ = np.array([0, 2, 3], np.int32), np.array([2, 3, 5], np.int32)
    assert shard_rows_by_bytes(b, e, 2, rb, re_) == [(0, 2), (2, 3)]


@pytest.mark.parametrize("world, n_rows, vocab, by_bytes, transport", [
    (2, 37, 50000, False, "allgather"), (3, 10, 130000, False, "allgather"), (2, 1, 100, False, "allgather"),
    (2, 41, 50000, True, "allgather"), (3, 64, 130000, True, "allgather"), (2, 37, 50000, False, "p2p"), (3, 64, 130000, True, "p2p")])
def test_shard_exchange_gloo(emu_lib, world, n_rows, vocab, by_bytes, transport):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, vocab, q, by_bytes, transport)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(r[1], list) for r in results), [r[1] for r in results]
    assert all(p.exitcode == 0 for p in procs)
    batches = _batches(7, n_rows, vocab)
    for _, got, one, regathers, id_bytes in results:
        assert len(got) == len(batches) and id_bytes == (2 if vocab <= 65536 else 4)
        for (lens, ids), (gb, ge, gi) in zip(batches, got):
            ends = np.cumsum(lens).astype(np.int32)
            assert np.array_equal(ge, ends) and np.array_equal(gb, ends - lens.astype(np.int32))
            assert np.array_equal(gi, ids.astype(np.int32))
        assert np.array_equal(one[2], batches[0][1].astype(np.int32))
        assert regathers >= 1 or n_rows == 1   # the growing batches must have forced a larger pad at least once


@pytest.mark.parametrize("by_bytes", [False, True])
@pytest.mark.parametrize("world, n_rows, vocab", [(2, 1000, 50000), (3, 4099, 130000), (8, 70000, 50000)])
def test_shard_unpack_of_several_ranks(backend, world, n_rows, vocab, by_bytes):
    """The kernels of the world > 1 path on ONE device (GPU box: one MI355X; here: the emulator): every rank's shard is
    packed into its wire, the wires are laid side by side as an all-gather would leave them, one unpack rebuilds the
    global ragged tensor.  Also a pad that is too small for the largest shard: OVTK_E_CAPACITY and max_shard."""
    import ctypes as C
    from openvino_tokenizers_amd import _lib as L
    lib = backend.lib
    if backend.name == "emu":
        n_rows = min(n_rows, 300)
    if backend.name == "hip-host":
        pytest.skip("the exchange hands over device (or emulator-host) buffers only")
    rng = np.random.default_rng(world * 100 + n_rows)
    lens = rng.integers(0, 40, size=n_rows).astype(np.int64)
    if by_bytes:
        lens[: n_rows // 7] *= 9   # skewed: the byte-balanced shards have very different row counts
    ids = rng.integers(0, vocab, size=int(lens.sum())).astype(np.int64)
    id_bytes = 2 if vocab <= 65536 else 4
    dev = backend.name != "emu"
    mem = L.MEM_DEVICE if dev else L.MEM_HOST
    spans = _spans(lens, world, by_bytes)
    h = C.c_void_p()
    L.check(lib, lib.ovtk_shard_exchange_create(world, C.c_int64(n_rows), id_bytes, C.c_int64(max(b - a for a, b in spans) if by_bytes else 0),
                                                0, C.byref(h)))
    try:
        shards = [_local_shard(lens, ids, r, world, device="cuda" if dev else "cpu", span=spans[r]) for r in range(world)]
        biggest = max(int(s[2].numel()) for s in shards)

        def alloc(n, dtype):
            return torch.empty(max(n, 1), dtype=dtype, device="cuda" if dev else "cpu")

        def ptr(t):
            return C.c_void_p(t.data_ptr())

        def exchange(pad):
            wire = int(lib.ovtk_shard_wire_bytes(h, C.c_int64(pad)))
            recv = alloc(wire * world, torch.uint8)
            for r, (b, e, d) in enumerate(shards):
                send = recv[r * wire:(r + 1) * wire]
                L.check(lib, lib.ovtk_shard_pack(h, ptr(b), ptr(e), ptr(d), C.c_int64(b.numel()), C.c_int64(d.numel()), C.c_int64(pad),
                                                 ptr(send), mem, None))
            cap = pad * world
            ob, oe, oi = alloc(n_rows, torch.int32), alloc(n_rows, torch.int32), alloc(cap, torch.int32)
            res = alloc(4, torch.int64)
            rc = lib.ovtk_shard_unpack(h, ptr(recv), C.c_int64(pad), ptr(ob), ptr(oe), ptr(oi), C.c_int64(cap), ptr(res), mem, None)
            if dev:
                torch.cuda.synchronize()
            return rc, res.cpu().numpy(), ob.cpu().numpy(), oe.cpu().numpy(), oi.cpu().numpy()

        rc, res, ob, oe, oi = exchange((biggest + 7) // 8 * 8)
        assert rc == L.OVTK_OK and res[2] == L.OVTK_OK and res[0] == len(ids) and res[1] == biggest
        ends = np.cumsum(lens).astype(np.int32)
        assert np.array_equal(oe[:n_rows], ends) and np.array_equal(ob[:n_rows], ends - lens.astype(np.int32))
        assert np.array_equal(oi[:len(ids)], ids.astype(np.int32))
        if biggest > 8:
            rc, res, *_ = exchange((biggest // 2) // 8 * 8)
            assert res[2] == L.E_CAPACITY and res[1] == biggest   # every rank learns the pad that is needed
    finally:
        lib.ovtk_shard_exchange_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("own_stream", [False, True])
def test_shard_exchange_rccl_one_rank(hip_lib, own_stream):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n_rows, vocab = 5000, 50257
        batches = _batches(11, n_rows, vocab, n_batches=5)
        ex = ShardExchange(n_rows, vocab, dev, lib=hip_lib, stream=torch.cuda.Stream(dev) if own_stream else None)
        got = []
        for lens, ids in batches:
            local = _local_shard(lens, ids, 0, 1, device=dev)
            torch.cuda.synchronize()   # a stream of its own: the shard must be complete when it is handed over
            done = ex.submit(*local)
            if done is not None:
                got.append([t.cpu().numpy() for t in done])
        got += [[t.cpu().numpy() for t in b] for b in ex.flush()]
        assert len(got) == len(batches) and ex.regathers >= 1
        ex.close()
        for (lens, ids), (gb, ge, gi) in zip(batches, got):
            ends = np.cumsum(lens).astype(np.int32)
            assert np.array_equal(ge, ends) and np.array_equal(gb, ends - lens.astype(np.int32))
            assert np.array_equal(gi, ids.astype(np.int32))
    finally:
        dist.destroy_process_group()
