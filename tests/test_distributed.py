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


@pytest.mark.parametrize("vocab_name, pad_frac", [("gpt2_small", 1.2), ("gpt2_small", 0.5), ("llama3_small", 1.1)])
def test_encode_straight_to_wire(backend, vocab_name, pad_frac):
    """ovtk_encode_enqueue_wire: the fused encode writes the exchange's wire itself (ids narrowed in compact_kernel) --
    byte for byte what ovtk_shard_pack builds from the ragged ids of ovtk_encode_run, a too small pad included (the wire
    is cut, the header still says how many ids the shard has)."""
    import ctypes as C
    from openvino_tokenizers_amd import _lib as L
    from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
    from tests.util import BpeTok
    from tools.workloads import TextModel, ragged_rows
    if backend.name == "hip-host":
        pytest.skip("the exchange hands over device (or emulator-host) buffers only")
    lib = backend.lib
    dev = backend.name != "emu"
    tok = BpeTok.load(vocab_name)
    n = 300 if dev else 70   # more than a small batch: the three-kernel pipeline with compact_kernel<WireSink>
    b, e, c = TextModel(17, "mixed" if "llama" in vocab_name else "zipf").batch(n, 300 if dev else 1100)
    rb, re_ = ragged_rows(n)
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**tok.attrs, lib=lib))
    data = backend.data([rb, re_, b, e, c])
    ob, oe, ids = fused.evaluate(data + [tok.pattern_u8()], tok.consts)
    n_ids = int(len(ids))
    id_bytes = 2 if len(tok.vocab) <= 65536 else 4
    pad = max(8, int(n_ids * pad_frac) // 8 * 8)
    h = C.c_void_p()
    L.check(lib, lib.ovtk_shard_exchange_create(1, C.c_int64(n), id_bytes, C.c_int64(0), 0, C.byref(h)))
    try:
        max_rows = int(lib.ovtk_shard_max_rows(h))
        nbytes = int(lib.ovtk_shard_wire_bytes(h, C.c_int64(pad)))

        def alloc():
            return torch.zeros(nbytes, dtype=torch.uint8, device="cuda" if dev else "cpu")

        def ptr(t):
            return C.c_void_p(t.data_ptr())

        t_ob, t_oe, t_ids = (torch.as_tensor(backend.host(x)).to("cuda" if dev else "cpu") for x in (ob, oe, ids))
        want, got = alloc(), alloc()
        mem = L.MEM_DEVICE if dev else L.MEM_HOST
        L.check(lib, lib.ovtk_shard_pack(h, ptr(t_ob), ptr(t_oe), ptr(t_ids), C.c_int64(n), C.c_int64(n_ids), C.c_int64(pad), ptr(want), mem, None))
        d = [torch.as_tensor(backend.host(x)).to("cuda" if dev else "cpu") for x in data]
        rs = L.RaggedStrings(ptr(d[0]), ptr(d[1]), n, L.Strings(ptr(d[2]), ptr(d[3]), ptr(d[4]), n, len(c)))
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_enqueue_wire(fused.split._h, fused.bpe._h, C.byref(rs), None, ptr(got), C.c_int64(max_rows),
                                                  C.c_int64(pad), id_bytes, None, C.byref(pending)))
        out = L.RaggedI32Out(None, None, None, 0, 0, 0)
        L.check(lib, lib.ovtk_encode_finish(pending, C.byref(out)))
        if dev:
            torch.cuda.synchronize()
        assert out.n_data == n_ids
        used = 16 + 4 * max_rows + id_bytes * min(n_ids, pad)
        assert torch.equal(want[:used].cpu(), got[:used].cpu())
    finally:
        lib.ovtk_shard_exchange_destroy(h)


@pytest.mark.parametrize("n_rows", [0, 5])
def test_encode_to_wire_of_a_shard_without_text(backend, n_rows):
    """A shard of empty strings (or of no rows at all: shard_rows_by_bytes next to one very long row) still yields a wire --
    header {0 ids, n_rows}, row ends of zero -- byte for byte what ovtk_shard_pack makes of n_rows empty rows; the other
    ranks are already in the collective, an error here would leave them there."""
    import ctypes as C
    from openvino_tokenizers_amd import _lib as L
    from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
    from tests.util import BpeTok
    from tools.workloads import ragged_rows
    if backend.name == "hip-host":
        pytest.skip("the exchange hands over device (or emulator-host) buffers only")
    lib = backend.lib
    dev = backend.name != "emu"
    where = "cuda" if dev else "cpu"
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**tok.attrs, lib=lib))
    rb, re_ = ragged_rows(n_rows)
    zeros = np.zeros(n_rows, np.int32)
    data = [torch.as_tensor(np.ascontiguousarray(x)).to(where) for x in (rb, re_, zeros, zeros, np.zeros(1, np.uint8))]
    h = C.c_void_p()
    L.check(lib, lib.ovtk_shard_exchange_create(2, C.c_int64(16), 2, C.c_int64(0), 0, C.byref(h)))
    try:
        max_rows, pad = int(lib.ovtk_shard_max_rows(h)), 64
        nbytes = int(lib.ovtk_shard_wire_bytes(h, C.c_int64(pad)))
        want = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device=where)
        got = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device=where)

        def ptr(t):
            return C.c_void_p(t.data_ptr())
        mem = L.MEM_DEVICE if dev else L.MEM_HOST
        L.check(lib, lib.ovtk_shard_pack(h, ptr(data[2]), ptr(data[3]), ptr(data[2]), C.c_int64(n_rows), C.c_int64(0), C.c_int64(pad), ptr(want), mem, None))
        n_ids = fused.enqueue_wire(data[:4] + [data[4][:0]] + [tok.pattern_u8()], tok.consts, got, max_rows, pad, 2)()
        if dev:
            torch.cuda.synchronize()
        used = 16 + 4 * max_rows
        assert n_ids == 0
        assert torch.equal(want[:used].cpu(), got[:used].cpu())
        assert got[:8].cpu().numpy().view(np.int32).tolist() == [0, n_rows]
    finally:
        lib.ovtk_shard_exchange_destroy(h)


def _wire_mode_run(lib, device, rank, world, n_rows, nbytes, n_batches=3):
    """Every rank encodes its row shard of each batch straight into a leased wire and submits it; returns (exchanged
    global tensors, single-process encodes of the whole batches, regathers).  Later batches are longer, so a wire leased
    with the first pad is outgrown and the shard is encoded again (refill)."""
    from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
    from tests.util import BpeTok
    from tools.workloads import TextModel, ragged_rows
    tok = BpeTok.load("gpt2_small")
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**tok.attrs, lib=lib))
    pat = tok.pattern_u8()

    def on_dev(x):
        return torch.as_tensor(np.ascontiguousarray(x), device=device)
    ex = ShardExchange(n_rows, len(tok.vocab), device, lib=lib, headroom=1.0)
    got, want = [], []
    for k in range(n_batches):
        b, e, c = TextModel(40 + k, "zipf").batch(n_rows, nbytes * (1 + k))
        whole = fused.evaluate([on_dev(x) for x in (*ragged_rows(n_rows), b, e, c)] + [pat], tok.consts)
        want.append([np.asarray(t.cpu() if hasattr(t, "cpu") else t).copy() for t in whole])
        lo, hi = shard_rows(n_rows, rank, world)
        rb, re_ = ragged_rows(hi - lo)
        mine = [on_dev(x) for x in (rb, re_, b[lo:hi], e[lo:hi], c)] + [pat]

        def encode(wire, wait=True, mine=mine):   # bound now: a refill comes batches later
            t = fused.enqueue_wire(mine, tok.consts, wire.t, ex.max_rows, wire.pad, ex.id_bytes)
            return t() if wait else t
        if k == 0:   # the pad comes from a count the caller knows: one throw-away encode into a minimal wire tells it
            ex.pad_ids = 8
            n0 = encode(ex.lease_wire())
            ex.pad_ids = 0
            ex.agree_pad(n0)
        wire = ex.lease_wire()
        encode(wire)
        done = ex.submit_wire(wire, encode)
        if done is not None:
            got.append([t.cpu().numpy().copy() for t in done])
    got += [[t.cpu().numpy().copy() for t in b] for b in ex.flush()]
    regathers = ex.regathers
    ex.close()
    return got, want, regathers


def _wire_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openvino_tokenizers_amd import _lib as L
        q.put((rank,) + _wire_mode_run(L.load(EMU), "cpu", rank, world, n_rows=22, nbytes=900))
    except BaseException as exc:
        q.put((rank, repr(exc), None, 0))
        raise
    finally:
        dist.destroy_process_group()


def test_encode_to_wire_exchange_gloo(emu_lib):
    """World 2: the fused encode of each rank writes its send wire itself (compact_kernel<WireSink>); what the exchange
    hands back equals the one-process encode of the whole batch, through a pad that had to grow."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wire_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(r[1], list) for r in results), [r[1] for r in results]
    for _, got, want, regathers in results:
        assert len(got) == len(want) == 3 and regathers >= 1
        for g, w in zip(got, want):
            for a, b in zip(g, w):
                assert np.array_equal(a, b)


@pytest.mark.gpu
def test_encode_to_wire_exchange_rccl_one_rank(hip_lib):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        got, want, regathers = _wire_mode_run(hip_lib, dev, 0, 1, n_rows=3000, nbytes=200, n_batches=4)
        assert len(got) == len(want) == 4 and regathers >= 1
        for g, w in zip(got, want):
            for a, b in zip(g, w):
                assert np.array_equal(a, b)
    finally:
        dist.destroy_process_group()


def _wire_worker_rccl(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from openvino_tokenizers_amd import _lib as L
        q.put((rank,) + _wire_mode_run(L.load(), dev, rank, world, n_rows=3001, nbytes=200, n_batches=4))
    except BaseException as exc:
        q.put((rank, repr(exc), None, 0))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_encode_to_wire_exchange_rccl_two_ranks(hip_lib):
    """The N > 1 path on real hardware, as far as this pool goes: TWO processes, one GPU each, RCCL over xGMI -- every rank's
    fused encode writes its send wire (ovtk_encode_enqueue_wire), ShardExchange gathers, each rank's result equals the
    one-process encode of the whole batch, through a pad that had to grow.  Skipped on a one-GPU box (the driver's 8-GPU node
    runs it without a new round)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wire_worker_rccl, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(isinstance(r[1], list) for r in results), [r[1] for r in results]
    for _, got, want, regathers in results:
        assert len(got) == len(want) == 4 and regathers >= 1
        for g, w in zip(got, want):
            for a, b in zip(g, w):
                assert np.array_equal(a, b)
