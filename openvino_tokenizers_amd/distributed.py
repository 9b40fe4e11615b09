"""Row-sharded encode across the GPUs of one node.

Rows of a batch are independent in every op of the path (the only cross-row state in the reference is the
running output offset, src/bpe_tokenizer.cpp:141-161), so each rank encodes a contiguous row shard with its own
replica of the read-only tables and there is exactly ONE exchange step: an all-gather of the per-shard ragged
token-id tensors (RCCL over xGMI; `torch.distributed` backend "nccl" on ROCm).  The reference has no counterpart.

RCCL has no all-gather-v, so every rank sends one fixed-size *wire* (include/ovtk_amd.h, "row-shard exchange"):
its row lengths, then its ids narrowed to 2 bytes when the vocabulary allows and padded to a size all ranks agreed
on.  One collective per batch, no counts exchange and no host round trip: the receiver's scan over the global row
lengths locates every shard.  `ShardExchange` keeps two wires in flight so the gather of batch k (RCCL's own
stream, xGMI) overlaps the encode of batch k+1: xGMI is point-to-point, 7 links x ~50 GB/s each way per GPU, so an
8-rank gather of 4-byte ids would take longer than the encode itself.  (Measured on one MI355X: driving the exchange
from a second host thread does not help; a HIP stream of its own for pack / unpack does -- see `stream` below --
once the process has a hardware queue per stream.)
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row range [lo, hi) of `rank` (balanced by count; rows of the synthetic batches have equal
    expected length)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows_by_bytes(begins, ends, world: int, ragged_begins=None, ragged_ends=None):
    """Contiguous row ranges balanced by BYTES (prefix sum over ends - begins, cut at row boundaries): the
    cost of a row is its text, and real batches are not of equal-length rows.  begins / ends: the strings' offsets
    (numpy or torch, host); ragged_begins / ragged_ends: the rows' string ranges (default: one string per row).
    Returns [(lo, hi)] * world, identical on every rank; every row lands in the shard that holds the midpoint of its
    bytes in the cumulative text, so no shard exceeds its fair share by more than one row."""
    import numpy as np
    b = np.asarray(begins, dtype=np.int64)
    e = np.asarray(ends, dtype=np.int64)
    str_len = e - b
    if ragged_begins is None:
        row_len = str_len
    else:
        csum = np.concatenate([[0], np.cumsum(str_len)])
        row_len = csum[np.asarray(ragged_ends, dtype=np.int64)] - csum[np.asarray(ragged_begins, dtype=np.int64)]
    n = len(row_len)
    cum = np.cumsum(row_len)
    total = int(cum[-1]) if n else 0
    if total == 0:
        return [shard_rows(n, r, world) for r in range(world)]
    mid = cum - row_len / 2.0
    cuts = [int(np.searchsorted(mid, total * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n
    return [(cuts[r], max(cuts[r], cuts[r + 1])) for r in range(world)]


def _round_up(x, m):
    return (int(x) + m - 1) // m * m


class ShardExchange:
    """All-gather of ragged token ids over the ranks of `group`, pipelined with the encode.

        ex = ShardExchange(n_rows_global, vocab_size, device)
        for batch in batches:
            done = ex.submit(begins, ends, ids)      # local shard -> an EARLIER batch's global (begins, ends, ids), or None
        rest = ex.flush()                            # list of the batches still in flight, oldest first

    submit(k) enqueues, on the caller's stream and without waiting for the GPU: the pack of batch k (HIP kernel); its
    all-gather (asynchronous: RCCL's own stream, so the wires travel over xGMI while the caller encodes batch k+1);
    and the unpack of batch k-1, whose gather had the whole encode of batch k to finish.  It hands back batch k-2,
    reading a 32-byte verdict the unpack left in pinned memory.  Up to three batches are with the exchange; the local
    tensors of a batch must stay untouched until that batch is handed back (a shard that outgrew the agreed pad is
    packed again from them), so a caller cycling output buffers needs four sets.
    stream: a torch.cuda.Stream for the pack / unpack kernels instead of the caller's current stream, so that they do
    not queue behind (or in front of) encodes the caller already launched on it; the shard handed to submit() must then
    be complete (the two-half encode's finish() is host-synchronised), and a handed-back tensor used on another stream
    wants record_stream().  Give the process enough hardware queues (GPU_MAX_HW_QUEUES=8): with the default 4 the
    encode streams, this one and RCCL's share queues and run one after another (0.31 instead of 0.25 ms per batch).
    Tensors may be CUDA (RCCL) or CPU (gloo; the kernels then go through the library's host-memory path -- the CPU
    tests run that with the emulator build).
    """

    def __init__(self, n_rows: int, vocab_size: int, device, group=None, lib=None, pad_ids: int = 0, headroom: float = 1.125,
                 stream=None, max_shard_rows: int = 0, transport: str = "allgather"):
        self.lib = lib if lib is not None else L.load()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_rows = int(n_rows)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = stream      # torch.cuda.Stream for pack / unpack (None: the stream current at each call)
        # How the wires travel.  "allgather": one all_gather_into_tensor (RCCL picks the algorithm -- on a ring every byte
        # crosses N-1 links one after the other).  "p2p": every rank sends its wire straight to each peer and receives
        # theirs, as ONE group of isend / irecv pairs (RCCL: grouped ncclSend / ncclRecv) -- on the MI355X node's fully
        # connected xGMI each pair has a link of its own, so a wire crosses exactly one link and the 7 transfers of a
        # rank run side by side ("direct 7-link exchange").  Same wires, same unpack, same result.
        if transport not in ("allgather", "p2p"):
            raise ValueError("transport must be 'allgather' or 'p2p'")
        self.transport = transport
        self.id_bytes = 2 if int(vocab_size) <= 65536 else 4
        self.headroom = float(headroom)
        self.pad_ids = _round_up(pad_ids, 8)
        self._h = C.c_void_p()
        # max_shard_rows: row slots per wire; 0 = shards balanced by row count, else the largest shard of the partition in
        # use (shard_rows_by_bytes: pass n_rows when the partition changes from batch to batch)
        L.check(self.lib, self.lib.ovtk_shard_exchange_create(self.world, C.c_int64(self.n_rows), self.id_bytes, C.c_int64(max_shard_rows),
                                                              (self.device.index or 0) if self.cuda else 0, C.byref(self._h)))
        self.max_rows = int(self.lib.ovtk_shard_max_rows(self._h))
        self._wires = {}          # slot -> (send, recv); two slots: a wire is free again once its batch is unpacked
        self._free_wires = []     # send wires handed out by lease_wire() and back from their batches (encode-to-wire mode)
        self._verdicts = [self._verdict_slot() for _ in range(2)]
        self._k = 0
        self._gathering = None    # batch whose all-gather is in flight
        self._unpacking = None    # batch whose unpack is enqueued
        self.regathers = 0

    def _verdict_slot(self):
        if not self.cuda:
            return torch.zeros(4, dtype=torch.int64), torch.zeros(4, dtype=torch.int64), None
        return (torch.zeros(4, dtype=torch.int64, device=self.device), torch.zeros(4, dtype=torch.int64).pin_memory(),
                torch.cuda.Event())

    def close(self):
        if self._h:
            if self.cuda:
                torch.cuda.synchronize(self.device)
            self.lib.ovtk_shard_exchange_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    # -- plumbing
    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr())

    def _stream_ptr(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.cuda else C.c_void_p(0)

    @property
    def _mem(self):
        return L.MEM_DEVICE if self.cuda else L.MEM_HOST

    def _buffers(self, slot):
        nbytes = int(self.lib.ovtk_shard_wire_bytes(self._h, C.c_int64(self.pad_ids)))
        w = self._wires.get(slot)
        if w is None or w[0].numel() != nbytes:
            w = (torch.empty(nbytes, dtype=torch.uint8, device=self.device),
                 torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device))
            self._wires[slot] = w
        return w

    def _agree_pad(self, n_local: int):
        """First batch: one MAX all-reduce fixes the pad for the batches that follow."""
        t = torch.tensor([n_local], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.pad_ids = max(self.pad_ids, _round_up(int(t.item()) * self.headroom + 8, 8))

    def agree_pad(self, n_local: int):
        """Fix the pad from an id count the caller already knows (a collective: every rank calls it).  The encode-to-wire
        mode needs the pad before its first encode, where submit() learns it from its first shard."""
        self._agree_pad(int(n_local))
        return self.pad_ids

    def lease_wire(self):
        """A send wire for FusedSplitBPE.enqueue_wire(..., wire.t, ex.max_rows, wire.pad, ex.id_bytes) to fill -- the
        encode's compact_kernel then writes header, row ends and narrowed ids itself, and submit_wire() sends the wire as
        it is: no ragged int32 ids in between, no pack kernel.  The wire comes back to the pool with its batch."""
        if self.pad_ids == 0:
            raise RuntimeError("the encode-to-wire mode needs a pad first: ShardExchange(pad_ids=...) or agree_pad()")
        while self._free_wires:
            w = self._free_wires.pop()
            if w.pad == self.pad_ids:
                return w
        nbytes = int(self.lib.ovtk_shard_wire_bytes(self._h, C.c_int64(self.pad_ids)))
        return _Wire(torch.empty(nbytes, dtype=torch.uint8, device=self.device), self.pad_ids)

    def submit_wire(self, wire, refill):
        """submit() for a wire the fused encode filled (the encode's ticket() has returned: the wire is complete).
        refill(wire) -> None encodes the same shard again, synchronously, into another leased wire: called when the pad grew
        after this wire was leased, or when this shard is the one that outgrew it (every rank learns that from the
        headers, together)."""
        with self._on_stream():
            return self._submit(_FromWire(wire, refill))

    def _start_gather(self, local, slot):
        if isinstance(local, _FromWire):
            if local.wire.pad != self.pad_ids:
                local.wire = self.lease_wire()
                local.refill(local.wire)
            send, recv = local.wire.t, self._buffers(slot)[1]
        else:
            begins, ends, ids = local
            send, recv = self._buffers(slot)
            L.check(self.lib, self.lib.ovtk_shard_pack(self._h, self._ptr(begins), self._ptr(ends), self._ptr(ids), C.c_int64(begins.numel()),
                                                       C.c_int64(ids.numel()), C.c_int64(self.pad_ids), self._ptr(send), self._mem,
                                                       self._stream_ptr()))
        if self.transport == "allgather" or self.world == 1:
            work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)  # RCCL's stream waits for the pack
        else:
            n = send.numel()
            recv[self.rank * n:(self.rank + 1) * n].copy_(send, non_blocking=True)
            ops = []
            for step in range(1, self.world):   # rank r's k-th pair: send to r+k, receive from r-k -- every link busy at once
                to, frm = (self.rank + step) % self.world, (self.rank - step) % self.world
                ops.append(dist.P2POp(dist.isend, send, self._global_rank(to), group=self.group))
                ops.append(dist.P2POp(dist.irecv, recv[frm * n:(frm + 1) * n], self._global_rank(frm), group=self.group))
            work = _Works(dist.batch_isend_irecv(ops))
        return dict(local=local, slot=slot, work=work, recv=recv, pad=self.pad_ids)

    def _global_rank(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _start_unpack(self, b):
        """Gathered wires -> global ragged tensor; only enqueues (CUDA)."""
        pad, cap = b["pad"], b["pad"] * self.world
        b["work"].wait()   # CUDA: the caller's stream waits for RCCL's stream; the host does not block
        begins = torch.empty(max(self.n_rows, 1), dtype=torch.int32, device=self.device)
        ends = torch.empty(max(self.n_rows, 1), dtype=torch.int32, device=self.device)
        ids = torch.empty(max(cap, 1), dtype=torch.int32, device=self.device)
        res, host, done = self._verdicts[b["slot"]]
        L.check(self.lib, self.lib.ovtk_shard_unpack(self._h, self._ptr(b["recv"]), C.c_int64(pad), self._ptr(begins), self._ptr(ends),
                                                     self._ptr(ids), C.c_int64(cap), self._ptr(res), self._mem, self._stream_ptr()))
        if self.cuda:
            host.copy_(res, non_blocking=True)
            done.record()
        else:
            host = res
        b.update(out=(begins, ends, ids), res=host, done=done)
        return b

    def _collect(self, b):
        """Verdict of an unpack (waits for it if need be) -> the global tensor, or the exchange again with a larger pad."""
        while True:
            if b["done"] is not None:
                b["done"].synchronize()
            n_ids, biggest, status = b["res"][:3].tolist()
            if status == L.E_CAPACITY and biggest > b["pad"]:
                # Every rank sees the same lens, so every rank takes this branch together.
                self.regathers += 1
                self.pad_ids = max(self.pad_ids, _round_up(biggest * self.headroom + 8, 8))
                b = self._start_unpack(self._start_gather(b["local"], b["slot"]))
                continue
            if status != L.OVTK_OK:
                raise L.OvtkError(status, "shard exchange failed (see ovtk_shard_result in include/ovtk_amd.h)")
            if isinstance(b["local"], _FromWire):
                self._free_wires.append(b["local"].wire)
            begins, ends, ids = b["out"]
            return begins[: self.n_rows], ends[: self.n_rows], ids[:n_ids]

    # -- API
    def submit(self, begins: torch.Tensor, ends: torch.Tensor, ids: torch.Tensor):
        with self._on_stream():
            local = (begins.contiguous(), ends.contiguous(), ids.contiguous())
            if self.pad_ids == 0:
                self._agree_pad(ids.numel())
            return self._submit(local)

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.cuda and self.stream is not None else _NullCtx()

    def _submit(self, local):
        done = self._collect(self._unpacking) if self._unpacking is not None else None   # frees the slot reused below
        self._unpacking = None
        started = self._start_gather(local, self._k & 1)
        self._k += 1
        if self._gathering is not None:
            self._unpacking = self._start_unpack(self._gathering)
        self._gathering = started
        return done

    def flush(self):
        with self._on_stream():
            return self._flush()

    def _flush(self):
        out = []
        if self._unpacking is not None:
            out.append(self._collect(self._unpacking))
        if self._gathering is not None:
            out.append(self._collect(self._start_unpack(self._gathering)))
        self._unpacking = self._gathering = None
        return out


class _Wire:
    """A send wire of the encode-to-wire mode: the tensor and the pad (id slots) it was sized for."""

    def __init__(self, t, pad):
        self.t, self.pad = t, pad


class _FromWire:
    def __init__(self, wire, refill):
        self.wire, self.refill = wire, refill


class _Works:
    """Several async work handles waited for as one."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def all_gather_ragged(begins: torch.Tensor, ends: torch.Tensor, ids: torch.Tensor, n_rows: int, vocab_size: int = 1 << 31,
                      group=None, lib=None, max_shard_rows: int = 0):
    """Unpipelined form: local ragged ids of every rank -> the global ragged tensor (begins, ends, ids) in rank
    order, identical on all ranks.  n_rows: global row count; the shards are any contiguous partition of the rows in
    rank order (shard_rows() / shard_rows_by_bytes(); max_shard_rows as for ShardExchange)."""
    ex = ShardExchange(n_rows, vocab_size, ids.device, group=group, lib=lib, max_shard_rows=max_shard_rows)
    ex.submit(begins, ends, ids)
    out = ex.flush()[-1]
    ex.close()
    return out
