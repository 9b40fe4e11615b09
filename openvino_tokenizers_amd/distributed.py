"""Row-sharded encode across the GPUs of one node.

Rows of a batch are independent in every op of the path (the only cross-row state in the reference is the
running output offset, src/bpe_tokenizer.cpp:141-161), so each rank encodes a contiguous row shard with its own
replica of the read-only tables and there is exactly ONE exchange step: an all-gather of the per-shard ragged
token-id tensors (RCCL over xGMI; `torch.distributed` backend "nccl" on ROCm).  The reference has no counterpart.

RCCL has no all-gather-v, so shards are padded to the largest one: first the (rows, tokens) counts are gathered
(16 bytes per rank), then row lengths and ids with `all_gather_into_tensor`; every rank rebuilds the global
begins/ends with one cumulative sum.  At config-4 sizes a shard is ~60 MB of ids: one bucket, no chunking.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row range [lo, hi) of `rank` (balanced by count; rows of the synthetic batches have equal
    expected length)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_ragged(begins: torch.Tensor, ends: torch.Tensor, ids: torch.Tensor, group=None):
    """Local ragged ids (begins/ends i32[rows_local], ids i32[n_local]) of every rank -> the global ragged tensor
    (begins, ends, ids) in rank order, identical on all ranks."""
    world = dist.get_world_size(group)
    dev = ids.device
    counts = torch.tensor([begins.numel(), ids.numel()], dtype=torch.int64, device=dev)
    all_counts = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    all_counts = all_counts.view(world, 2).cpu()
    max_rows, max_ids = int(all_counts[:, 0].max()), int(all_counts[:, 1].max())

    lens_pad = torch.zeros(max_rows, dtype=torch.int32, device=dev)
    lens_pad[: begins.numel()] = ends - begins
    ids_pad = torch.zeros(max(max_ids, 1), dtype=torch.int32, device=dev)
    ids_pad[: ids.numel()] = ids
    all_lens = torch.empty(world * max_rows, dtype=torch.int32, device=dev)
    all_ids = torch.empty(world * ids_pad.numel(), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_lens, lens_pad, group=group)
    dist.all_gather_into_tensor(all_ids, ids_pad, group=group)

    lens = torch.cat([all_lens[r * max_rows: r * max_rows + int(all_counts[r, 0])] for r in range(world)])
    out_ids = torch.cat([all_ids[r * ids_pad.numel(): r * ids_pad.numel() + int(all_counts[r, 1])] for r in range(world)])
    g_ends = torch.cumsum(lens, 0, dtype=torch.int64).to(torch.int32)
    return g_ends - lens, g_ends, out_ids
