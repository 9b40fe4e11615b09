#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): input MB/s encoded, GPT-2-shaped byte-level BPE, 65 536 x ~512-byte strings
per GPU (config 2), inputs resident in HBM, fused RegexSplit+BPETokenizer through the C ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W]           (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch.  With N > 1 every rank encodes its own 65 536-row shard
(weak scaling) and the step includes the all-gather of the ragged token ids over RCCL/xGMI
(openvino_tokenizers_amd/distributed.py); `value` = bytes of all ranks / max-over-ranks time.
Prints ONE JSON line on rank 0.  The oracle is used only for the cpu_baseline leg and a parity spot-check.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.ops import BPETokenizer, RegexSplit  # noqa: E402
from tools.harness import BpeTok  # noqa: E402
from tools.workloads import TextModel, ragged_rows  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
KERNEL_NAMES = {"lookup_fused": "lookup_kernel<kFused>", "lookup_pieces": "lookup_kernel<kPieces>",
                "bpe_merge": "merge_kernel", "bpe_exact": "exact_kernel", "compact": "compact_kernel",
                "scan_rows": "tile_{reduce,scan,apply}_kernel"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--bytes", type=int, default=512)
    ap.add_argument("--text", default="zipf", choices=["zipf", "uniform", "mixed"])
    ap.add_argument("--tokenizer", default="gpt2")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the all-gather (rank-local consumer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        from openvino_tokenizers_amd.distributed import all_gather_ragged

    lib = L.load()
    tok = BpeTok.load(args.tokenizer)
    begins, ends, chars = TextModel(1234, args.text).batch(args.rows, args.bytes, seed=1000 + rank)
    rb, re_ = ragged_rows(args.rows)
    n_chars = int(len(chars))
    d = [torch.as_tensor(x, device=dev) for x in (rb, re_, begins, ends, chars)]

    split = RegexSplit("isolate", device=local_rank, lib=lib)
    bpe = BPETokenizer(**tok.attrs, device=local_rank, lib=lib)
    split._ensure(tok.pattern_u8())
    bpe._ensure(d + tok.consts)

    # pre-built C-ABI arguments: nothing but the library call is inside a step
    rs = L.RaggedStrings(d[0].data_ptr(), d[1].data_ptr(), args.rows,
                         L.Strings(d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), args.rows, n_chars))
    o_begins = torch.empty(args.rows, dtype=torch.int32, device=dev)
    o_ends = torch.empty(args.rows, dtype=torch.int32, device=dev)
    o_ids = torch.empty(n_chars, dtype=torch.int32, device=dev)
    out = L.RaggedI32Out(o_begins.data_ptr(), o_ends.data_ptr(), o_ids.data_ptr(), n_chars, 0, 0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        L.check(lib, lib.ovtk_encode_run(split._h, bpe._h, C.byref(rs), None, C.byref(out), L.MEM_DEVICE, stream))
        if world > 1 and not args.no_gather:
            return all_gather_ragged(o_begins, o_ends, o_ids[: out.n_data])
        return None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib.ovtk_profile_reset()
    lib.ovtk_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    lib.ovtk_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nb = torch.tensor([n_chars], dtype=torch.int64, device=dev)
        dist.all_reduce(nb)
        total_bytes = int(nb.item())
    else:
        total_bytes = n_chars

    n_tokens = int(out.n_data)
    ms_per_step = dt / args.steps * 1e3
    value = total_bytes * args.steps / dt / 1e6  # MB/s, whole job

    # ---- roofline of the dominant kernel (HIP events recorded by the library on the launch stream)
    buf = C.create_string_buffer(8192)
    lib.ovtk_profile_dump(buf, 8192)
    prof = {ln.split()[0]: (float(ln.split()[1]), int(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.strip()}
    kernels = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}
    per_step = {k: v[0] / args.steps for k, v in prof.items()}  # ms of each kernel family per step
    roofline = None
    if per_step:
        dom = max(per_step, key=per_step.get)  # the dominant kernel of the step
        k_ms = prof[dom][0] / max(prof[dom][1], 1)
        algo_bytes = n_chars + 4 * n_tokens + 16 * args.rows  # A_enc = N_c + 4 N_t + 16 B per launch
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": KERNEL_NAMES.get(dom, dom), "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": None, "kernel_ms": round(k_ms, 4), "algorithmic_bytes": algo_bytes,
                    "input_GBps_kernel_only": round(n_chars / (k_ms * 1e-3) / 1e9, 2),
                    "all_kernels_ms_per_step": round(sum(per_step.values()), 4)}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- parity spot-check + CPU baseline (oracle = "port" of the reference's algorithm), rank 0 only
    cpu_baseline, parity = None, None
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        orc, ors = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
        n_chk = min(args.rows, 1024)
        ref = orc(*ors(rb[:n_chk], re_[:n_chk], begins[:n_chk], ends[:n_chk], chars)[:5])  # also warms the piece cache
        got_ends = o_ends[:n_chk].cpu().numpy()
        got_ids = o_ids[: int(got_ends[-1])].cpu().numpy()
        parity = bool(np.array_equal(ref[1], got_ends) and np.array_equal(ref[2], got_ids))
        if world == 1:
            n_s = min(args.rows, 32768)
            t1 = time.perf_counter()
            sp = ors(rb[:n_s], re_[:n_s], begins[:n_s], ends[:n_s], chars)
            orc(*sp[:5])
            cdt = time.perf_counter() - t1
            sample_bytes = int(ends[n_s - 1] - begins[0])
            cpu_baseline = {"value": round(sample_bytes / cdt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                            "sample": f"first {n_s} rows ({sample_bytes} bytes) of the same batch, RegexSplit(PCRE2 "
                                      f"JIT)+BPETokenizer restatement with warm piece cache, {cdt:.2f} s",
                            "host_cpus": os.cpu_count()}

    line = {
        "metric": "input MB/s encoded (GPT-2 BPE, 512-byte strings)", "value": round(value, 1), "unit": "MB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
        "config": {"workload": f"config 2: GPT-2-shaped byte-level BPE (V=50257, 50000 merges, trained in-process), "
                               f"{args.rows} x ~{args.bytes}-byte {args.text} strings per GPU, fused RegexSplit+BPETokenizer, "
                               f"inputs and outputs in HBM",
                   "rows_per_gpu": args.rows, "bytes_per_gpu": n_chars, "tokens_per_gpu": n_tokens,
                   "exchange": ("none (1 GPU)" if world == 1 else ("none (--no-gather)" if args.no_gather else
                                                                    "all-gather of ragged ids over RCCL"))},
        "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_prefix_bit_exact": parity,
        "kernel_ms": kernels,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
