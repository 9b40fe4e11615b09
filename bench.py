#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): input MB/s encoded, GPT-2-shaped byte-level BPE, 65 536 x ~512-byte strings
per GPU (config 2), inputs resident in HBM, fused RegexSplit+BPETokenizer through the C ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W]           (N > 1: launched by torch.distributed.run)
    python bench.py --config 3      BERT-shaped WordPiece, 65 536 x ~256-byte strings (fused BERT split + WordPiece)
    python bench.py --config 4      one 8-GPU shard of the Llama-3-shaped config (RegexSplit -> BPETokenizer chain, mixed scripts)
    python bench.py --config 5      detokenizer (VocabDecoder + ByteFallback + FuzeRagged fused), rows x 2048 ids

One "step" = one pass of the hot path over one batch.  With N > 1 every rank encodes its own shard of the same size
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

# The encode streams, the exchange stream and RCCL's own stream must not share a hardware queue (kernels of streams that
# do run one after another): the HIP runtime folds streams onto 4 queues by default.  Read when the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from openvino_tokenizers_amd import _lib as L  # noqa: E402
from openvino_tokenizers_amd.ops import (BPETokenizer, RegexSplit, VocabDecoder, WordpieceTokenizer)  # noqa: E402
from tools.harness import BpeTok, pack_strings  # noqa: E402
from tools.make_tokenizers import load_tokenizer  # noqa: E402
from tools.workloads import TextModel, ragged_rows  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
KERNEL_NAMES = {"lookup_fused": "lookup_kernel<kFused>", "lookup_pieces": "lookup_kernel<kPieces>", "shard_pack": "shard_pack_kernel",
                "shard_unpack": "shard_unpack_kernel",
                "lookup_words": "lookup_kernel<kFused> (BERT words)", "wordpiece_deferred": "wordpiece_deferred_kernel",
                "bpe_merge": "merge_kernel", "bpe_exact": "exact_kernel", "compact": "compact_kernel",
                "prep_rows": "prep_rows_kernel", "count_scan": "count_scan_kernel", "split_count": "split_kernel<0>",
                "split_write": "split_kernel<1>",
                "detokenize": "decode_write_kernel", "decode_count": "decode_count_kernel", "decode_scan": "tile_{reduce,scan,apply}_kernel<UnitLen>"}
BERT_WS = r"\s+"
BERT_PUNCT = "|".join([r"[!-/]", r"[:-@]", r"[\[-`]", r"[{-~]", r"[\p{P}]", r"[\x{4E00}-\x{9FFF}]", r"[\x{3400}-\x{4DBF}]",
                       r"[\x{20000}-\x{2A6DF}]", r"[\x{2A700}-\x{2B73F}]", r"[\x{2B740}-\x{2B81F}]",
                       r"[\x{2B820}-\x{2CEAF}]", r"[\x{F900}-\x{FAFF}]", r"[\x{2F800}-\x{2FA1F}]"])
PMC_FILE = ROOT / "profiles" / "latest_pmc.json"  # HBM traffic of the dominant kernels from a separate rocprofv3 --pmc run


class IdOut:
    """Six sets of ragged-id output buffers used in turn: up to three batches are being encoded (--depth 2) while,
    with N > 1, the exchange still holds the two before them (gathering / unpacking -- a shard that outgrew the agreed
    pad is packed again from its local ids)."""
    SETS = 6

    def __init__(self, rows, cap, dev):
        self.sets = []
        for _ in range(self.SETS):
            b = torch.empty(rows, dtype=torch.int32, device=dev)
            e = torch.empty(rows, dtype=torch.int32, device=dev)
            ids = torch.empty(cap, dtype=torch.int32, device=dev)
            self.sets.append((b, e, ids, L.RaggedI32Out(b.data_ptr(), e.data_ptr(), ids.data_ptr(), cap, 0, 0)))
        self.k = 0
        self.last = self.sets[0]

    def take(self):
        self.last = self.sets[self.k % self.SETS]
        self.k += 1
        return self.last

    @property
    def n_data(self):
        return self.last[3].n_data


class Workload:
    """One BASELINE.json configuration: how to build the inputs, run a step, count units, and check/baseline it."""
    metric = "input MB/s encoded (GPT-2 BPE, 512-byte strings)"
    unit = "MB/s"


def make_encode_bpe(args, lib, dev, rank):
    tok = BpeTok.load(args.tokenizer)
    begins, ends, chars = TextModel(1234, args.text).batch(args.rows, args.bytes, seed=1000 + rank)
    rb, re_ = ragged_rows(args.rows)
    n_chars = int(len(chars))
    d = [torch.as_tensor(x, device=dev) for x in (rb, re_, begins, ends, chars)]
    split = RegexSplit("isolate", device=dev.index, lib=lib)
    bpe = BPETokenizer(**dict(tok.attrs, cache_capacity=0 if args.no_memo else tok.attrs.get("cache_capacity", 20000)),
                       device=dev.index, lib=lib)
    split._ensure(tok.pattern_u8())
    bpe._ensure(d + tok.consts)
    rs = L.RaggedStrings(d[0].data_ptr(), d[1].data_ptr(), args.rows,
                         L.Strings(d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), args.rows, n_chars))
    out = IdOut(args.rows, n_chars, dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        o_begins, o_ends, o_ids, o = out.take()
        L.check(lib, lib.ovtk_encode_run(split._h, bpe._h, C.byref(rs), None, C.byref(o), L.MEM_DEVICE, stream))
        return o_begins, o_ends, o_ids[: o.n_data]

    def enqueue(st=stream):
        """The same step in two halves (ovtk_encode_enqueue / ovtk_encode_finish) on HIP stream `st`: -> finish() ->
        (begins, ends, ids)."""
        o_begins, o_ends, o_ids, o = out.take()
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_enqueue(split._h, bpe._h, C.byref(rs), None, C.byref(o), st, C.byref(pending)))

        def finish():
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(o)))
            return o_begins, o_ends, o_ids[: o.n_data]
        return finish

    def cpu(n_s):
        from oracle import oracle as O
        orc, ors = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
        ref = orc(*ors(rb[:1024], re_[:1024], begins[:1024], ends[:1024], chars)[:5])  # parity prefix + warms the piece cache
        t1 = time.perf_counter()
        orc(*ors(rb[:n_s], re_[:n_s], begins[:n_s], ends[:n_s], chars)[:5])
        return ref, time.perf_counter() - t1, int(ends[n_s - 1] - begins[0]), "RegexSplit(PCRE2 JIT)+BPETokenizer restatement with warm piece cache"

    workload = (f"config 2: GPT-2-shaped byte-level BPE (V=50257, 50000 merges, trained in-process), {args.rows} x "
                f"~{args.bytes}-byte {args.text} strings per GPU, fused RegexSplit+BPETokenizer, inputs and outputs in HBM"
                + (", piece memo disabled (cache_capacity=0)" if args.no_memo else ""))
    return dict(step=step, enqueue=enqueue, cpu=cpu, n_units=n_chars, out=out, keep=(d, split, bpe), workload=workload, vocab=len(tok.vocab),
                dominant="lookup_fused",
                metric="input MB/s encoded (GPT-2 BPE, 512-byte strings)", dtype="u8/int32",
                algo=lambda n_tok: n_chars + 4 * n_tok + 16 * args.rows, sample_rows=args.rows)


def make_encode_llama3(args, lib, dev, rank):
    """Config 4 shard: Llama-3-shaped byte-level BPE (tiktoken-style split pattern, 128k merges) on mixed-script text,
    fused RegexSplit + BPETokenizer (bit-parallel Llama-3 scanner)."""
    tok = BpeTok.load("llama3")
    rows = args.rows if args.rows != 65536 else 131072  # 1 M rows / 8 GPUs
    begins, ends, chars = TextModel(1234, "mixed").batch(rows, args.bytes, seed=4000 + rank)
    rb, re_ = ragged_rows(rows)
    n_chars = int(len(chars))
    d = [torch.as_tensor(x, device=dev) for x in (rb, re_, begins, ends, chars)]
    split = RegexSplit("isolate", device=dev.index, lib=lib)
    bpe = BPETokenizer(**tok.attrs, device=dev.index, lib=lib)
    split._ensure(tok.pattern_u8())
    bpe._ensure(d + tok.consts)
    rs = L.RaggedStrings(d[0].data_ptr(), d[1].data_ptr(), rows, L.Strings(d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), rows, n_chars))
    out = IdOut(rows, n_chars, dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        o_begins, o_ends, o_ids, o = out.take()
        L.check(lib, lib.ovtk_encode_run(split._h, bpe._h, C.byref(rs), None, C.byref(o), L.MEM_DEVICE, stream))
        return o_begins, o_ends, o_ids[: o.n_data]

    def enqueue(st=stream):
        o_begins, o_ends, o_ids, o = out.take()
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_enqueue(split._h, bpe._h, C.byref(rs), None, C.byref(o), st, C.byref(pending)))

        def finish():
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(o)))
            return o_begins, o_ends, o_ids[: o.n_data]
        return finish

    def cpu(n_s):
        from oracle import oracle as O
        orc, ors = tok.oracle(), O.RegexSplit(tok.pattern, "isolate")
        ref = orc(*ors(rb[:1024], re_[:1024], begins[:1024], ends[:1024], chars)[:5])
        t1 = time.perf_counter()
        orc(*ors(rb[:n_s], re_[:n_s], begins[:n_s], ends[:n_s], chars)[:5])
        return ref, time.perf_counter() - t1, int(ends[n_s - 1] - begins[0]), "RegexSplit(PCRE2 JIT)+BPETokenizer restatement with warm piece cache"

    workload = (f"config 4 shard: Llama-3-shaped byte-level BPE (V=128256, 127999 merges, trained in-process), {rows} x "
                f"~{args.bytes}-byte mixed-script strings per GPU, fused RegexSplit (tiktoken-style pattern) + BPETokenizer, "
                f"inputs and outputs in HBM")
    return dict(step=step, enqueue=enqueue, cpu=cpu, n_units=n_chars, out=out, keep=(d, split, bpe), vocab=len(tok.vocab),
                dominant="lookup_fused",
                workload=workload, metric="input MB/s encoded (Llama-3 BPE, 512-byte mixed-script strings)", dtype="u8/int32",
                rows=rows, algo=lambda n_tok: n_chars + 4 * n_tok + 16 * rows, sample_rows=min(rows, 16384))


def make_encode_wordpiece(args, lib, dev, rank):
    tok = load_tokenizer("bert")
    nbytes = args.bytes if args.bytes != 512 else 256
    begins, ends, chars = TextModel(1234, "zipf").batch(args.rows, nbytes, seed=2000 + rank)
    chars = np.frombuffer(chars.tobytes().lower(), np.uint8).copy()  # BERT-uncased normalisation is upstream of this path
    rb, re_ = ragged_rows(args.rows)
    n_chars = int(len(chars))
    d = [torch.as_tensor(x, device=dev) for x in (rb, re_, begins, ends, chars)]
    ws = RegexSplit("remove", device=dev.index, lib=lib)
    pu = RegexSplit("isolate", device=dev.index, lib=lib)
    wp = WordpieceTokenizer(tok["suffix_indicator"], tok["max_bytes_per_word"], device=dev.index, lib=lib)
    ws._ensure(BERT_WS)
    pu._ensure(BERT_PUNCT)
    consts = list(pack_strings(tok["vocab"])) + [np.asarray(tok["unk_id"], np.int32)]
    wp._ensure(d + consts)
    rs = L.RaggedStrings(d[0].data_ptr(), d[1].data_ptr(), args.rows,
                         L.Strings(d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), args.rows, n_chars))
    out = IdOut(args.rows, n_chars, dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    unk = C.c_int32(int(tok["unk_id"]))

    def step():
        o_begins, o_ends, o_ids, o = out.take()
        L.check(lib, lib.ovtk_wordpiece_encode_run(wp._h, ws._h, pu._h, C.byref(rs), unk, C.byref(o), L.MEM_DEVICE, stream))
        return o_begins, o_ends, o_ids[: o.n_data]

    def enqueue(st=stream):
        o_begins, o_ends, o_ids, o = out.take()
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_wordpiece_encode_enqueue(wp._h, ws._h, pu._h, C.byref(rs), unk, C.byref(o), st, C.byref(pending)))

        def finish():
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(o)))
            return o_begins, o_ends, o_ids[: o.n_data]
        return finish

    def cpu(n_s):
        from oracle import oracle as O
        s1, s2 = O.RegexSplit(BERT_WS, "remove"), O.RegexSplit(BERT_PUNCT, "isolate")
        owp = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])
        chain = lambda n: owp(*s2(*s1(rb[:n], re_[:n], begins[:n], ends[:n], chars)[:5])[:5], tok["unk_id"])
        ref = chain(1024)
        t1 = time.perf_counter()
        chain(n_s)
        return ref, time.perf_counter() - t1, int(ends[n_s - 1] - begins[0]), "2 x RegexSplit(PCRE2 JIT) + WordpieceTokenizer restatement"

    workload = (f"config 3: BERT-shaped WordPiece (V=30522, trained in-process), {args.rows} x ~{nbytes}-byte lower-cased zipf "
                f"strings per GPU, fused RegexSplit(\\s+)+RegexSplit(delimiters)+WordpieceTokenizer, inputs and outputs in HBM")
    return dict(step=step, cpu=cpu, n_units=n_chars, out=out, enqueue=enqueue, keep=(d, ws, pu, wp), workload=workload, vocab=len(tok["vocab"]),
                dominant="lookup_words",
                metric="input MB/s encoded (BERT WordPiece, 256-byte strings)", dtype="u8/int32",
                algo=lambda n_tok: n_chars + 4 * n_tok + 16 * args.rows, sample_rows=args.rows)


def make_detokenize(args, lib, dev, rank):
    tok = BpeTok.load(args.tokenizer)
    rows = args.rows if args.rows != 65536 else 16384
    S = 2048
    V = len(tok.vocab)
    rng = np.random.default_rng(3000 + rank)
    ids = rng.integers(0, V - 1, size=(rows, S), dtype=np.int32)
    pad = V - 1                                   # the special token: 1 % of the positions, skipped by the decoder
    ids[rng.random((rows, S)) < 0.01] = pad
    d_ids = torch.as_tensor(ids, device=dev)
    dec = VocabDecoder(skip_tokens=[pad], device=dev.index, lib=lib)
    vconst = list(pack_strings(tok.vocab))
    dec._ensure([d_ids] + vconst)
    lens = (vconst[1] - vconst[0]).astype(np.int64)
    n_out = int(lens[ids[ids != pad]].sum())
    cap = n_out + 64
    class CharsOut:
        """Four output sets used in turn (up to three calls are in flight with --depth 2)."""
        def __init__(self):
            self.sets = []
            for _ in range(4):
                b = torch.empty(rows, dtype=torch.int32, device=dev)
                e = torch.empty(rows, dtype=torch.int32, device=dev)
                c = torch.empty(cap, dtype=torch.uint8, device=dev)
                self.sets.append((b, e, c, L.StringsOut(b.data_ptr(), e.data_ptr(), c.data_ptr(), cap, 0)))
            self.k = 0
            self.last = self.sets[0]

        def take(self):
            self.last = self.sets[self.k % 4]
            self.k += 1
            return self.last

        @property
        def n_chars(self):
            return self.last[3].n_chars

    out = CharsOut()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    pids = C.c_void_p(d_ids.data_ptr())

    def step():
        o_begins, o_ends, o_chars, o = out.take()
        L.check(lib, lib.ovtk_detokenize_run(dec._h, pids, C.c_int64(rows), C.c_int64(S), None, C.c_int64(0), 1, C.byref(o),
                                             L.MEM_DEVICE, stream))
        return o_begins, o_ends, o_chars[: o.n_chars]

    def enqueue(st=stream):
        o_begins, o_ends, o_chars, o = out.take()
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_detokenize_enqueue(dec._h, pids, C.c_int64(rows), C.c_int64(S), None, C.c_int64(0), 1, C.byref(o), st,
                                                 C.byref(pending)))

        def finish():
            L.check(lib, lib.ovtk_detokenize_finish(pending, C.byref(o)))
            return o_begins, o_ends, o_chars[: o.n_chars]
        return finish

    def cpu(n_s):
        from oracle import oracle as O
        def chain(n):
            r = O.vocab_decoder(ids[:n], tok.vocab, [pad])
            bf = O.byte_fallback(*r[2:5])
            fz = O.fuze(r[0], r[1], bf[0], bf[1])
            return fz[0], fz[1], bf[2]
        ref = chain(16)
        t1 = time.perf_counter()
        chain(n_s)
        return ref, time.perf_counter() - t1, n_s * S, "VocabDecoder + ByteFallback + FuzeRagged restatement"

    workload = (f"config 5 chunk: detokenize {rows} x {S} ids (GPT-2-shaped vocabulary, 1 % skipped special ids) per GPU, "
                f"fused VocabDecoder+ByteFallback+FuzeRagged, inputs and outputs in HBM ({n_out} output bytes < 2^31)")
    return dict(step=step, enqueue=enqueue, cpu=cpu, n_units=rows * S, out=out, keep=(d_ids, dec), workload=workload,
                metric="token ids/s detokenized (seq 2048)", dtype="int32/u8", unit="Mtok/s", rows=rows, dominant="detokenize",
                algo=lambda _n: 4 * rows * S + n_out + 8 * rows, sample_rows=min(rows, 1024), is_detok=True, n_out=n_out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--bytes", type=int, default=512)
    ap.add_argument("--text", default="zipf", choices=["zipf", "uniform", "mixed"])
    ap.add_argument("--tokenizer", default="gpt2")
    ap.add_argument("--no-memo", action="store_true", help="config 2: BPETokenizer with cache_capacity=0 (no piece memo)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the all-gather (rank-local consumer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hog", type=int, default=0, help="debug: occupy CU slots with N idle 512-thread blocks on a side stream during "
                                                        "every step (stands in for RCCL's all-gather kernel; tools/cu_hog.hip)")
    ap.add_argument("--row-tickets", type=int, default=-1, help="ovtk_set_row_tickets(n); default 0 (static row assignment)")
    ap.add_argument("--hog-lds", type=int, default=0, help="debug: dynamic LDS bytes per hog block")
    ap.add_argument("--lib", default=None, help="debug: load this build of libovtk_amd.so")
    ap.add_argument("--no-alone-leg", action="store_true", help="skip the one-stream leg behind roofline.alone (profile runs: "
                    "rocprofv3's per-kernel average then covers the overlapped launches only)")
    ap.add_argument("--depth", type=int, default=2, help="batches launched ahead of the one being completed (two-half calls)")
    ap.add_argument("--exchange-stream", type=int, default=1, help="1: pack/unpack of the exchange on a HIP stream of their own")
    ap.add_argument("--streams", type=int, default=3, help="consecutive batches alternate between this many HIP streams (two-half calls)")
    ap.add_argument("--sync", action="store_true", help="one blocking ovtk_encode_run per step (no launch/complete overlap)")
    ap.add_argument("--force-exchange", action="store_true", help="N = 1: still run the exchange, in a one-rank RCCL group (debug)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_exchange
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        from openvino_tokenizers_amd.distributed import ShardExchange

    lib = L.load(args.lib)
    wl = {2: make_encode_bpe, 3: make_encode_wordpiece, 4: make_encode_llama3, 5: make_detokenize}[args.config](args, lib, dev, rank)
    is_detok = wl.get("is_detok", False)
    # N > 1: every rank's ragged ids are all-gathered (RCCL), one batch behind the encode so that the gather of batch k
    # travels over xGMI while batch k + 1 is encoded; flush() completes the last one inside the timed region.
    exchange = None
    if (world > 1 or args.force_exchange) and not args.no_gather and not is_detok:
        xstream = torch.cuda.Stream(dev) if args.exchange_stream else None
        # equal shards of the same text model: the ranks' id counts differ by well under 0.1 %, so 3 % of padding on the
        # wire (instead of the class's default 12.5 %) never triggers a re-gather and the gather moves 8 % fewer bytes
        exchange = ShardExchange(wl.get("rows", args.rows) * world, wl["vocab"], dev, lib=lib, stream=xstream, headroom=1.03)

    # A step = one batch through the hot path.  Where the op has the two-half form the host launches batch k, then
    # completes batch k-1 (status check, and with N > 1 its exchange) while the GPU works on k: the reference's
    # evaluate() semantics per batch, without the GPU idling while the host reads a status word.  Consecutive batches go to
    # alternating HIP streams: the latency-bound merge kernel of one batch then shares the CUs with the issue-bound
    # lookup kernel of the next (per-launch durations grow, the step shrinks).  --sync: one blocking call per step.
    row_tickets = max(args.row_tickets, 0)
    L.check(lib, lib.ovtk_set_row_tickets(row_tickets))
    inflight = []
    side_streams = [torch.cuda.Stream(dev) for _ in range(max(args.streams - 1, 0))]
    stream_ptrs = [C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)] + [C.c_void_p(x.cuda_stream) for x in side_streams]
    launched = [0]
    hog = None
    if args.hog:
        hog_lib = C.CDLL(str(ROOT / "tools" / "build" / "libcuhog.so"))
        hog_stream = torch.cuda.Stream(dev)
        hog = lambda: hog_lib.cu_hog(args.hog, 512, args.hog_lds, C.c_double(300.0), C.c_void_p(hog_stream.cuda_stream))  # noqa: E731

    def complete(finish):
        res = finish()
        return exchange.submit(*res) if exchange is not None else res

    def step():
        if hog is not None:
            hog()
        if "enqueue" in wl and not args.sync:
            inflight.append(wl["enqueue"](stream_ptrs[launched[0] % len(stream_ptrs)]))
            launched[0] += 1
            return complete(inflight.pop(0)) if len(inflight) > args.depth else None
        return complete(wl["step"])

    def drain():
        while inflight:
            complete(inflight.pop(0))
        if exchange is not None:
            exchange.flush()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    lib.ovtk_profile_reset()
    lib.ovtk_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()   # every one of the K batches is complete (and, N > 1, gathered on every rank) before the clock stops
    barrier()
    dt = time.perf_counter() - t0
    lib.ovtk_profile_enable(0)
    n_units = wl["n_units"]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nb = torch.tensor([n_units], dtype=torch.int64, device=dev)
        dist.all_reduce(nb)
        total_units = int(nb.item())
    else:
        total_units = n_units

    n_out = int(wl["out"].n_chars) if is_detok else int(wl["out"].n_data)
    ms_per_step = dt / args.steps * 1e3
    value = total_units * args.steps / dt / 1e6  # MB/s (or Mtok/s), whole job

    # ---- roofline of the dominant kernel (HIP events recorded by the library on the launch stream)
    buf = C.create_string_buffer(8192)
    lib.ovtk_profile_dump(buf, 8192)
    prof = {ln.split()[0]: (float(ln.split()[1]), int(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.strip()}
    kernels = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}
    per_step = {k: v[0] / args.steps for k, v in prof.items()}  # ms of each kernel family per step
    roofline = None
    if per_step:
        # The same kernels with the chip to themselves: with several streams the durations above are of launches that
        # shared the CUs with the neighbouring batches' kernels (that is the point of the streams), and which of two
        # kernels of similar length looks longer then changes from run to run.  A short one-stream leg outside the timed
        # region gives each kernel's own duration; the dominant kernel is the longest one THERE.
        alone = {}
        if "enqueue" in wl and not args.sync and len(stream_ptrs) > 1 and not args.no_alone_leg:
            lib.ovtk_profile_reset()
            lib.ovtk_profile_enable(1)
            for _ in range(20):
                wl["enqueue"](stream_ptrs[0])()   # launch, then finish: no exchange in this leg
            torch.cuda.synchronize()
            lib.ovtk_profile_enable(0)
            lib.ovtk_profile_dump(buf, 8192)
            alone = {ln.split()[0]: (float(ln.split()[1]), int(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.strip()}
            alone = {k: v for k, v in alone.items() if k in per_step and v[1]}
        hint = wl.get("dominant")
        if alone:
            dom = max(alone, key=lambda k: alone[k][0])
        elif hint in per_step and per_step[hint] >= 0.8 * max(per_step.values()):
            dom = hint   # profile runs (--no-alone-leg): the workload's usual dominant kernel unless another is clearly longer
        else:
            dom = max(per_step, key=per_step.get)
        launches_per_step = max(1, round(prof[dom][1] / args.steps))
        k_ms = prof[dom][0] / max(prof[dom][1], 1)
        algo_bytes = wl["algo"](n_out)  # algorithmic bytes of one pass
        achieved = algo_bytes / launches_per_step / (k_ms * 1e-3) / 1e9
        traffic = None
        if PMC_FILE.exists():  # per-launch FETCH_SIZE (doubled: gfx950 correction) + WRITE_SIZE of this kernel, see profiles/README.md
            pmc = json.loads(PMC_FILE.read_text())
            traffic = pmc.get(f"config{args.config}", {}).get(dom)
        roofline = {"bound": "hbm", "kernel": KERNEL_NAMES.get(dom, dom), "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "kernel_ms": round(k_ms, 4), "algorithmic_bytes": algo_bytes,
                    "launches_per_step": launches_per_step,
                    "all_kernels_ms_per_step": round(sum(per_step.values()), 4)}
        step_gbs = algo_bytes * (total_units / n_units) / (ms_per_step * 1e-3) / 1e9 / world
        roofline["step"] = {"achieved": round(step_gbs, 2), "frac": round(step_gbs / HBM_PEAK_GBS, 5),
                            "note": "algorithmic bytes of one batch / ms_per_step, per GPU: every kernel of the path, overlapped as run"}
        if alone:
            a_ms = alone[dom][0] / alone[dom][1]
            a_gbs = algo_bytes / launches_per_step / (a_ms * 1e-3) / 1e9
            roofline["alone"] = {"kernel_ms": round(a_ms, 4), "achieved": round(a_gbs, 2), "frac": round(a_gbs / HBM_PEAK_GBS, 5),
                                 "note": "same kernel without a neighbouring batch on the CUs (20 one-stream launches, untimed leg)"}

    if rank != 0:
        if exchange is not None:
            exchange.close()
        if dist_on:
            dist.destroy_process_group()
        return

    # ---- parity spot-check + CPU baseline (oracle = "port" of the reference's algorithm), rank 0 only
    cpu_baseline, parity = None, None
    if not args.no_cpu_baseline and world == 1:  # N > 1: the ranks are done once the line is printed
        n_s = wl["sample_rows"]
        ref, cdt, sample_units, what = wl["cpu"](n_s)
        passes = 1
        while cdt < 10.0 and passes < 16:  # about 10 s of CPU work in all: repeat the sample (same rows, cache stays warm)
            _, more, units, _ = wl["cpu"](n_s)
            cdt, sample_units, passes = cdt + more, sample_units + units, passes + 1
        b, e, payload = wl["step"]()
        n_chk = len(ref[1])
        got_ends = e[:n_chk].cpu().numpy()
        got = payload[: int(got_ends[-1])].cpu().numpy()
        parity = bool(np.array_equal(ref[1], got_ends) and np.array_equal(ref[2], got))
        if world == 1:
            cpu_baseline = {"value": round(sample_units / cdt / 1e6, 2), "unit": wl.get("unit", "MB/s"), "cores": 1, "kind": "port",
                            "sample": f"{passes} pass(es) over the first {n_s} rows of the same batch ({sample_units} "
                                      f"{'ids' if is_detok else 'bytes'} in all), {what}, {cdt:.2f} s",
                            "host_cpus": os.cpu_count()}

    line = {
        "metric": wl["metric"], "value": round(value, 1), "unit": wl.get("unit", "MB/s"),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
        "config": {"workload": wl["workload"],
                   "row_tickets": row_tickets,
                   "host_loop": (f"launch batch k, then complete batch k-{args.depth} (two-half calls), batches alternate between {len(stream_ptrs)} "
                                 f"HIP stream(s)" if "enqueue" in wl and not args.sync else "one blocking call per batch"),
                   "rows_per_gpu": wl.get("rows", args.rows), "units_per_gpu": n_units, "outputs_per_gpu": n_out,
                   "exchange": ("none (1 GPU)" if exchange is None and world == 1 else ("none (rank-local consumer)" if exchange is None else
                                                                    f"all-gather of ragged ids over RCCL, {exchange.id_bytes}-byte ids on the wire, "
                                                                    f"gather overlapped with the next encode, unpack one batch later ({exchange.regathers} re-gathers)"))},
        "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_prefix_bit_exact": parity,
        "kernel_ms": kernels,
    }
    print(json.dumps(line))
    if exchange is not None:
        exchange.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
