#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): input MB/s encoded, GPT-2-shaped byte-level BPE, 65 536 x ~512-byte strings
per GPU (config 2), inputs resident in HBM, fused RegexSplit+BPETokenizer through the C ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W]           (N > 1: launched by torch.distributed.run)
    python bench.py --config 3      BERT-shaped WordPiece, 65 536 x ~256-byte strings (fused BERT split + WordPiece)
    python bench.py --config 4      one 8-GPU shard of the Llama-3-shaped config (fused Llama-3 split + BPE, mixed scripts)
    python bench.py --config 5      detokenizer (VocabDecoder + ByteFallback + FuzeRagged fused), rows x 2048 ids
    python bench.py --config 1      small-batch latency: 32 x ~128-byte strings, one blocking call per step
    python bench.py --config r2d    RaggedToDense on config 2's ids ([65 536, T] ids + mask)
    python bench.py --config vocab_encoder   VocabEncoder on ~3 M words

One "step" = one pass of the hot path over one batch.  The timed loop ROTATES `--batches` (default 8) distinct input
batches -- more than 256 MB of text in all, so a step reads its text from HBM, not from the 256 MB Infinity Cache a single
re-fed batch would sit in.  With N > 1 every rank encodes its own shards of the same size (weak scaling) and the step
includes the all-gather of the ragged token ids over RCCL/xGMI (openvino_tokenizers_amd/distributed.py);
`value` = units of all ranks / max-over-ranks time.
Prints ONE JSON line on rank 0.  At N = 1 the line also carries, measured after the timed region: the per-kernel
one-stream leg behind `roofline`, `stress` (uniform-random text; piece memo off), `end_to_end` (pinned host buffers in
and out over PCIe) and the CPU baselines (1 core, and all host cores sharing one tokenizer).  The oracle is used only for
the cpu_baseline legs and a parity spot-check.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
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
from openvino_tokenizers_amd.ops import (BPETokenizer, RegexSplit, SpecialTokensSplit, VocabDecoder, VocabEncoder, WordpieceTokenizer)  # noqa: E402
from tools.harness import BERT_PUNCT, BERT_WS, BpeTok, pack_strings  # noqa: E402
from tools.make_tokenizers import load_tokenizer  # noqa: E402
from tools.workloads import TextModel, ragged_rows  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
KERNEL_NAMES = {"lookup_span": "lookup_span_kernel", "lookup_rows": "lookup_rows_kernel", "regex_split": "regex_sparse_kernel", "lookup_fused": "lookup_kernel<kFused>", "lookup_pieces": "lookup_kernel<kPieces>", "shard_pack": "shard_pack_kernel",
                "shard_unpack": "shard_unpack_kernel", "scan_rows": "scan_kernel", "lookup_flat": "piece_lookup_kernel",
                "lookup_words": "lookup_span_kernel<BERT words>", "wordpiece_deferred": "wordpiece_deferred_kernel",
                "bpe_merge": "merge_kernel", "bpe_exact": "exact_kernel", "compact": "compact_kernel",
                "prep_rows": "prep_rows_kernel", "count_scan": "count_scan_kernel", "split_count": "split_kernel<0>",
                "split_write": "split_kernel<1>", "ragged_to_dense": "ragged_to_dense_kernel", "vocab_encoder": "vocab_encoder_kernel",
                "detokenize": "decode_write_kernel", "decode_count": "decode_count_kernel", "decode_scan": "tile_{reduce,scan,apply}_kernel<UnitLen>"}
PMC_FILE = ROOT / "profiles" / "latest_pmc.json"  # HBM traffic of the dominant kernels from a separate rocprofv3 --pmc run


class TextBatches:
    """`n` distinct text batches (one string per row) resident in HBM; batch k of the rotation is k % n."""

    def __init__(self, model, rows, nbytes, seeds, dev, lower=False):
        self.rows, self.dev, self.host, self.d, self.rs, self.n_chars = rows, dev, [], [], [], []
        rb, re_ = ragged_rows(rows)
        for seed in seeds:
            begins, ends, chars = model.batch(rows, nbytes, seed=seed)
            if lower:
                chars = np.frombuffer(chars.tobytes().lower(), np.uint8).copy()
            d = [torch.as_tensor(x, device=dev) for x in (rb, re_, begins, ends, chars)]
            self.host.append((rb, re_, begins, ends, chars) if not self.host else None)  # batch 0 stays on the host: CPU legs
            self.d.append(d)
            self.n_chars.append(int(len(chars)))
            self.rs.append(L.RaggedStrings(d[0].data_ptr(), d[1].data_ptr(), rows,
                                           L.Strings(d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), rows, len(chars))))
        self.n = len(self.rs)
        self.cap = max(self.n_chars)


class IdOut:
    """Six sets of ragged-id output buffers used in turn: up to three batches are being encoded (--depth 2) while,
    with N > 1, the exchange still holds the two before them (gathering / unpacking -- a shard that outgrew the agreed
    pad is packed again from its local ids)."""
    SETS = 6

    def __init__(self, rows, cap, dev):
        self.SETS = max(6, int(os.environ.get("OVTK_BENCH_DEPTH", "2")) + 4)   # (set by main() from --depth)
        self.sets = []
        for _ in range(self.SETS):
            b = torch.empty(rows, dtype=torch.int32, device=dev)
            e = torch.empty(rows, dtype=torch.int32, device=dev)
            ids = torch.empty(cap, dtype=torch.int32, device=dev)
            self.sets.append((b, e, ids, L.RaggedI32Out(b.data_ptr(), e.data_ptr(), ids.data_ptr(), cap, 0, 0)))
        self.k = 0

    def take(self):
        s = self.sets[self.k % self.SETS]
        self.k += 1
        return s


class EncodeWorkload:
    """A ragged-strings -> ragged-ids workload over rotating batches; `launch(rs, skips, out, stream, pending)` /
    `run(rs, out, stream)` are the library calls of the concrete op chain."""
    unit = "MB/s"
    dtype = "u8/int32"

    def __init__(self, lib, dev, batches):
        self.lib, self.dev, self.batches = lib, dev, batches
        self.out = IdOut(batches.rows, batches.cap, dev)
        self.n_out = {}  # batch index -> ids produced
        self.stream0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def units(self, i):
        return self.batches.n_chars[i % self.batches.n]

    def step(self, i):
        k = i % self.batches.n
        o_begins, o_ends, o_ids, o = self.out.take()
        L.check(self.lib, self.run(self.batches.rs[k], o, self.stream0))
        self.n_out[k] = int(o.n_data)
        return o_begins, o_ends, o_ids[: o.n_data]

    def enqueue(self, i, st=None):
        k = i % self.batches.n
        o_begins, o_ends, o_ids, o = self.out.take()
        pending = C.c_void_p()
        L.check(self.lib, self.launch(self.batches.rs[k], o, st or self.stream0, pending))

        def finish():
            L.check(self.lib, self.lib.ovtk_encode_finish(pending, C.byref(o)))
            self.n_out[k] = int(o.n_data)
            return o_begins, o_ends, o_ids[: o.n_data]
        return finish

    def algo(self):
        """algorithmic bytes of one pass, A_enc = N_c + 4 N_t + 16 B, averaged over the rotation."""
        ks = sorted(self.n_out)
        return float(np.mean([self.batches.n_chars[k] + 4 * self.n_out[k] + 16 * self.batches.rows for k in ks]))

    def kernel_algo(self, dom):
        """The algorithmic bytes of the dominant kernel ALONE, where it is one of the lookup kernels: it reads the text and
        the strings' begins / ends and writes (stages) the ids; the rows' begins / ends are written by compact_kernel."""
        if not dom.startswith("lookup"):
            return None
        ks = sorted(self.n_out)
        return float(np.mean([self.batches.n_chars[k] + 4 * self.n_out[k] + 8 * self.batches.rows for k in ks]))

    def mean_out(self):
        return float(np.mean(list(self.n_out.values())))


class BpeEncode(EncodeWorkload):
    def __init__(self, args, lib, dev, rank, tokenizer, kind, rows, nbytes, seed0, no_memo=False, n_batches=None, cache_capacity=None,
                 pattern=None, model=None, memo_learn=None):
        self.tok = BpeTok.load(tokenizer)
        if pattern:   # the same tables behind another model's split pattern (tools/workloads.py MODEL_PATTERNS)
            self.tok.pattern = pattern
        model = model or TextModel(1234, kind)
        nb = n_batches or args.batches
        super().__init__(lib, dev, TextBatches(model, rows, nbytes, [seed0 + 100 * rank + 7 * j for j in range(nb)], dev))
        self.split = RegexSplit("isolate", device=dev.index, lib=lib)
        attrs = dict(self.tok.attrs)
        if no_memo:
            attrs["cache_capacity"] = 0
        elif cache_capacity is not None:
            attrs["cache_capacity"] = cache_capacity
        self.cache_capacity = attrs.get("cache_capacity", 20000)
        # ovtk_bpe_params::memo_learn (include/ovtk_amd.h): 0 = the library's default, the first level learns up to
        # max(cache_capacity, the store's capacity) pieces; -1 = exactly cache_capacity, the reference's count
        self.memo_learn = int(memo_learn if memo_learn is not None else (getattr(args, "memo_learn", 0) or 0))
        self.bpe = BPETokenizer(**attrs, device=dev.index, lib=lib, memo_learn=self.memo_learn)
        self.split._ensure(self.tok.pattern_u8())
        self.bpe._ensure(self.batches.d[0] + self.tok.consts)
        self.vocab = len(self.tok.vocab)

    def memo(self):
        """Entries of the piece memo: from the vocabulary at create / learned from the text since (memo_learn says how many at most)."""
        fixed, learned = C.c_int64(), C.c_int64()
        L.check(self.lib, self.lib.ovtk_bpe_memo_entries(self.bpe._h, C.byref(fixed), C.byref(learned)))
        stored, cap = C.c_int64(), C.c_int64()
        L.check(self.lib, self.lib.ovtk_bpe_store_entries(self.bpe._h, C.byref(stored), C.byref(cap)))
        return {"fixed": int(fixed.value), "learned": int(learned.value), "cache_capacity": int(self.cache_capacity),
                "memo_learn": self.memo_learn,
                "note": ("memo_learn = 0, the library's default: the first level learns up to max(cache_capacity, the store's capacity) pieces of at most 15 bytes "
                         "and 6 ids (3 when an id needs more than 16 bits); -1: exactly cache_capacity pieces, the reference's count -- the "
                         "stress leg `reference_cache_count` runs the headline text that way (include/ovtk_amd.h ovtk_bpe_params)"),
                "store": {"entries": int(stored.value), "capacity": int(cap.value),
                          "note": "second level, probed by merge_kernel only: pieces it had to merge once (include/ovtk_amd.h ovtk_set_memo_store)"}}

    def run(self, rs, o, st):
        return self.lib.ovtk_encode_run(self.split._h, self.bpe._h, C.byref(rs), None, C.byref(o), L.MEM_DEVICE, st)

    def launch(self, rs, o, st, pending):
        return self.lib.ovtk_encode_enqueue(self.split._h, self.bpe._h, C.byref(rs), None, C.byref(o), st, C.byref(pending))

    def enqueue_wire(self, i, st, wire, ex):
        """The encode of batch i straight into a send wire of the exchange (ovtk_encode_enqueue_wire)."""
        k = i % self.batches.n
        pending = C.c_void_p()
        L.check(self.lib, self.lib.ovtk_encode_enqueue_wire(self.split._h, self.bpe._h, C.byref(self.batches.rs[k]), None,
                                                            C.c_void_p(wire.t.data_ptr()), C.c_int64(ex.max_rows), C.c_int64(wire.pad),
                                                            ex.id_bytes, st or self.stream0, C.byref(pending)))
        o = L.RaggedI32Out(None, None, None, 0, 0, 0)

        def finish():
            L.check(self.lib, self.lib.ovtk_encode_finish(pending, C.byref(o)))
            self.n_out[k] = int(o.n_data)
            return wire, (lambda w: self.enqueue_wire(i, st, w, ex)())
        return finish

    def cpu_chain(self):
        from oracle import oracle as O
        orc, ors = self.tok.oracle(), O.RegexSplit(self.tok.pattern, "isolate")
        return (lambda rb, re_, b, e, c: orc(*ors(rb, re_, b, e, c)[:5])), "RegexSplit(PCRE2 JIT)+BPETokenizer restatement, piece cache warm"


class SpecialTextModel(TextModel):
    """The zipf text with a special token written over the middle of one row in a hundred (same offsets: 13 bytes replaced)."""
    TOKEN = b"<|endoftext|>"

    def batch(self, n_rows, target_len, seed=None):
        b, e, c = super().batch(n_rows, target_len, seed=seed)
        rng = np.random.default_rng(0 if seed is None else seed + 1)
        tok = np.frombuffer(self.TOKEN, np.uint8)
        for i in np.flatnonzero(rng.random(n_rows) < 0.01):
            if e[i] - b[i] > 2 * len(tok):
                at = int(b[i]) + int(rng.integers(0, e[i] - b[i] - len(tok)))
                c[at:at + len(tok)] = tok
        return b, e, c


class PipelineEncode(BpeEncode):
    """The graph a converted GPT-2 tokenizer runs (python/openvino_tokenizers/tokenizer_pipeline.py:1613-1636) from
    the decomposed string tensor on: SpecialTokensSplit -> RegexSplit -> BPETokenizer (one call: ovtk_encode_special_enqueue) ->
    Truncate(max_length 1024) -> CombineSegments -> RaggedToDense x 2 = input_ids [B, T] and attention_mask [B, T], T = the batch's
    longest row (the PaddingStep's ReduceMax) -- ONE library call per batch (ovtk_encode_dense_enqueue / _finish: the encode's last
    pass writes the dense tensors, no ragged ids tensor), or with --pipeline-calls 2 the encode and ovtk_encode_tail_run."""
    MAX_LENGTH = 1024

    def __init__(self, args, lib, dev, rank):
        super().__init__(args, lib, dev, rank, "gpt2", "zipf", args.rows, args.bytes, 1000, model=SpecialTextModel(1234, "zipf"))
        self.special_pattern = r"(\<\|endoftext\|\>)"   # what SpecialTokensSplitStep builds for this token (tokenizer_pipeline.py:138-159, quote_meta)
        self.special = SpecialTokensSplit(device=dev.index, lib=lib)
        self.special._ensure(np.frombuffer(self.special_pattern.encode(), np.uint8))
        self.pad_id = int(self.tok.added[b"<|endoftext|>"])
        rows = self.batches.rows
        self.tcap = 256 if args.bytes <= 600 else self.MAX_LENGTH   # room per row in the dense outputs (a row of n bytes has at most n ids)
        self.dense = [(torch.empty(rows * self.tcap, dtype=torch.int32, device=dev), torch.empty(rows * self.tcap, dtype=torch.uint8, device=dev))
                      for _ in range(self.out.SETS)]
        self.dk = 0
        self.seg_ids = np.zeros(1, np.int32)
        self.width = {}
        self.two_calls = getattr(args, "pipeline_calls", 1) == 2

    def run(self, rs, o, st):
        return self.lib.ovtk_encode_special_run(self.special._h, self.split._h, self.bpe._h, C.byref(rs), None, C.byref(o), L.MEM_DEVICE, st)

    def launch(self, rs, o, st, pending):
        return self.lib.ovtk_encode_special_enqueue(self.special._h, self.split._h, self.bpe._h, C.byref(rs), None, C.byref(o), st, C.byref(pending))

    def _tail(self, k, res, st):
        b, e, ids = res
        rows = self.batches.rows
        seg = (L.RaggedI32 * 1)(L.RaggedI32(b.data_ptr(), e.data_ptr(), ids.data_ptr(), rows, len(ids)))
        p = L.EncodeTailParams(C.cast(seg, C.c_void_p), 1, self.seg_ids.ctypes.data_as(C.c_void_p), 0, -1, C.c_int32(self.MAX_LENGTH), b"right",
                               b"longest_first", -1, self.pad_id, 0, 1)
        d_ids, d_mask = self.dense[self.dk % len(self.dense)]
        self.dk += 1
        T = C.c_int32(0)
        L.check(self.lib, self.lib.ovtk_encode_tail_run(C.byref(p), C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_mask.data_ptr()), None,
                                                        C.c_int64(rows * self.tcap), C.byref(T), L.MEM_DEVICE, self.dev.index, st or self.stream0))
        self.width[k] = int(T.value)
        n = rows * int(T.value)
        return d_ids[:n].view(rows, int(T.value)), d_mask[:n].view(rows, int(T.value))

    def step(self, i):
        if self.two_calls:
            return self._tail(i % self.batches.n, super().step(i), self.stream0)
        return self.enqueue(i)()

    def enqueue(self, i, st=None):
        if self.two_calls:   # (--pipeline-calls 2: the ragged ids between two library calls, as rounds 1-4 could do it)
            fin = super().enqueue(i, st)
            return lambda: self._tail(i % self.batches.n, fin(), st)
        k = i % self.batches.n
        rows = self.batches.rows
        d_ids, d_mask = self.dense[self.dk % len(self.dense)]
        self.dk += 1
        p = L.DenseParams(C.c_int32(self.MAX_LENGTH), 0, 1, self.pad_id, -1, None, 0, None, 0)
        pending = C.c_void_p()
        L.check(self.lib, self.lib.ovtk_encode_dense_enqueue(self.special._h, self.split._h, self.bpe._h, C.byref(self.batches.rs[k]), None, C.byref(p),
                                                             C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_mask.data_ptr()), C.c_int64(rows * self.tcap),
                                                             st or self.stream0, C.byref(pending)))

        def finish(_keep=p):
            width, n_ids = C.c_int32(0), C.c_int64(0)
            L.check(self.lib, self.lib.ovtk_encode_dense_finish(pending, C.byref(width), C.byref(n_ids)))
            self.n_out[k] = int(n_ids.value)
            self.width[k] = int(width.value)
            n = rows * int(width.value)
            return d_ids[:n].view(rows, int(width.value)), d_mask[:n].view(rows, int(width.value))
        return finish

    # (algo(): the fused encode's algorithmic bytes, SURVEY 8d -- the roofline fields stay comparable with config 2's; the dense
    # tensors are 5 bytes per cell of [rows, T] on top, written by compact_kernel<DenseSink>, not by the dominant kernel)

    def cpu_chain(self):
        from oracle import oracle as O
        orc, ors, osp = self.tok.oracle(), O.RegexSplit(self.tok.pattern, "isolate"), O.SpecialTokensSplit(self.special_pattern)

        def chain(rb, re_, b, e, c):
            s = osp(rb, re_, b, e, c)
            ob, oe, ids = orc(*ors(*s[:5], skips=s[5])[:5])
            (tb, te), = O.truncate([(ob, oe)], self.MAX_LENGTH, "right", "longest_first")
            return tb, te, ids
        return chain, "SpecialTokensSplit + RegexSplit (PCRE2 JIT) + BPETokenizer restatement + Truncate, piece cache warm"

    def check(self, ref, res):
        """The oracle chain's ragged ids of the first rows against the dense outputs: ids, padding, mask."""
        tb, te, ids = ref
        dense, mask = (x.cpu().numpy() for x in res)
        for i in range(len(tb)):
            n = int(te[i] - tb[i])
            if not (np.array_equal(dense[i, :n], ids[tb[i]:te[i]]) and (dense[i, n:] == self.pad_id).all() and mask[i, :n].all() and not mask[i, n:].any()):
                return False
        return True


class WordpieceEncode(EncodeWorkload):
    def __init__(self, args, lib, dev, rank):
        self.tok = load_tokenizer("bert")
        nbytes = args.bytes if args.bytes != 512 else 256
        self.nbytes = nbytes
        model = TextModel(1234, "zipf")
        super().__init__(lib, dev, TextBatches(model, args.rows, nbytes, [2000 + 100 * rank + 7 * j for j in range(args.batches)], dev,
                                               lower=True))  # BERT-uncased normalisation is upstream of this path
        self.ws = RegexSplit("remove", device=dev.index, lib=lib)
        self.pu = RegexSplit("isolate", device=dev.index, lib=lib)
        self.wp = WordpieceTokenizer(self.tok["suffix_indicator"], self.tok["max_bytes_per_word"], device=dev.index, lib=lib)
        self.ws._ensure(BERT_WS)
        self.pu._ensure(BERT_PUNCT)
        consts = list(pack_strings(self.tok["vocab"])) + [np.asarray(self.tok["unk_id"], np.int32)]
        self.wp._ensure(self.batches.d[0] + consts)
        self.unk = C.c_int32(int(self.tok["unk_id"]))
        self.vocab = len(self.tok["vocab"])

    def run(self, rs, o, st):
        return self.lib.ovtk_wordpiece_encode_run(self.wp._h, self.ws._h, self.pu._h, C.byref(rs), self.unk, C.byref(o), L.MEM_DEVICE, st)

    def launch(self, rs, o, st, pending):
        return self.lib.ovtk_wordpiece_encode_enqueue(self.wp._h, self.ws._h, self.pu._h, C.byref(rs), self.unk, C.byref(o), st,
                                                      C.byref(pending))

    def cpu_chain(self):
        from oracle import oracle as O
        tok = self.tok
        s1, s2 = O.RegexSplit(BERT_WS, "remove"), O.RegexSplit(BERT_PUNCT, "isolate")
        owp = O.WordpieceTokenizer(tok["vocab"], tok["suffix_indicator"], tok["max_bytes_per_word"])
        return (lambda rb, re_, b, e, c: owp(*s2(*s1(rb, re_, b, e, c)[:5])[:5], tok["unk_id"])), \
            "2 x RegexSplit(PCRE2 JIT) + WordpieceTokenizer restatement"


def make_workload(args, lib, dev, rank):
    """-> (workload object or dict, description pieces)."""
    cfg = args.config
    if cfg == "2":
        from tools.workloads import MODEL_PATTERNS
        w = BpeEncode(args, lib, dev, rank, args.tokenizer, args.text, args.rows, args.bytes, 1000, no_memo=args.no_memo,
                      pattern=MODEL_PATTERNS.get(args.pattern),
                      cache_capacity=getattr(args, "cache_capacity", None), memo_learn=getattr(args, "memo_learn", None))
        w.metric = "input MB/s encoded (GPT-2 BPE, 512-byte strings)"
        w.dominant_hint = "lookup_span"
        w.workload = (f"config 2: GPT-2-shaped byte-level BPE (V=50257, 50000 merges, trained in-process), {args.rows} x "
                      f"~{args.bytes}-byte {args.text} strings per GPU and batch, {w.batches.n} distinct batches in rotation "
                      f"({sum(w.batches.n_chars) / 1e6:.0f} MB of text), fused RegexSplit+BPETokenizer, inputs and outputs in HBM"
                      + (", piece memo disabled (cache_capacity=0)" if args.no_memo else ""))
        w.sample_rows = args.rows
        return w
    if cfg == "4":
        rows = args.rows if args.rows != 65536 else 131072  # 1 M rows / 8 GPUs
        from tools.workloads import MODEL_PATTERNS
        w = BpeEncode(args, lib, dev, rank, "llama3", "mixed", rows, args.bytes, 4000, pattern=MODEL_PATTERNS.get(args.pattern))
        w.metric = "input MB/s encoded (Llama-3 BPE, 512-byte mixed-script strings)"
        w.dominant_hint = "lookup_span"
        w.workload = (f"config 4 shard: Llama-3-shaped byte-level BPE (V=128256, 127999 merges, trained in-process), {rows} x "
                      f"~{args.bytes}-byte mixed-script strings per GPU and batch, {w.batches.n} distinct batches in rotation "
                      f"({sum(w.batches.n_chars) / 1e6:.0f} MB of text), fused RegexSplit (tiktoken-style pattern) + BPETokenizer, "
                      f"inputs and outputs in HBM")
        w.sample_rows = min(rows, 16384)
        return w
    if cfg == "pipeline":
        w = PipelineEncode(args, lib, dev, rank)
        w.metric = "input MB/s through the tokenizer graph (GPT-2 BPE, 512-byte strings -> input_ids, attention_mask)"
        w.dominant_hint = "lookup_span"
        w.workload = (f"the converted GPT-2 graph (tokenizer_pipeline.py:1613-1636): SpecialTokensSplit (<|endoftext|>, in one row of a hundred) "
                      f"-> RegexSplit -> BPETokenizer (V=50257) -> Truncate({w.MAX_LENGTH}) -> CombineSegments -> RaggedToDense x 2, "
                      f"{args.rows} x ~{args.bytes}-byte strings per GPU and batch, {w.batches.n} distinct batches in rotation, "
                      + ("two library calls per batch (ovtk_encode_special_enqueue / _finish, ovtk_encode_tail_run)" if w.two_calls else
                         "one library call per batch (ovtk_encode_dense_enqueue / _finish: no ragged ids tensor)") + ", inputs and outputs in HBM")
        w.sample_rows = args.rows
        return w
    if cfg == "3":
        w = WordpieceEncode(args, lib, dev, rank)
        w.metric = "input MB/s encoded (BERT WordPiece, 256-byte strings)"
        w.dominant_hint = "lookup_words"
        w.workload = (f"config 3: BERT-shaped WordPiece (V=30522, trained in-process), {args.rows} x ~{w.nbytes}-byte lower-cased zipf "
                      f"strings per GPU and batch, {w.batches.n} distinct batches in rotation, fused RegexSplit(\\s+)+RegexSplit(delimiters)+"
                      f"WordpieceTokenizer, inputs and outputs in HBM")
        w.sample_rows = args.rows
        return w
    raise SystemExit(f"unknown config {cfg}")


# ---------------------------------------------------------------------------------------------- other ops
class Detokenize:
    """Config 5 chunk: VocabDecoder + ByteFallback + FuzeRagged fused, rows x 2048 ids, rotating id batches."""
    unit, dtype, metric, dominant_hint = "Mtok/s", "int32/u8", "token ids/s detokenized (seq 2048)", "detokenize"

    def __init__(self, args, lib, dev, rank):
        self.lib, self.dev = lib, dev
        tok = BpeTok.load(args.tokenizer)
        self.tok = tok
        self.rows = rows = args.rows if args.rows != 65536 else 16384
        self.S = S = 2048
        V = len(tok.vocab)
        self.pad = pad = V - 1                       # the special token: 1 % of the positions, skipped by the decoder
        nb = max(2, min(args.batches, 4))            # 4 x 134 MB of ids + 4 x ~0.55 GB of output: beyond the Infinity Cache
        self.ids_host, self.d_ids, self.n_out = [], [], []
        vconst = list(pack_strings(tok.vocab))
        lens = (vconst[1] - vconst[0]).astype(np.int64)
        for j in range(nb):
            rng = np.random.default_rng(3000 + 100 * rank + j)
            ids = rng.integers(0, V - 1, size=(rows, S), dtype=np.int32)
            ids[rng.random((rows, S)) < 0.01] = pad
            self.d_ids.append(torch.as_tensor(ids, device=dev))
            self.n_out.append(int(lens[ids[ids != pad]].sum()))
            if j == 0:
                self.ids_host = ids
        self.dec = VocabDecoder(skip_tokens=[pad], device=dev.index, lib=lib)
        self.dec._ensure([self.d_ids[0]] + vconst)
        cap = max(self.n_out) + 64
        self.sets = []
        for _ in range(4):  # four output sets used in turn (up to three calls are in flight with --depth 2)
            b = torch.empty(rows, dtype=torch.int32, device=dev)
            e = torch.empty(rows, dtype=torch.int32, device=dev)
            c = torch.empty(cap, dtype=torch.uint8, device=dev)
            self.sets.append((b, e, c, L.StringsOut(b.data_ptr(), e.data_ptr(), c.data_ptr(), cap, 0)))
        self.k = 0
        self.stream0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.nb = nb
        self.workload = (f"config 5 chunk: detokenize {rows} x {S} ids (GPT-2-shaped vocabulary, 1 % skipped special ids) per GPU and "
                         f"batch, {nb} distinct batches in rotation, fused VocabDecoder+ByteFallback+FuzeRagged, inputs and outputs in HBM "
                         f"({self.n_out[0]} output bytes < 2^31)")
        self.sample_rows = min(rows, 1024)
        self.vocab = V

    def units(self, i):
        return self.rows * self.S

    def _take(self):
        s = self.sets[self.k % 4]
        self.k += 1
        return s

    def step(self, i):
        o_begins, o_ends, o_chars, o = self._take()
        L.check(self.lib, self.lib.ovtk_detokenize_run(self.dec._h, C.c_void_p(self.d_ids[i % self.nb].data_ptr()), C.c_int64(self.rows),
                                                       C.c_int64(self.S), None, C.c_int64(0), 1, C.byref(o), L.MEM_DEVICE, self.stream0))
        return o_begins, o_ends, o_chars[: o.n_chars]

    def enqueue(self, i, st=None):
        o_begins, o_ends, o_chars, o = self._take()
        pending = C.c_void_p()
        L.check(self.lib, self.lib.ovtk_detokenize_enqueue(self.dec._h, C.c_void_p(self.d_ids[i % self.nb].data_ptr()), C.c_int64(self.rows),
                                                           C.c_int64(self.S), None, C.c_int64(0), 1, C.byref(o), st or self.stream0,
                                                           C.byref(pending)))

        def finish():
            L.check(self.lib, self.lib.ovtk_detokenize_finish(pending, C.byref(o)))
            return o_begins, o_ends, o_chars[: o.n_chars]
        return finish

    def algo(self):
        return float(np.mean([4 * self.rows * self.S + n + 8 * self.rows for n in self.n_out]))

    def mean_out(self):
        return float(np.mean(self.n_out))

    def cpu_sample(self, n_s):
        from oracle import oracle as O
        ids, tok, pad = self.ids_host, self.tok, self.pad

        def chain(n):
            r = O.vocab_decoder(ids[:n], tok.vocab, [pad])
            bf = O.byte_fallback(*r[2:5])
            fz = O.fuze(r[0], r[1], bf[0], bf[1])
            return fz[0], fz[1], bf[2]
        ref = chain(16)
        t1 = time.perf_counter()
        chain(n_s)
        return ref, time.perf_counter() - t1, n_s * self.S, "VocabDecoder + ByteFallback + FuzeRagged restatement"


class DetokenizeFull:
    """Config 5 as BASELINE states it: ONE batch of `rows` x 2048 ids (1 048 576 rows = 2.1 G ids = 8.6 GB, 9-10 GB of text out),
    which the reference's int32 char offsets cannot hold in one call (src/vocab_decoder.cpp:62-80).  A step = the whole batch
    through FusedDetokenizer.evaluate_chunked: row chunks below 2^31 output bytes each, pipelined over three HIP streams."""
    unit, dtype, metric, dominant_hint = "Mtok/s", "int32/u8", "token ids/s detokenized (seq 2048)", "detokenize"

    def __init__(self, args, lib, dev, rank):
        from openvino_tokenizers_amd.ops import FusedDetokenizer
        self.lib, self.dev = lib, dev
        tok = BpeTok.load(args.tokenizer)
        self.tok = tok
        self.rows, self.S = args.rows, 2048
        V = len(tok.vocab)
        self.pad = pad = V - 1
        g = torch.Generator(device=dev)
        g.manual_seed(3000 + 100 * rank)
        self.ids = torch.empty((self.rows, self.S), dtype=torch.int32, device=dev)
        slab = 32768
        for a in range(0, self.rows, slab):   # in slabs: the int64 draw and the mask of a slab are 0.8 GB of temporaries
            b = min(a + slab, self.rows)
            part = torch.randint(0, V - 1, (b - a, self.S), generator=g, device=dev).to(torch.int32)
            part[torch.rand((b - a, self.S), generator=g, device=dev) < 0.01] = pad
            self.ids[a:b] = part
            del part
        self.vconst = list(pack_strings(tok.vocab))
        self.dec = VocabDecoder(skip_tokens=[pad], device=dev.index, lib=lib)
        self.fused = FusedDetokenizer(self.dec, byte_fallback=True)
        self.n_out, self.chunks = [], []
        self.n_streams = 3
        self.sample_rows = 1024
        self.vocab = V
        self.first = None
        self.step(0)   # (sizes the estimate; the caching allocator keeps the chunk buffers for the timed passes)
        torch.cuda.synchronize()
        self.workload = (f"config 5: detokenize ONE batch of {self.rows} x {self.S} ids (GPT-2-shaped vocabulary, 1 % skipped special ids) "
                         f"per GPU and step = {self.rows * self.S} ids in, {self.n_out[0]} bytes of text out, in {self.chunks[0]} row "
                         f"chunks below 2^31 output bytes each (FusedDetokenizer.evaluate_chunked: ovtk_detokenize_enqueue / _finish, "
                         f"three HIP streams, two chunks ahead), fused VocabDecoder+ByteFallback+FuzeRagged, inputs and outputs in HBM")

    def units(self, i):
        return self.rows * self.S

    def step(self, i):
        total, first = [0], []

        def sink(a, b, cb, ce, cc):
            total[0] += int(cc.numel())
            if a == 0:
                first.append((cb, ce, cc))
        n = self.fused.evaluate_chunked([self.ids] + self.vconst, sink=sink, streams=self.n_streams, depth=2 if self.n_streams > 1 else 0)
        self.n_out.append(total[0])
        self.chunks.append(n)
        self.first = first[0]
        return self.first

    def algo(self):
        return float(np.mean([4 * self.rows * self.S + n + 8 * self.rows for n in self.n_out]))

    def mean_out(self):
        return float(np.mean(self.n_out))

    def _chain(self, ids):
        from oracle import oracle as O
        r = O.vocab_decoder(ids, self.tok.vocab, [self.pad])
        bf = O.byte_fallback(*r[2:5])
        fz = O.fuze(r[0], r[1], bf[0], bf[1])
        return fz[0], fz[1], bf[2]

    def cpu_sample(self, n_s):
        ids = self.ids[:n_s].cpu().numpy()
        ref = self._chain(ids[:16])
        t1 = time.perf_counter()
        self._chain(ids)
        return ref, time.perf_counter() - t1, n_s * self.S, "VocabDecoder + ByteFallback + FuzeRagged restatement"

    def parity_per_chunk(self, n_rows=8):
        """The first rows of EVERY chunk against the oracle chain (one more, untimed pass)."""
        ok, seen = [], []

        def sink(a, b, cb, ce, cc):
            n = min(n_rows, b - a)
            rb, re_, rc = self._chain(self.ids[a:a + n].cpu().numpy())
            e = ce[:n].cpu().numpy()
            ok.append(bool(np.array_equal(cb[:n].cpu().numpy(), rb) and np.array_equal(e, re_) and
                           np.array_equal(cc[: int(e[-1])].cpu().numpy(), rc[: int(re_[-1])])))
            seen.append((a, b, int(cc.numel())))
        self.fused.evaluate_chunked([self.ids] + self.vconst, sink=sink)
        return {"chunks": len(ok), "all_bit_exact": bool(all(ok)), "rows_checked_per_chunk": n_rows,
                "chunk_rows_and_bytes": seen, "max_chunk_bytes": max(x[2] for x in seen), "limit": (1 << 31) - 2}


class RaggedToDenseBench:
    """a7 at config-2 size: the ragged ids of a config-2 batch -> input_ids [rows, T] + attention mask, T = longest row."""
    unit, dtype, metric, dominant_hint = "GB/s", "int32/u8", "algorithmic GB/s (RaggedToDense, config-2 ids)", "ragged_to_dense"

    def __init__(self, args, lib, dev, rank):
        self.lib, self.dev = lib, dev
        enc = BpeEncode(args, lib, dev, rank, args.tokenizer, "zipf", args.rows, args.bytes, 1000)
        self.rows = rows = args.rows
        self.inputs, self.T = [], 0
        for k in range(enc.batches.n):
            b, e, ids = enc.step(k)
            torch.cuda.synchronize()
            self.inputs.append((b.clone(), e.clone(), ids.clone()))
            self.T = max(self.T, int((e - b).max()))
        del enc
        self.nb = len(self.inputs)
        self.outs = [(torch.empty((rows, self.T), dtype=torch.int32, device=dev), torch.empty((rows, self.T), dtype=torch.uint8, device=dev))
                     for _ in range(self.nb)]
        self.copied = [int(torch.clamp(e - b, max=self.T).sum()) for b, e, _ in self.inputs]
        self.pad = np.asarray([50256], np.int32)
        self.stream0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.workload = (f"RaggedToDense (a7): {rows} rows of config-2 token ids -> [rows, {self.T}] i32 + mask u8, right padding, "
                         f"{self.nb} distinct input / output sets in rotation, inputs and outputs in HBM")
        self.vocab = 0
        self.sample_rows = 4096

    def units(self, i):  # algorithmic bytes: 8 B + 4 sum(min(len, T)) + 5 B T  
        k = i % self.nb
        return 8 * self.rows + 4 * self.copied[k] + 5 * self.rows * self.T

    def step(self, i):
        k = i % self.nb
        b, e, ids = self.inputs[k]
        dense, mask = self.outs[k]
        L.check(self.lib, self.lib.ovtk_ragged_to_dense(C.c_void_p(b.data_ptr()), C.c_void_p(e.data_ptr()), C.c_int64(self.rows),
                                                        C.c_void_p(ids.data_ptr()), C.c_int64(ids.numel()), 4, C.c_int64(1), C.c_int32(self.T),
                                                        C.c_void_p(self.pad.ctypes.data), 1, 0, C.c_void_p(dense.data_ptr()),
                                                        C.c_void_p(mask.data_ptr()), L.MEM_DEVICE, self.dev.index, self.stream0))
        return dense, mask

    def algo(self):
        return float(np.mean([self.units(k) for k in range(self.nb)]))

    def mean_out(self):
        return self.rows * self.T

    def cpu_sample(self, n_s):
        from oracle import oracle as O
        b, e, ids = (x.cpu().numpy() for x in self.inputs[0])
        ref = O.ragged_to_dense(b[:n_s], e[:n_s], ids, self.T, 50256)
        t1 = time.perf_counter()
        O.ragged_to_dense(b[:n_s], e[:n_s], ids, self.T, 50256)
        return ref, time.perf_counter() - t1, 8 * n_s + 4 * int(np.minimum(e[:n_s] - b[:n_s], self.T).sum()) + 5 * n_s * self.T, \
            "ragged_to_dense restatement"


class VocabEncoderBench:
    """a6: VocabEncoder on the words of config-3 text (~3 M words per batch), keys = the BERT-shaped vocabulary."""
    unit, dtype, metric, dominant_hint = "Mwords/s", "u8/int32", "words/s looked up (VocabEncoder, BERT-shaped vocabulary)", "vocab_encoder"

    def __init__(self, args, lib, dev, rank):
        self.lib, self.dev = lib, dev
        tok = load_tokenizer("bert")
        self.tok = tok
        model = TextModel(1234, "zipf")
        tb = TextBatches(model, args.rows, 256, [2000 + 7 * j for j in range(min(args.batches, 4))], dev, lower=True)
        ws = RegexSplit("remove", device=dev.index, lib=lib)
        self.words = []
        for k in range(tb.n):
            out = ws.evaluate(tb.d[k] + [np.frombuffer(BERT_WS.encode(), np.uint8)])
            torch.cuda.synchronize()
            self.words.append((out[2].clone(), out[3].clone(), tb.d[k][4]))
        self.host0 = tb.host[0]
        self.nb = len(self.words)
        keys = list(pack_strings(tok["vocab"]))
        self.values = np.arange(len(tok["vocab"]), dtype=np.int32)
        self.enc = VocabEncoder(device=dev.index, lib=lib)
        self.enc._ensure(list(self.words[0]) + keys + [self.values, np.int32(-1)])
        self.keys = keys
        self.outs = [torch.empty(w[0].numel(), dtype=torch.int32, device=dev) for w in self.words]
        self.dflt = np.asarray([-1], np.int32)
        self.stream0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.workload = (f"VocabEncoder (a6): {self.words[0][0].numel()} words per batch (config-3 text split at white space), "
                         f"V={len(tok['vocab'])} keys, i32 values, {self.nb} distinct batches in rotation, inputs and outputs in HBM")
        self.vocab = len(tok["vocab"])
        self.sample_rows = 4096

    def units(self, i):
        return self.words[i % self.nb][0].numel()

    def step(self, i):
        k = i % self.nb
        b, e, c = self.words[k]
        s = L.Strings(b.data_ptr(), e.data_ptr(), c.data_ptr(), b.numel(), c.numel())
        L.check(self.lib, self.lib.ovtk_vocab_encoder_run(self.enc._h, C.byref(s), C.c_void_p(self.dflt.ctypes.data),
                                                          C.c_void_p(self.outs[k].data_ptr()), L.MEM_DEVICE, self.stream0))
        return (self.outs[k],)

    def algo(self):  # N_c(words) + 8 N_e + 4 N_e  
        return float(np.mean([int((e - b).sum()) + 12 * b.numel() for b, e, _ in self.words]))

    def mean_out(self):
        return float(np.mean([w[0].numel() for w in self.words]))

    def cpu_sample(self, n_s):
        from oracle import oracle as O
        b, e, c = (x.cpu().numpy() for x in self.words[0])
        n = min(len(b), n_s * 50)
        enc = O.VocabEncoder(self.tok["vocab"], self.values)
        ref = enc(b[:n], e[:n], c, -1)
        t1 = time.perf_counter()
        enc(b[:n], e[:n], c, -1)
        return (ref,), time.perf_counter() - t1, n, "VocabEncoder restatement (std::unordered_map<string,T>)"


class SmallBatchLatency:
    """Config 1: 32 x ~128-byte ASCII strings, one BLOCKING ovtk_encode_run per step -- the latency a single evaluate()
    of the converted GPT-2 tokenizer sees (device-resident buffers)."""
    unit, dtype, metric, dominant_hint = "us/call", "u8/int32", "latency per blocking encode call (32 x 128-byte strings)", "lookup_fused"
    higher_is_better = False

    def __init__(self, args, lib, dev, rank):
        args2 = argparse.Namespace(**vars(args))
        args2.batches = 8
        self.enc = BpeEncode(args2, lib, dev, rank, args.tokenizer, "zipf", 32, 128, 1234, n_batches=8)
        self.workload = ("config 1: GPT-2-shaped byte-level BPE, 32 x ~128-byte zipf strings, one blocking ovtk_encode_run per step "
                         "(device-resident buffers), 8 distinct batches in rotation")
        self.vocab = self.enc.vocab
        self.sample_rows = 32
        self.batches = self.enc.batches
        self.cpu_chain = self.enc.cpu_chain

    def units(self, i):
        return 1

    def step(self, i):
        return self.enc.step(i)

    def algo(self):
        return self.enc.algo()

    def mean_out(self):
        return self.enc.mean_out()


# ---------------------------------------------------------------------------------------------- measurement helpers
def bind_to_gpu_numa_node(dev):
    """Runs this process on the CPUs of the NUMA node the GPU hangs off (2-socket hosts: pinned buffers allocated from
    the other socket cross the inter-socket link on every PCIe transfer, and launches pay for it too).  Returns
    (original affinity, description)."""
    try:
        before = os.sched_getaffinity(0)
        p = torch.cuda.get_device_properties(dev)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        cpus = Path(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read_text().strip()
        want = set()
        for part in cpus.split(","):
            lo, _, hi = part.partition("-")
            want.update(range(int(lo), int(hi or lo) + 1))
        want &= before
        if want:
            os.sched_setaffinity(0, want)
            return before, f"bound to the GPU's NUMA node (PCI {bdf}, CPUs {cpus})"
        return before, "GPU NUMA node unknown: not bound"
    except Exception as err:  # noqa: BLE001 -- best effort: containers may hide sysfs
        return None, f"not bound ({type(err).__name__})"


def profile_table(lib):
    buf = C.create_string_buffer(16384)
    lib.ovtk_profile_dump(buf, 16384)
    return {ln.split()[0]: (float(ln.split()[1]), int(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.strip()}


def run_pipelined(wl, steps, stream_ptrs, depth, first=0, complete=lambda f: f()):
    """The host loop: launch batch k (streams in turn), then complete batch k - depth."""
    inflight = []
    for i in range(first, first + steps):
        inflight.append(wl.enqueue(i, stream_ptrs[i % len(stream_ptrs)]))
        if len(inflight) > depth:
            complete(inflight.pop(0))
    while inflight:
        complete(inflight.pop(0))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def cpu_one_core(wl, n_s):
    """~10 s of the oracle ("port" of the reference's algorithm) on one core over the first rows of batch 0."""
    if hasattr(wl, "cpu_sample"):
        ref, cdt, units, what = wl.cpu_sample(n_s)
        passes = 1
        while cdt < 8.0 and passes < 64:
            _, more, u, _ = wl.cpu_sample(n_s)
            cdt, units, passes = cdt + more, units + u, passes + 1
        return ref, cdt, units, what, passes
    chain, what = wl.cpu_chain()
    rb, re_, b, e, c = wl.batches.host[0]
    ref = chain(rb[:1024], re_[:1024], b[:1024], e[:1024], c)  # parity prefix + warms the piece cache
    cdt, units, passes = 0.0, 0, 0
    while cdt < 8.0 and passes < 32:
        t1 = time.perf_counter()
        chain(rb[:n_s], re_[:n_s], b[:n_s], e[:n_s], c)
        cdt += time.perf_counter() - t1
        units += int(e[n_s - 1] - b[0])
        passes += 1
    return ref, cdt, units, what, passes


def cpu_all_cores(wl, seconds=8.0, threads=None, private=False):
    """`threads` host threads (default: every host core), each on a contiguous row shard of batch 0.
    private=False: ONE tokenizer object shared by all threads (its piece cache behind a shared_mutex as in the reference): what
    OpenVINO's THROUGHPUT streams do with one compiled tokenizer (benchmark/benchmark.py:301-302).
    private=True: a tokenizer -- tables and piece cache -- of its own per thread: what the same cores give when nothing is
    shared (N independent processes).  The oracle calls release the GIL (ctypes)."""
    if not hasattr(wl, "cpu_chain"):
        return None
    chain, what = wl.cpu_chain()
    rb, re_, b, e, c = wl.batches.host[0]
    threads = max(1, threads or os.cpu_count() or 1)
    rows = len(rb)
    per = max(1, min(rows // threads, 4096))   # (a shard of a few thousand rows per pass: the short legs still complete passes)
    threads = min(threads, rows // per)
    chain(rb[:1024], re_[:1024], b[:1024], e[:1024], c)  # warm cache
    done = [0] * threads
    ready = threading.Barrier(threads + 1)
    t_stop = [0.0]

    def work(t):
        # the thread's shard as a tensor of its own (rebased offsets, its slice of the chars): the op sizes its outputs by
        # the chars tensor it is given (regex_split.cpp:182, bpe_tokenizer.cpp:135)
        lo, hi = t * per, (t + 1) * per
        c0, c1 = int(b[lo]), int(e[hi - 1])
        rbt = (rb[lo:hi] - rb[lo]).astype(np.int32)
        ret = (re_[lo:hi] - rb[lo]).astype(np.int32)
        bt, et, ct = (b[lo:hi] - c0).astype(np.int32), (e[lo:hi] - c0).astype(np.int32), c[c0:c1].copy()
        nbytes = c1 - c0
        mine = chain
        if private:
            mine, _ = wl.cpu_chain()
            mine(rbt[:256], ret[:256], bt[:256], et[:256], ct)   # its own cache, warm
        ready.wait()
        while time.perf_counter() < t_stop[0]:
            mine(rbt, ret, bt, et, ct)
            done[t] += nbytes

    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    t_stop[0] = time.perf_counter() + seconds + 3600.0
    ready.wait()   # every thread has its shard (and, private, its tokenizer)
    t0 = time.perf_counter()
    t_stop[0] = t0 + seconds
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    how = ("a tokenizer and piece cache of its own per thread (nothing shared)" if private else
           "ONE shared tokenizer as under OpenVINO's THROUGHPUT streams (the piece cache sits behind a std::shared_mutex taken per piece, "
           "bpe_tokenizer.cpp:197-205: the reader lock's cache line is what the threads contend for)")
    return {"value": round(sum(done) / dt / 1e6, 2), "unit": "MB/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads x {per} rows of batch 0 each, {how}, repeated for {dt:.1f} s ({sum(done)} bytes in all), {what}",
            "host_cpus": os.cpu_count()}


def cpu_thread_curve(wl):
    """The shared-tokenizer rate at 1 / 8 / 32 / 64 / all threads (3 s each): whether the flat all-cores figure is the lock or
    the cores."""
    if not hasattr(wl, "cpu_chain"):
        return None
    n = os.cpu_count() or 1
    out = {}
    for t in sorted({1, 8, 32, 64, n}):
        if t <= n:
            r = cpu_all_cores(wl, seconds=3.0, threads=t)
            out[str(r["cores"])] = r["value"]
    return {"unit": "MB/s", "shared_tokenizer_by_threads": out}


def end_to_end_leg(wl, lib, dev, n_streams=4, steps=64, depth=3):
    """Host buffers in, host buffers out -- what a CPU-plugin evaluate() holds.  In: the batch as ONE packed u8 string tensor
    ([i32 n][i32 begin_0][i32 end_i x n][bytes], src/utils.cpp:18-29) in pinned memory, handed to ovtk_encode_enqueue_packed
    (StringTensorUnpack -> RegexSplit -> BPETokenizer; one H2D copy per batch, the decomposed tensors are views of the device
    copy).  Out: pinned begins / ends / ids written by the kernels themselves.  Batches alternate between `n_streams` HIP
    streams, `depth` ahead, so the copy of batch k+1 runs under the kernels and PCIe stores of batch k."""
    if not isinstance(wl, BpeEncode):
        return None
    tb = wl.batches
    nb = min(tb.n, 4)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    ptrs = [C.c_void_p(x.cuda_stream) for x in streams]
    packed = []
    for k in range(nb):
        b, e, c = (x.cpu().numpy() for x in tb.d[k][2:5])
        head = np.concatenate([[tb.rows, int(b[0])], e]).astype(np.int32).view(np.uint8)
        buf = torch.empty(len(head) + len(c), dtype=torch.uint8, pin_memory=True)
        v = buf.numpy()
        v[: len(head)] = head
        v[len(head):] = c      # (the generator's strings are back to back from 0: the packed form's layout)
        packed.append(buf)
    outs = []
    for _ in range(depth + 2):
        b = torch.empty(tb.rows, dtype=torch.int32, pin_memory=True)
        e = torch.empty(tb.rows, dtype=torch.int32, pin_memory=True)
        ids = torch.empty(tb.cap, dtype=torch.int32, pin_memory=True)
        outs.append((b, e, ids, L.RaggedI32Out(b.data_ptr(), e.data_ptr(), ids.data_ptr(), tb.cap, 0, 0)))
    moved_in, moved_out = [0], [0]
    stamps = []

    def loop(n):
        inflight = []
        for i in range(n):
            o = outs[i % len(outs)]
            pk = packed[i % nb]
            pending = C.c_void_p()
            L.check(lib, lib.ovtk_encode_enqueue_packed(wl.split._h, wl.bpe._h, C.c_void_p(pk.data_ptr()), C.c_int64(pk.numel()), C.byref(o[3]),
                                                        L.MEM_HOST, ptrs[i % n_streams], C.byref(pending)))
            inflight.append((pending, o, i % nb))
            if len(inflight) > depth:
                p, oo, k = inflight.pop(0)
                L.check(lib, lib.ovtk_encode_finish(p, C.byref(oo[3])))
                stamps.append(time.perf_counter())
                moved_in[0] += packed[k].numel()
                moved_out[0] += 8 * tb.rows + 4 * int(oo[3].n_data)
        for p, oo, k in inflight:
            L.check(lib, lib.ovtk_encode_finish(p, C.byref(oo[3])))
            moved_in[0] += packed[k].numel()
            moved_out[0] += 8 * tb.rows + 4 * int(oo[3].n_data)
    loop(48)   # the first ~30 batches on fresh streams run ahead of the steady state (tools/e2e_age_probe.py): not timed
    moved_in[0] = moved_out[0] = 0
    stamps.clear()
    dt_all = timed(lambda: loop(steps + 8))
    # the pipeline's rate: from the return of one finish() to the return of the next, without the 8 calls it takes to fill
    dt = stamps[-1] - stamps[7]
    units = sum(tb.n_chars[i % nb] for i in range(8, 8 + (len(stamps) - 8)))
    steps = len(stamps) - 8
    # the result is the same as the device-resident path's
    o = outs[(steps + 8 - 1) % len(outs)]
    k_last = (steps + 8 - 1) % nb
    same = None
    if k_last in wl.n_out:
        same = bool(int(o[3].n_data) == wl.n_out[k_last])
    return {"value": round(units / dt / 1e6, 1), "unit": "MB/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "h2d_GBps": round(moved_in[0] / dt_all / 1e9, 2), "d2h_GBps": round(moved_out[0] / dt_all / 1e9, 2),
            "pcie_GBps_both_directions": round((moved_in[0] + moved_out[0]) / dt_all / 1e9, 2),
            "note": f"ONE pinned packed u8 string tensor in per batch (ovtk_encode_enqueue_packed: the f2 wire form, one H2D copy), pinned "
                    f"begins / ends / ids out written by the kernels themselves (no D2H copy), {n_streams} streams, {depth} batches ahead; "
                    f"{steps} steps over {nb} batches after 48 untimed ones, in a process of its own (fresh HIP streams); never reported as `value`",
            "ids_last_batch": int(o[3].n_data), "same_id_count_as_device_path": same}


def end_to_end_in_child(args):
    """The end_to_end leg in a process of its own: every stream a process has ever created keeps its hardware queue, and the
    streams of the legs before this one would share queues with the leg's own (0.78 ms per batch alone, 0.92 ms
    behind a dozen older streams)."""
    import subprocess
    cmd = [sys.executable, str(Path(__file__).resolve()), "--e2e-child", "--rows", str(args.rows), "--bytes", str(args.bytes), "--tokenizer",
           args.tokenizer, "--text", args.text, "--batches", "4"] + (["--lib", args.lib] if args.lib else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": (p.stderr or p.stdout)[-500:]}
    return json.loads(lines[-1])


def relaunch_argv(n_gpus, argv, port=None):
    """The command `bench.py --gpus N` replaces itself with when nobody launched it as N ranks (WORLD_SIZE unset): one
    process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    port = port or int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="2", choices=["1", "2", "3", "4", "5", "r2d", "vocab_encoder", "pipeline"])
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--bytes", type=int, default=512)
    ap.add_argument("--batches", type=int, default=8, help="distinct input batches rotated through the timed loop")
    ap.add_argument("--text", default="zipf", choices=["zipf", "uniform", "mixed"])
    ap.add_argument("--tokenizer", default="gpt2")
    ap.add_argument("--pattern", default=None, choices=["qwen2", "cl100k", "o200k", "deepseek-v3"],
                    help="configs 2 / 4: run the tokenizer's tables behind this model's split pattern instead of its own")
    ap.add_argument("--no-memo", action="store_true", help="config 2: BPETokenizer with cache_capacity=0 (no piece memo)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the all-gather (rank-local consumer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stress / end_to_end / all-cores legs behind the timed region")
    ap.add_argument("--hog", type=int, default=0, help="debug: occupy CU slots with N idle 512-thread blocks on a side stream during "
                                                        "every step (stands in for RCCL's all-gather kernel; tools/cu_hog.hip)")
    ap.add_argument("--row-tickets", type=int, default=-1, help="ovtk_set_row_tickets(n); default 0 (static row assignment)")
    ap.add_argument("--short-path", type=int, default=1, choices=[0, 1, 2],
                    help="ovtk_set_short_path(n): 1 (the library's default) = span -> compact where a handle's last calls say every piece is in its tables, "
                         "0 = always span -> left-over rows -> merge -> compact (round 5's four launches), 2 = every call tries")
    ap.add_argument("--hog-lds", type=int, default=0, help="debug: dynamic LDS bytes per hog block")
    ap.add_argument("--lib", default=None, help="debug: load this build of libovtk_amd.so")
    ap.add_argument("--no-alone-leg", action="store_true", help="skip the one-stream leg behind `roofline` (profile runs of the "
                    "overlapped loop: rocprofv3's per-kernel average then covers the overlapped launches only)")
    ap.add_argument("--depth", type=int, default=3, help="batches launched ahead of the one being completed (two-half calls); 3 since round 6: the same rate over 200 steps as 2, a shorter way into the pipeline for the first batches of a 20-step window (profiles/r06/x_depth_ab.txt)")
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "p2p"],
                    help="N > 1: one RCCL all-gather of the wires, or grouped direct sends / receives (one xGMI link per pair)")
    ap.add_argument("--wire", type=int, default=1,
                    help="N > 1, fused BPE: 1 = the encode writes the exchange's send wire itself (compact_kernel narrows the ids; no "
                         "ragged int32 ids, no pack kernel), 0 = encode to ragged ids, then ovtk_shard_pack")
    ap.add_argument("--exchange-stream", type=int, default=1, help="1: pack/unpack of the exchange on a HIP stream of their own")
    ap.add_argument("--pipeline-calls", type=int, default=1, choices=[1, 2], help="--config pipeline: 1 = ovtk_encode_dense_* (the whole graph in one call), 2 = encode + ovtk_encode_tail_run")
    ap.add_argument("--streams", type=int, default=3, help="consecutive batches alternate between this many HIP streams (two-half calls)")
    ap.add_argument("--sync", action="store_true", help="one blocking ovtk_encode_run per step (no launch/complete overlap)")
    ap.add_argument("--force-exchange", action="store_true", help="N = 1: still run the exchange, in a one-rank RCCL group (debug)")
    ap.add_argument("--memo-learn", type=int, default=None, help="config 2 / 4: ovtk_bpe_params::memo_learn (0 = the library's default, -1 = exactly cache_capacity)")
    ap.add_argument("--cache-capacity", type=int, default=None, help="config 2: the BPETokenizer attribute (debug; default: the converter's 20000)")
    ap.add_argument("--memo-store", type=int, default=-1, help="ovtk_set_memo_store(n) before the tokenizer is built (debug; default: the library's)")
    ap.add_argument("--e2e-child", action="store_true", help="internal: run only the end_to_end leg and print its JSON object")
    ap.add_argument("--spawn", action="store_true", help="take the `--gpus N` relaunch under torch.distributed.run even for N = 1 (test)")
    args = ap.parse_args()
    os.environ["OVTK_BENCH_DEPTH"] = str(args.depth)   # the output rings are sized for the batches in flight

    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:   # `python bench.py --gpus N`: become N ranks
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s)")
        os.execv(sys.executable, relaunch_argv(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} was launched as {world} rank(s) (WORLD_SIZE): the two must agree")
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

    affinity0, numa_note = bind_to_gpu_numa_node(dev)
    lib = L.load(args.lib)
    if args.memo_store >= 0:
        L.check(lib, lib.ovtk_set_memo_store(C.c_int64(args.memo_store)))
    if args.e2e_child:
        wl = make_workload(args, lib, dev, rank)
        for k in range(wl.batches.n):   # the id counts of the device-resident path (and the memo learns what it will know)
            wl.step(k)
        torch.cuda.synchronize()
        print(json.dumps(end_to_end_leg(wl, lib, dev)))
        return
    special = {"5": Detokenize, "r2d": RaggedToDenseBench, "vocab_encoder": VocabEncoderBench, "1": SmallBatchLatency}
    if args.config == "5" and args.rows > 131072:   # more than one int32-offset call can hold: the chunk loop
        special["5"] = DetokenizeFull
    wl = special[args.config](args, lib, dev, rank) if args.config in special else make_workload(args, lib, dev, rank)
    is_encode = isinstance(wl, EncodeWorkload)
    two_half = hasattr(wl, "enqueue") and not args.sync
    # N > 1: every rank's ragged ids are all-gathered (RCCL), one batch behind the encode so that the gather of batch k
    # travels over xGMI while batch k + 1 is encoded; flush() completes the last one inside the timed region.
    exchange = None
    if dist_on and not args.no_gather and is_encode:
        xstream = torch.cuda.Stream(dev) if args.exchange_stream else None
        # equal shards of the same text model: the ranks' id counts differ by well under 0.1 %, so 3 % of padding on the
        # wire (instead of the class's default 12.5 %) never triggers a re-gather and the gather moves 8 % fewer bytes
        exchange = ShardExchange(wl.batches.rows * world, wl.vocab, dev, lib=lib, stream=xstream, headroom=1.03, transport=args.exchange)

    # A step = one batch through the hot path.  Where the op has the two-half form the host launches batch k, then
    # completes batch k-depth (status check, and with N > 1 its exchange) while the GPU works on k: the reference's
    # evaluate() semantics per batch, without the GPU idling while the host reads a status word.  Consecutive batches go to
    # alternating HIP streams: the latency-bound merge kernel of one batch then shares the CUs with the issue-bound
    # lookup kernel of the next (per-launch durations grow, the step shrinks).  --sync: one blocking call per step.
    row_tickets = max(args.row_tickets, 0)
    L.check(lib, lib.ovtk_set_row_tickets(row_tickets))
    L.check(lib, lib.ovtk_set_short_path(args.short_path))
    side_streams = [torch.cuda.Stream(dev) for _ in range(max(args.streams - 1, 0))]
    stream_ptrs = [C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)] + [C.c_void_p(x.cuda_stream) for x in side_streams]
    hog = None
    if args.hog:
        hog_lib = C.CDLL(str(ROOT / "tools" / "build" / "libcuhog.so"))
        hog_stream = torch.cuda.Stream(dev)
        hog = lambda: hog_lib.cu_hog(args.hog, 512, args.hog_lds, C.c_double(300.0), C.c_void_p(hog_stream.cuda_stream))  # noqa: E731

    latency_cfg = isinstance(wl, SmallBatchLatency)
    to_wire = exchange is not None and args.wire and hasattr(wl, "enqueue_wire") and two_half
    if to_wire:
        # the pad the wires are sized with: the largest shard of the rotation (one untimed pass), agreed over the ranks
        for k in range(wl.batches.n):
            wl.step(k)
        exchange.agree_pad(max(wl.n_out.values()))

    def complete(finish):
        res = finish()
        if exchange is None:
            return res
        return exchange.submit_wire(*res) if to_wire else exchange.submit(*res)

    def run_steps(first, n, stamps=None):
        def done(f):
            complete(f)
            if stamps is not None:
                stamps.append(time.perf_counter())
        if two_half:
            inflight = []
            for i in range(first, first + n):
                if hog is not None:
                    hog()
                st = stream_ptrs[i % len(stream_ptrs)]
                inflight.append(wl.enqueue_wire(i, st, exchange.lease_wire(), exchange) if to_wire else wl.enqueue(i, st))
                if len(inflight) > args.depth:
                    done(inflight.pop(0))
            while inflight:
                done(inflight.pop(0))
        else:
            for i in range(first, first + n):
                done(lambda i=i: wl.step(i))
        if exchange is not None:
            exchange.flush()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Priming, before the W warm-up steps: every DISTINCT batch of the rotation once.  The piece memo and the piece store are the
    # reference's piece cache (bpe_tokenizer.cpp:197-205,331-338) -- they fill during the first pass over new text, as its cache
    # does, and the metric is the rate of a server in its steady state; with --warmup smaller than the rotation the timed region
    # would otherwise hold first passes (20 steps after 5: 0.157 ms per step against 0.139 after 16).  Said in config.priming.
    if hasattr(wl, "batches") and not latency_cfg:
        run_steps(0, wl.batches.n)
    run_steps(0, args.warmup)
    lib.ovtk_profile_enable(0)   # the timed region runs without the library's per-kernel event brackets
    barrier()
    timed_stamps = [] if os.environ.get("OVTK_BENCH_STAMPS") else None   # (a diagnostic: when each of the K batches completed, to stderr)
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, timed_stamps)   # every one of the K batches is complete (and, N > 1, gathered on every rank) when it returns
    barrier()
    dt = time.perf_counter() - t0
    if timed_stamps:
        print("timed region: completions at (us)", [round((x - t0) * 1e6) for x in timed_stamps], "end", round(dt * 1e6), file=sys.stderr)
    # The same loop for 200 further steps: the K steps above are the reported region -- 1.3 ms of wall time at K = 20,
    # of which one fill of the three-stream pipeline is a visible part --, this leg says what the loop settles at, and how evenly the
    # batches complete (the host-side interval between two completions: its median, and the 90th percentile).
    steady = None
    if not latency_cfg and world == 1:
        n_steady = 200
        stamps = []
        barrier()
        ts0 = time.perf_counter()
        run_steps(args.warmup + args.steps, n_steady, stamps)
        barrier()
        dts = time.perf_counter() - ts0
        su = sum(wl.units(i) for i in range(args.warmup + args.steps, args.warmup + args.steps + n_steady))
        gaps = np.diff(np.asarray(stamps)) * 1e3
        steady = {"steps": n_steady, "ms_per_step": round(dts / n_steady * 1e3, 4),
                  "value": round(su / dts / (1e9 if wl.unit == "GB/s" else 1e6), 1), "unit": wl.unit,
                  "completion_interval_ms": {"p50": round(float(np.median(gaps)), 4), "p90": round(float(np.percentile(gaps, 90)), 4)},
                  "note": "the same host loop for 200 further steps behind the reported region; not `value`"}
    # per-kernel wall times of the same loop (overlapped launches), collected in a leg of its own behind the timed region
    n_prof = min(args.steps, 48)
    lib.ovtk_profile_reset()
    lib.ovtk_profile_enable(1)
    run_steps(args.warmup + args.steps, n_prof)
    barrier()
    lib.ovtk_profile_enable(0)
    my_units = sum(wl.units(i) for i in range(args.warmup, args.warmup + args.steps))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nb = torch.tensor([my_units], dtype=torch.int64, device=dev)
        dist.all_reduce(nb)
        total_units = int(nb.item())
    else:
        total_units = my_units

    ms_per_step = dt / args.steps * 1e3
    latency = isinstance(wl, SmallBatchLatency)
    value = ms_per_step * 1e3 if latency else total_units / dt / (1e9 if wl.unit == "GB/s" else 1e6)

    # ---- roofline of the dominant kernel: HIP events recorded by the library on the launch stream.
    # `roofline.achieved / frac` come from a ONE-STREAM leg run right here (each kernel with the chip to itself: the duration
    # rocprofv3 --kernel-trace --stats reproduces for `bench.py --streams 1`, profiles/); the timed loop's own events are of
    # launches that share the CUs with the neighbouring batches' kernels (the point of the streams) and are reported as
    # `overlapped` for information only.
    prof = profile_table(lib)
    kernels = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}
    per_step = {k: v[0] / n_prof for k, v in prof.items()}
    roofline = None
    if per_step:
        alone = {}
        if not args.no_alone_leg:
            lib.ovtk_profile_reset()
            lib.ovtk_profile_enable(1)
            n_leg = 24 if not isinstance(wl, DetokenizeFull) else 3
            if hasattr(wl, "n_streams"):
                wl.n_streams = 1   # (a workload that pipelines inside a step: its chunks one after the other)
            for i in range(n_leg):
                (wl.enqueue(i, stream_ptrs[0])() if two_half else wl.step(i))
            torch.cuda.synchronize()
            if hasattr(wl, "n_streams"):
                wl.n_streams = 3
            lib.ovtk_profile_enable(0)
            alone = {k: v for k, v in profile_table(lib).items() if v[1]}
        src = alone or prof
        n_src = n_leg if alone else n_prof
        hint = getattr(wl, "dominant_hint", None)
        per_launch = {k: v[0] / v[1] for k, v in src.items()}
        per_step_src = {k: v[0] / n_src for k, v in src.items()}
        dom = max(per_step_src, key=per_step_src.get)
        if not alone and hint in per_step_src and per_step_src[hint] >= 0.8 * per_step_src[dom]:
            dom = hint
        launches_per_step = max(1, round(src[dom][1] / n_src))
        k_ms = per_launch[dom]
        algo_bytes = wl.algo()   # algorithmic bytes of one pass
        achieved = algo_bytes / launches_per_step / (k_ms * 1e-3) / 1e9
        traffic, pmc_cfg = None, None
        if PMC_FILE.exists():  # per-launch FETCH_SIZE (doubled: gfx950 correction) + WRITE_SIZE of this kernel, see profiles/README.md
            pmc_cfg = json.loads(PMC_FILE.read_text()).get(f"config{args.config}", {})
            traffic = pmc_cfg.get(dom)
        roofline = {"bound": "hbm", "kernel": KERNEL_NAMES.get(dom, dom), "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "kernel_ms": round(k_ms, 4), "algorithmic_bytes": round(algo_bytes),
                    "launches_per_step": launches_per_step,
                    "bytes_note": ("algorithmic bytes of the whole pass (SURVEY 8d) over the dominant kernel's own time.  For the encode "
                                   "configurations that kernel reads all the text and stages all ids but those of the deferred pieces, "
                                   "so its own algorithmic bytes are the same figure within ~5 %; `step` prices every kernel of the path.  "
                                   "`traffic` / `step_traffic` = 2 x FETCH_SIZE + WRITE_SIZE (profiles/latest_pmc.json): the guide calibrates the doubling "
                                   "for wide streaming reads only (MI355X_MICROARCH.md, HBM counters), so on the 32-byte random probes of the memo -- "
                                   "two thirds of this kernel's fetches -- it is an upper bound"),
                    "measured": (f"one-stream leg of {n_leg} batches behind the timed region (every kernel alone on the chip)" if alone
                                 else "the timed loop's own launches"),
                    "one_stream_kernel_ms": {k: round(v, 4) for k, v in sorted(per_launch.items())},
                    "one_stream_kernel_sum_ms_per_step": round(sum(per_step_src.values()), 4)}
        step_gbs = algo_bytes / (ms_per_step * 1e-3) / 1e9 * (total_units / max(my_units, 1)) / world
        # counted HBM bytes of EVERY kernel of the step (profiles/latest_pmc.json, a separate --pmc run) over the algorithmic bytes
        step_traffic = sum(v for k, v in (pmc_cfg or {}).items() if k in per_step_src and src[k][1]) if pmc_cfg else None
        roofline["step_traffic"] = step_traffic
        roofline["traffic_ratio"] = round(step_traffic / algo_bytes, 3) if step_traffic else None
        kb = getattr(wl, "kernel_algo", lambda _dom: None)(dom)
        roofline["kernel_bytes"] = round(kb) if kb else None
        roofline["kernel_frac"] = round(kb / launches_per_step / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kb else None
        roofline["step"] = {"achieved": round(step_gbs, 2), "frac": round(step_gbs / HBM_PEAK_GBS, 5),
                            "note": "algorithmic bytes of one batch / ms_per_step, per GPU: every kernel of the path, overlapped as run"}
        if alone:
            roofline["alone"] = {"kernel_ms": round(k_ms, 4), "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5)}
            roofline["overlapped"] = {"kernel_ms": kernels.get(dom), "note": f"wall time of the same kernel's launches in the timed loop "
                                      f"({len(stream_ptrs)} streams: neighbouring batches share the CUs); not used for frac"}

    if rank != 0:
        if exchange is not None:
            exchange.close()
        if dist_on:
            dist.destroy_process_group()
        return

    # ---- parity spot-check + CPU baselines (oracle = "port" of the reference's algorithm) + extra legs, rank 0, N = 1
    cpu_baseline, cpu_all, parity, stress, e2e = None, None, None, None, None
    cpu_all_private, cpu_curve = None, None
    if world == 1 and not args.no_cpu_baseline:
        n_s = wl.sample_rows
        ref, cdt, sample_units, what, passes = cpu_one_core(wl, n_s)
        res = wl.step(0)
        torch.cuda.synchronize()
        if isinstance(wl, (RaggedToDenseBench,)):
            parity = bool(np.array_equal(ref[0], res[0][: len(ref[0])].cpu().numpy()))
        elif isinstance(wl, VocabEncoderBench):
            parity = bool(np.array_equal(ref[0], res[0][: len(ref[0])].cpu().numpy()))
        elif isinstance(wl, PipelineEncode):
            parity = bool(wl.check(ref, res))
        else:
            b, e, payload = res
            n_chk = len(ref[1])
            got_ends = e[:n_chk].cpu().numpy()
            parity = bool(np.array_equal(ref[1], got_ends) and np.array_equal(ref[2][: int(ref[1][-1])], payload[: int(got_ends[-1])].cpu().numpy()))
        cpu_baseline = {"value": round(sample_units / cdt / (1e9 if wl.unit == "GB/s" else 1e6), 3),
                        "unit": wl.unit if not latency else "MB/s", "cores": 1, "kind": "port",
                        "sample": f"{passes} pass(es) over the first {n_s} rows of batch 0 ({sample_units} units in all), {what}, {cdt:.2f} s",
                        "host_cpus": os.cpu_count()}
        if not args.no_extras:
            if affinity0:
                os.sched_setaffinity(0, affinity0)   # this leg uses every host core
            cpu_all = cpu_all_cores(wl)
            cpu_all_private = cpu_all_cores(wl, seconds=6.0, private=True)
            cpu_curve = cpu_thread_curve(wl)
    if world == 1 and not args.no_extras and args.config == "2" and args.text == "zipf" and not args.no_memo:
        stress = {}
        from tools.workloads import MODEL_PATTERNS
        legs = [("uniform_text", dict(kind="uniform")), ("no_memo", dict(kind="zipf", no_memo=True)),
                ("fixed_memo_only", dict(kind="zipf", cache_capacity=1, memo_learn=-1)),
                ("reference_cache_count", dict(kind="zipf", memo_learn=-1)), ("mixed_script_text", dict(kind="mixed")),
                # first sight of every text: 20 distinct batches, none encoded before -- memo and store learn as they go
                ("first_sight", dict(kind="zipf", fresh=True)),
                # rows longer than a scan block (VERDICT r03 missing 3): the same bytes per batch in fewer, longer rows
                ("rows_2048_bytes", dict(kind="zipf", nbytes=2048)), ("rows_8192_bytes", dict(kind="zipf", nbytes=8192)),
                # the Llama-3-shaped tokenizer on config 4's text behind other models' split patterns (VERDICT r03 missing 1)
                ("pattern_qwen2", dict(kind="mixed", tok="llama3", pattern="qwen2")), ("pattern_cl100k", dict(kind="mixed", tok="llama3", pattern="cl100k")),
                ("pattern_o200k", dict(kind="mixed", tok="llama3", pattern="o200k")),
                ("pattern_deepseek_v3", dict(kind="mixed", tok="llama3", pattern="deepseek-v3"))]
        notes = {"uniform_text": "uniform-random printable bytes (SURVEY 8d stress text: cache-hostile, 3x the pieces)",
                 "no_memo": "zipf text with cache_capacity=0: every piece takes the merge path",
                 "fixed_memo_only": "zipf text with cache_capacity=1 and memo_learn=-1: the memo holds the vocabulary's own tokens and learns "
                                    "nothing from the text (every multi-token word goes to merge_kernel every time, which finds it in the store)",
                 "reference_cache_count": "the headline text with memo_learn=-1: the first level learns exactly cache_capacity = 20 000 pieces, the "
                                          "reference's count (round 4's configuration); everything else goes to merge_kernel and its store every call",
                 "mixed_script_text": "config 4's text (30 % of the words Greek / Cyrillic / CJK / kana / emoji, rows of 700-1000 "
                                      "bytes) through THIS tokenizer: blocks with a non-ASCII byte take the ballot form of the rules inside "
                                      "lookup_span_kernel (window by window), and random non-Latin words never hit the memo: half of the step is merge_kernel",
                 "first_sight": "zipf text never encoded before: 20 timed batches, each seen for the first time, on a handle that has seen 4 others "
                                "(the memo and the piece store keep learning)",
                 "rows_2048_bytes": "rows of ~2048 bytes: lookup_span_kernel's 2048-byte blocks slide along the row (a block starts where the one before stopped)",
                 "rows_8192_bytes": "rows of ~8192 bytes: four to five sliding blocks of lookup_span_kernel per row",
                 "pattern_qwen2": "Qwen2's pattern (\\p{N} for \\p{N}{1,3}): the Llama-3 scanners with l3_digits1",
                 "pattern_cl100k": "tiktoken's cl100k_base pattern (possessive, \\s++$): the Llama-3 scanners with l3_tail_ws",
                 "pattern_o200k": "o200k_base's pattern: its rules on bit masks in lookup_span_kernel<kSpanO200k> (csrc/span_fam.hpp; until round 5 the compiled DFA)",
                 "pattern_deepseek_v3": "DeepSeek-V3's main pattern: lookup_span_kernel<kSpanDs3> (csrc/span_fam.hpp)"}
        for name, kw in legs:
            s_args = argparse.Namespace(**vars(args))
            nbytes = kw.get("nbytes", args.bytes)
            rows = max(256, args.rows * args.bytes // nbytes) if "nbytes" in kw else (args.rows // 2 if "pattern" in kw else args.rows)
            fresh = kw.get("fresh", False)
            n_b = 24 if fresh else 4
            w2 = BpeEncode(s_args, lib, dev, rank, kw.get("tok", args.tokenizer), kw["kind"], rows, nbytes, 5000, no_memo=kw.get("no_memo", False),
                           n_batches=n_b, cache_capacity=kw.get("cache_capacity"), pattern=MODEL_PATTERNS.get(kw.get("pattern")),
                           memo_learn=kw.get("memo_learn"))
            run_pipelined(w2, 4, stream_ptrs, args.depth)
            n2 = 20 if fresh else 16
            d2 = timed(lambda: run_pipelined(w2, n2, stream_ptrs, args.depth, first=4))
            u2 = sum(w2.units(i) for i in range(4, 4 + n2))
            stress[name] = {"value": round(u2 / d2 / 1e6, 1), "unit": "MB/s", "ms_per_step": round(d2 / n2 * 1e3, 4),
                            "ids_per_batch": round(w2.mean_out()), "rows": rows, "bytes_per_row": nbytes, "note": notes[name]}
            del w2
        e2e = end_to_end_in_child(args)

    line = {
        "metric": wl.metric, "value": round(value, 1 if not latency else 2), "unit": wl.unit,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": getattr(wl, "higher_is_better", True), "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype,
        "data": "synthetic",
        "config": {"workload": wl.workload,
                   "priming": (f"every distinct batch of the rotation once ({wl.batches.n} untimed steps) before the {args.warmup} warm-up steps: "
                               "the piece memo / store fill on first sight of a text, like the reference's piece cache"
                               if hasattr(wl, "batches") and not latency_cfg else None),
                   "world": {"ranks": (dist.get_world_size() if dist_on else 1),
                             "launched_by": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "python"},
                   "row_tickets": row_tickets, "host_cpu_affinity": numa_note,
                   "host_loop": (f"launch batch k, then complete batch k-{args.depth} (two-half calls), batches alternate between {len(stream_ptrs)} "
                                 f"HIP stream(s)" if two_half else "one blocking call per batch"),
                   "units_per_gpu_and_step": round(my_units / args.steps), "outputs_per_gpu_and_step": round(wl.mean_out()),
                   "exchange": ("none (1 GPU)" if exchange is None and world == 1 else ("none (rank-local consumer)" if exchange is None else
                                                                    f"{'all-gather' if exchange.transport == 'allgather' else 'grouped direct send/recv'} of ragged ids over RCCL, {exchange.id_bytes}-byte ids on the wire, "
                                                                    f"{'written by the encode itself (compact_kernel), ' if to_wire else 'packed by shard_pack_kernel, '}"
                                                                    f"gather overlapped with the next encode, unpack one batch later ({exchange.regathers} re-gathers)"))},
        "value_steady": steady["value"] if steady else None, "steady": steady,
        "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baseline_all_cores": cpu_all, "cpu_baseline_all_cores_private": cpu_all_private,
        "cpu_baseline_thread_curve": cpu_curve, "stress": stress, "end_to_end": e2e,
        "parity_prefix_bit_exact": parity,
        "parity_per_chunk": (wl.parity_per_chunk() if hasattr(wl, "parity_per_chunk") and world == 1 and not args.no_cpu_baseline else None),
        "kernel_ms": kernels,
    }
    op_t, op_x = C.c_int64(), C.c_int64()
    L.check(lib, lib.ovtk_short_path_stats(C.byref(op_t), C.byref(op_x)))
    line["config"]["short_path"] = {"mode": args.short_path, "calls_tried": int(op_t.value), "calls_that_needed_no_other_kernel": int(op_x.value),
                                    "note": "whole process (priming, warm-up, timed steps, stress legs): calls launched as lookup_span_kernel -> compact_kernel"}
    if hasattr(wl, "memo"):
        # the reference's piece cache (m_cache, cache_capacity = 20000) fills during ITS first calls too; here it is full
        # before the warm-up steps are over, so the timed steps run with the learned entries in place
        line["config"]["piece_memo"] = wl.memo()
    print(json.dumps(line))
    if exchange is not None:
        exchange.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
