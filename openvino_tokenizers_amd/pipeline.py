"""The reference's tokenizer graph as a list of steps over the mirror ops of ops.py, and `fuse()`: the rewrite that puts the fused
entry points behind the operator interface.

The reference builds a converted tokenizer as a chain of custom-op nodes, each one an `evaluate()` of its own
(python/openvino_tokenizers/tokenizer_pipeline.py: SpecialTokensSplit -> RegexSplitStep :392-489 -> BPETokenizationStep :773-822 /
WordPieceTokenizationStep :641-659 -> TruncationStep -> CombineSegmentsStep -> PaddingStep :1211-1236, assembled :1613-1636; the
detokenizer VocabDecoderStep -> ByteFallbackStep / FuseStep :1321-1371).  `Pipeline(steps).run(...)` executes such a chain op by op
through the C ABI, one library call per node -- what `core.add_extension()` + the op classes of adapter/ give an OpenVINO user.
`fuse(steps)` recognises the sub-chains the library has one call for and replaces them:

  [SpecialTokensSplit] RegexSplit(isolate, a pattern the span kernel scans) BPETokenizer Truncate CombineSegments(constant ids in front
      / behind) Padding                                         -> FusedEncodeDenseStep      ovtk_encode_dense_enqueue / _finish
  [SpecialTokensSplit] RegexSplit BPETokenizer                  -> FusedSplitBPEStep          ovtk_encode_run / ovtk_encode_special_run
  RegexSplit(\\s+, remove) RegexSplit(BERT delimiters, isolate) WordpieceTokenizer
                                                                -> FusedSplitWordpieceStep    ovtk_wordpiece_encode_run
  Truncate CombineSegments Padding                              -> FusedEncodeTailStep        ovtk_encode_tail_run
  VocabDecoder [ByteFallback] FuzeRagged                        -> FusedDetokenizeStep        ovtk_detokenize_run

and leaves every other step as it is.  The rewritten list gives the same outputs as the original one (tests/test_pipeline_fuse.py:
bit for bit, on BASELINE.json's configurations); adapter/fuse_pass.cpp is the same recogniser over ov::Node chains.
"""
from __future__ import annotations

import numpy as np

from . import ops as K

# ---------------------------------------------------------------------------------------------------------------- state
# What flows between steps: ("strings", [ragged_begins, ragged_ends, begins, ends, chars, skips | None]) -- a ragged tensor of strings --,
# ("ids", [begins, ends, ids]) -- ragged token ids --, ("dense", [input_ids, attention_mask]), ("tokens", [ids[B, S]]) and
# ("token_strings", [ragged_begins, ragged_ends, begins, ends, chars]) / ("text", [begins, ends, chars]) on the way back.


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _u8(s):
    return np.frombuffer(s if isinstance(s, bytes) else s.encode(), np.uint8)


def _like(ref, values, dtype=np.int32):
    """A small constant next to `ref` (a device tensor stays on its device)."""
    a = np.asarray(values, dtype)
    if _is_torch(ref):
        import torch
        return torch.as_tensor(a, device=ref.device)
    return a


class Step:
    lib = None

    def apply(self, kind, vals):
        raise NotImplementedError


class SpecialTokensSplitStep(Step):
    """src/special_tokens_split.cpp:61-162; in front of every RegexSplit of a converted HF tokenizer (tokenizer_pipeline.py:1613-1636)."""

    def __init__(self, pattern, lib=None):
        self.pattern, self.op = _u8(pattern), K.SpecialTokensSplit(lib=lib)

    def apply(self, kind, vals):
        assert kind == "strings"
        ins = list(vals[:5]) + ([vals[5]] if vals[5] is not None else []) + [self.pattern]
        out = self.op.evaluate(ins)
        return "strings", list(out[:6])


class RegexSplitStep(Step):
    """src/regex_split.cpp:124-324 (tokenizer_pipeline.py:475-489)."""

    def __init__(self, pattern, behaviour="isolate", invert=False, max_splits=-1, lib=None):
        self.pattern, self.behaviour, self.invert, self.max_splits = pattern, behaviour, invert, max_splits
        self.op = K.RegexSplit(behaviour, invert=invert, max_splits=max_splits, lib=lib)

    def apply(self, kind, vals):
        assert kind == "strings"
        ins = list(vals[:5]) + ([vals[5]] if vals[5] is not None else []) + [_u8(self.pattern)]
        out = self.op.evaluate(ins)
        return "strings", list(out[:5]) + [out[5] if len(out) > 5 else None]


class BPETokenizationStep(Step):
    """src/bpe_tokenizer.cpp:47-164 (tokenizer_pipeline.py:773-822).  consts: inputs 5.. of the op."""

    def __init__(self, consts, lib=None, **attrs):
        self.consts, self.op = list(consts), K.BPETokenizer(**attrs, lib=lib)

    def apply(self, kind, vals):
        assert kind == "strings"
        return "ids", self.op.evaluate(list(vals[:5]) + self.consts)   # (BPETokenizer has no skips input: skipped strings arrive as whole pieces)


class WordPieceTokenizationStep(Step):
    """src/wordpiece_tokenizer.cpp:49-133 (tokenizer_pipeline.py:641-659).  consts: vocab (3) + unk_token_id."""

    def __init__(self, consts, suffix_indicator="##", max_bytes_per_word=100, lib=None):
        self.consts, self.op = list(consts), K.WordpieceTokenizer(suffix_indicator, max_bytes_per_word, lib=lib)

    def apply(self, kind, vals):
        assert kind == "strings"
        return "ids", self.op.evaluate(list(vals[:5]) + self.consts)


class TruncationStep(Step):
    """src/truncate.cpp:37-150, one input."""

    def __init__(self, max_length, side="right", lib=None):
        self.max_length, self.side, self.op = int(max_length), side, K.Truncate(lib=lib)

    def apply(self, kind, vals):
        assert kind == "ids"
        b, e, ids = vals
        return "ids", self.op.evaluate([b, e, ids, np.int32(self.max_length), self.side.encode(), b"longest_first"])


class CombineSegmentsStep(Step):
    """src/combine_segments.cpp:36-134 with constant segments in front of / behind the sequence (the post-processor's BOS / EOS)."""

    def __init__(self, prefix=(), suffix=(), lib=None):
        self.prefix, self.suffix, self.op = [int(x) for x in prefix], [int(x) for x in suffix], K.CombineSegments(lib=lib)

    def apply(self, kind, vals):
        assert kind == "ids"
        b, e, ids = vals
        segs = []
        if self.prefix:
            segs += [_like(ids, [0]), _like(ids, [len(self.prefix)]), _like(ids, self.prefix)]
        segs += [b, e, ids]
        if self.suffix:
            segs += [_like(ids, [0]), _like(ids, [len(self.suffix)]), _like(ids, self.suffix)]
        k = len(segs) // 3
        return "ids", self.op.evaluate(segs + [np.arange(k, dtype=np.int32)])[:3]


class PaddingStep(Step):
    """src/ragged_to_dense.cpp:70-174 twice: input_ids (pad id) and attention_mask (tokenizer_pipeline.py:1211-1236: the target is the
    longest row, or `target_dim`)."""

    def __init__(self, pad_value=0, pad_right=True, target_dim=None, lib=None):
        self.pad_value, self.pad_right, self.target_dim, self.op = int(pad_value), bool(pad_right), target_dim, K.RaggedToDense(pad_right=pad_right, lib=lib)

    def apply(self, kind, vals):
        assert kind == "ids"
        b, e, ids = vals
        lens = (e - b)
        width = int(self.target_dim) if self.target_dim is not None else (int(lens.max()) if len(lens) else 0)
        dense, mask = self.op.evaluate([b, e, ids, np.int32(width), np.int32(self.pad_value)])
        return "dense", [dense, mask]


class VocabDecoderStep(Step):
    """src/vocab_decoder.cpp:23-87 (tokenizer_pipeline.py:1321-1338).  vocab: begins, ends, chars."""

    def __init__(self, vocab, skip_tokens=(), lib=None):
        self.vocab, self.op = list(vocab), K.VocabDecoder(skip_tokens=skip_tokens, lib=lib)

    def apply(self, kind, vals):
        assert kind == "tokens"
        return "token_strings", self.op.evaluate([vals[0]] + self.vocab)


class ByteFallbackStep(Step):
    """src/byte_fallback.cpp:16-50 (tokenizer_pipeline.py:1363-1371)."""

    def __init__(self, lib=None):
        self.op = K.ByteFallback(lib=lib)

    def apply(self, kind, vals):
        assert kind == "token_strings"
        rb, re_, b, e, c = vals
        return "token_strings", [rb, re_] + self.op.evaluate([b, e, c])


class FuseStep(Step):
    """src/fuze.cpp:20-40 (tokenizer_pipeline.py:1347-1351): a row's token strings become one string."""

    def __init__(self, lib=None):
        self.op = K.FuzeRagged(lib=lib)

    def apply(self, kind, vals):
        assert kind == "token_strings"
        rb, re_, b, e, c = vals
        return "text", self.op.evaluate([rb, re_, b, e]) + [c]


# ---------------------------------------------------------------------------------------------------------------- fused steps
class FusedSplitBPEStep(Step):
    """[SpecialTokensSplit ->] RegexSplit -> BPETokenizer: ovtk_encode_run / ovtk_encode_special_run."""

    def __init__(self, special, split, bpe):
        self.special, self.split, self.bpe = special, split, bpe
        self.op = K.FusedSpecialSplitBPE(special.op, split.op, bpe.op) if special is not None else K.FusedSplitBPE(split.op, bpe.op)

    def apply(self, kind, vals):
        assert kind == "strings"
        ins = list(vals[:5]) + ([vals[5]] if vals[5] is not None else [])
        if self.special is not None:
            return "ids", self.op.evaluate(ins + [self.special.pattern], _u8(self.split.pattern), self.bpe.consts)
        return "ids", self.op.evaluate(ins + [_u8(self.split.pattern)], self.bpe.consts)


class FusedEncodeDenseStep(Step):
    """... -> Truncate -> CombineSegments -> Padding as the sink of the encode's last pass: ovtk_encode_dense_enqueue / _finish (device tensors).
    Host arrays (a CPU-plugin style caller) take the two-call form instead -- ovtk_encode_run / ovtk_encode_special_run, then
    ovtk_encode_tail_run --, both of which stage host memory themselves."""

    def __init__(self, special, split, bpe, trunc, comb, pad):
        self.special, self.split, self.bpe, self.pad = special, split, bpe, pad
        self.op = K.FusedEncodeDense(split.op, bpe.op, special.op if special is not None else None,
                                     max_length=trunc.max_length if trunc is not None else 2**31 - 1, trunc_side=trunc.side if trunc is not None else "right",
                                     pad_right=pad.pad_right, pad_value=pad.pad_value, prefix=comb.prefix if comb is not None else (),
                                     suffix=comb.suffix if comb is not None else ())
        self.host_form = [FusedSplitBPEStep(special, split, bpe), FusedEncodeTailStep(trunc, comb, pad, lib=pad.op._lib)]

    def apply(self, kind, vals):
        assert kind == "strings"
        if not _is_torch(vals[4]) and not hasattr(self.bpe.op._lib, "ovtk_emulator_build"):
            for s in self.host_form:
                kind, vals = s.apply(kind, vals)
            return kind, vals
        ins = list(vals[:5]) + ([vals[5]] if vals[5] is not None else [])
        return "dense", self.op.evaluate(ins, _u8(self.split.pattern), self.bpe.consts,
                                         special_pattern=self.special.pattern if self.special is not None else None, target_dim=self.pad.target_dim)


class FusedSplitWordpieceStep(Step):
    """RegexSplit(\\s+, remove) -> RegexSplit(BERT delimiters, isolate) -> WordpieceTokenizer: ovtk_wordpiece_encode_run."""

    def __init__(self, ws, pu, wp):
        self.ws, self.pu, self.wp = ws, pu, wp
        self.op = K.FusedSplitWordpiece(ws.op, pu.op, wp.op)

    def apply(self, kind, vals):
        assert kind == "strings" and vals[5] is None
        return "ids", self.op.evaluate(list(vals[:5]), _u8(self.ws.pattern), _u8(self.pu.pattern), self.wp.consts)


class FusedEncodeTailStep(Step):
    """Truncate -> CombineSegments -> Padding in one kernel: ovtk_encode_tail_run (the ragged ids exist already: WordPiece, or a BPE chain
    whose split is none the span kernel scans)."""

    def __init__(self, trunc, comb, pad, lib=None):
        self.trunc, self.comb, self.pad = trunc, comb, pad
        self.op = K.FusedEncodeTail(max_length=trunc.max_length if trunc is not None else 2**31 - 1, trunc_side=trunc.side if trunc is not None else "right",
                                    pad_right=pad.pad_right, lib=lib)

    def apply(self, kind, vals):
        assert kind == "ids"
        b, e, ids = vals
        segs, main = [], 0
        if self.comb is not None and self.comb.prefix:
            segs.append((_like(ids, [0]), _like(ids, [len(self.comb.prefix)]), _like(ids, self.comb.prefix)))
            main = 1
        segs.append((b, e, ids))
        if self.comb is not None and self.comb.suffix:
            segs.append((_like(ids, [0]), _like(ids, [len(self.comb.suffix)]), _like(ids, self.comb.suffix)))
        out = self.op.evaluate(segs, np.arange(len(segs), dtype=np.int32), truncated=(main,) if self.trunc is not None else (), pad_value=self.pad.pad_value,
                               target_dim=self.pad.target_dim)
        return "dense", out[:2]


class FusedDetokenizeStep(Step):
    """VocabDecoder -> [ByteFallback] -> FuzeRagged: ovtk_detokenize_run."""

    def __init__(self, dec, byte_fallback):
        self.dec = dec
        self.op = K.FusedDetokenizer(dec.op, byte_fallback=byte_fallback)

    def apply(self, kind, vals):
        assert kind == "tokens"
        return "text", self.op.evaluate([vals[0]] + self.dec.vocab)


# ---------------------------------------------------------------------------------------------------------------- the rewrite
def _span_pattern(step):
    """A split the fused encode has a scanner for (api_encode.cpp ovtk_regex_split_create: pattern equality picks it): isolate, no invert, no
    max_splits.  Any other RegexSplit is still fusable with BPETokenizer (the compiled DFA runs inside the call), but only ONE split."""
    return isinstance(step, RegexSplitStep) and step.behaviour in ("isolate", "contiguous") and not step.invert and step.max_splits == -1


# the converter's two patterns of the BERT pre-tokenizer (tokenizer_pipeline.py:392-431)
BERT_WS = r"\s+"
BERT_PUNCT = "|".join([r"[!-/]", r"[:-@]", r"[\[-`]", r"[{-~]", r"[\p{P}]", r"[\x{4E00}-\x{9FFF}]", r"[\x{3400}-\x{4DBF}]", r"[\x{20000}-\x{2A6DF}]",
                       r"[\x{2A700}-\x{2B73F}]", r"[\x{2B740}-\x{2B81F}]", r"[\x{2B820}-\x{2CEAF}]", r"[\x{F900}-\x{FAFF}]", r"[\x{2F800}-\x{2FA1F}]"])


def _is_bert_split(a, b):
    return (isinstance(a, RegexSplitStep) and isinstance(b, RegexSplitStep) and a.pattern == BERT_WS and a.behaviour == "remove" and not a.invert and
            b.pattern == BERT_PUNCT and b.behaviour == "isolate" and not b.invert and a.max_splits == -1 and b.max_splits == -1)


def _tail(steps, i):
    """[Truncation] [CombineSegments with at most four constant ids each side] Padding at steps[i:] -> (trunc, comb, pad, next index) or None."""
    trunc = comb = None
    j = i
    if j < len(steps) and isinstance(steps[j], TruncationStep):
        trunc, j = steps[j], j + 1
    if j < len(steps) and isinstance(steps[j], CombineSegmentsStep):
        if len(steps[j].prefix) > 4 or len(steps[j].suffix) > 4:
            return None
        comb, j = steps[j], j + 1
    if j < len(steps) and isinstance(steps[j], PaddingStep):
        return trunc, comb, steps[j], j + 1
    return None


def fuse(steps):
    """The list with every recognised sub-chain replaced by its fused step (module docstring); the input list is not modified."""
    out, i, n = [], 0, len(steps)
    while i < n:
        s = steps[i]
        # ---- [SpecialTokensSplit] RegexSplit BPETokenizer [tail]
        j = i
        special = None
        if isinstance(s, SpecialTokensSplitStep) and j + 1 < n:
            special, j = s, j + 1
        if j + 1 < n and isinstance(steps[j], RegexSplitStep) and isinstance(steps[j + 1], BPETokenizationStep) and _span_pattern(steps[j]):
            split, bpe = steps[j], steps[j + 1]
            t = _tail(steps, j + 2)
            if t is not None:
                out.append(FusedEncodeDenseStep(special, split, bpe, *t[:3]))
                i = t[3]
            else:
                out.append(FusedSplitBPEStep(special, split, bpe))
                i = j + 2
            continue
        # ---- the BERT pre-tokenizer + WordPiece
        if i + 2 < n and _is_bert_split(steps[i], steps[i + 1]) and isinstance(steps[i + 2], WordPieceTokenizationStep):
            out.append(FusedSplitWordpieceStep(steps[i], steps[i + 1], steps[i + 2]))
            i += 3
            continue
        # ---- a tail on ragged ids that exist
        if isinstance(s, (TruncationStep, CombineSegmentsStep, PaddingStep)):
            t = _tail(steps, i)
            if t is not None and (t[0] is not None or t[1] is not None):
                out.append(FusedEncodeTailStep(t[0], t[1], t[2], lib=t[2].op._lib))
                i = t[3]
                continue
        # ---- the detokenizer
        if isinstance(s, VocabDecoderStep):
            j = i + 1
            bf = j < n and isinstance(steps[j], ByteFallbackStep)
            if bf:
                j += 1
            if j < n and isinstance(steps[j], FuseStep):
                out.append(FusedDetokenizeStep(s, bf))
                i = j + 1
                continue
        out.append(s)
        i += 1
    return out


class Pipeline:
    """A chain of steps.  run(kind, values): the state a chain starts from -- ("strings", [ragged_begins, ragged_ends, begins, ends, chars, skips
    or None]) for a tokenizer, ("tokens", [ids[B, S]]) for a detokenizer -- through every step; returns the last state's values."""

    def __init__(self, steps):
        self.steps = list(steps)

    def fused(self):
        return Pipeline(fuse(self.steps))

    def run(self, kind, values):
        vals = list(values)
        for s in self.steps:
            kind, vals = s.apply(kind, vals)
        return vals
