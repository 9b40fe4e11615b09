"""openvino_tokenizers_amd.pipeline: the reference's tokenizer graphs as chains of steps over the mirror ops, and fuse() -- the rewrite
that replaces the sub-chains the library has ONE call for (VERDICT r05 missing 2: the fused entry points behind the operator
interface).  Every fused chain must give what the op-by-op chain gives, bit for bit, and the op-by-op chain what the oracle's chain
gives (tokenizer_pipeline.py:1613-1636, :392-435 + :641-659, :1321-1371)."""
import numpy as np
import pytest

from openvino_tokenizers_amd import pipeline as P
from oracle import oracle as O
from tests.util import BpeTok, assert_same, one_string_per_row
from tools.harness import pack_strings
from tools.make_tokenizers import load_tokenizer
from tools.workloads import TextModel, ragged_rows


def _state(backend, rb, re_, b, e, c, skips=None):
    d = backend.data([rb, re_, b, e, c])
    return d + [backend.data([np.asarray(skips, np.uint8)])[0] if skips is not None else None]


def _kinds(steps):
    return [type(s).__name__ for s in steps]


@pytest.mark.parametrize("name, with_special, tail", [("gpt2", False, True), ("gpt2", True, True), ("gpt2", False, False), ("llama3", False, True)])
def test_bpe_graph_fused_equals_op_by_op(backend, name, with_special, tail):
    """[SpecialTokensSplit ->] RegexSplit -> BPETokenizer [-> Truncate -> CombineSegments -> Padding]."""
    from tests.test_special_tokens import _texts_with_specials
    lib = backend.lib
    small = backend.name == "emu"
    tok = BpeTok.load(name + "_small" if small else name)
    n = 300 if small else 4000
    if with_special:
        rng = np.random.default_rng(11)
        inputs = one_string_per_row(_texts_with_specials(rng, n))
    else:
        b, e, c = TextModel(5, "mixed" if name == "llama3" else "zipf").batch(n, 120)
        rb, re_ = ragged_rows(n)
        inputs = [rb, re_, b, e, c]
    steps = []
    if with_special:
        steps.append(P.SpecialTokensSplitStep(O.special_tokens_pattern([("<|endoftext|>", False, False)]), lib=lib))
    steps += [P.RegexSplitStep(tok.pattern, "isolate", lib=lib), P.BPETokenizationStep(tok.consts, lib=lib, **tok.attrs)]
    if tail:
        steps += [P.TruncationStep(48, "right", lib=lib), P.CombineSegmentsStep(prefix=[1], suffix=[2, 3], lib=lib), P.PaddingStep(pad_value=0, lib=lib)]
    pipe = P.Pipeline(steps)
    fused = pipe.fused()
    assert _kinds(fused.steps) == (["FusedEncodeDenseStep"] if tail else ["FusedSplitBPEStep"])
    ref = pipe.run("strings", _state(backend, *inputs))
    for rep in range(2):   # (the second call runs on what the tables learned)
        got = fused.run("strings", _state(backend, *inputs))
        assert_same([backend.host(x) for x in ref], got, backend.host, f"{name} special={with_special} tail={tail}, call {rep}")
    # ... and the op-by-op chain is the oracle's
    if not with_special and not tail:
        o = tok.oracle()(*O.RegexSplit(tok.pattern, "isolate")(*inputs)[:5])
        assert_same(o, ref, backend.host, "op by op vs oracle")


def test_split_the_span_kernel_has_no_scanner_for(backend):
    """Two splits in a row (DeepSeek-style) are not one fused call: the first stays an op, the second fuses with BPETokenizer."""
    lib = backend.lib
    tok = BpeTok.load("gpt2_small")
    b, e, c = TextModel(6, "zipf").batch(280, 90)
    rb, re_ = ragged_rows(280)
    steps = [P.RegexSplitStep(r"\d", "isolate", lib=lib), P.RegexSplitStep(tok.pattern, "isolate", lib=lib), P.BPETokenizationStep(tok.consts, lib=lib, **tok.attrs),
             P.PaddingStep(pad_value=7, pad_right=False, lib=lib)]
    pipe = P.Pipeline(steps)
    fused = pipe.fused()
    assert _kinds(fused.steps) == ["RegexSplitStep", "FusedEncodeDenseStep"]
    ref = pipe.run("strings", _state(backend, rb, re_, b, e, c))
    assert_same([backend.host(x) for x in ref], fused.run("strings", _state(backend, rb, re_, b, e, c)), backend.host, "two splits")


def test_wordpiece_graph_fused_equals_op_by_op(backend):
    """RegexSplit(\\s+, remove) -> RegexSplit(delimiters, isolate) -> WordpieceTokenizer -> Truncate -> CombineSegments -> Padding (BERT)."""
    lib = backend.lib
    small = backend.name == "emu"
    tok = load_tokenizer("bert_small" if small else "bert")
    n = 300 if small else 4000
    b, e, c = TextModel(7, "zipf").batch(n, 100)
    c = np.frombuffer(c.tobytes().lower(), np.uint8).copy()
    rb, re_ = ragged_rows(n)
    consts = list(pack_strings(tok["vocab"])) + [np.asarray(tok["unk_id"], np.int32)]
    steps = [P.RegexSplitStep(P.BERT_WS, "remove", lib=lib), P.RegexSplitStep(P.BERT_PUNCT, "isolate", lib=lib),
             P.WordPieceTokenizationStep(consts, tok["suffix_indicator"], tok["max_bytes_per_word"], lib=lib),
             P.TruncationStep(30, "left", lib=lib), P.CombineSegmentsStep(prefix=[101], suffix=[102], lib=lib), P.PaddingStep(pad_value=0, lib=lib)]
    pipe = P.Pipeline(steps)
    fused = pipe.fused()
    assert _kinds(fused.steps) == ["FusedSplitWordpieceStep", "FusedEncodeTailStep"]
    ref = pipe.run("strings", _state(backend, rb, re_, b, e, c))
    for rep in range(2):
        assert_same([backend.host(x) for x in ref], fused.run("strings", _state(backend, rb, re_, b, e, c)), backend.host, f"BERT graph, call {rep}")


@pytest.mark.parametrize("byte_fallback", [False, True])
def test_detokenizer_graph_fused_equals_op_by_op(backend, byte_fallback):
    """VocabDecoder -> [ByteFallback] -> FuzeRagged."""
    lib = backend.lib
    tok = BpeTok.load("gpt2_small")
    vocab = list(pack_strings(tok.vocab))
    rng = np.random.default_rng(3)
    ids = rng.integers(0, len(tok.vocab), size=(70, 33)).astype(np.int32)
    steps = [P.VocabDecoderStep(vocab, skip_tokens=[0, 5], lib=lib)] + ([P.ByteFallbackStep(lib=lib)] if byte_fallback else []) + [P.FuseStep(lib=lib)]
    pipe = P.Pipeline(steps)
    fused = pipe.fused()
    assert _kinds(fused.steps) == ["FusedDetokenizeStep"]
    ref = pipe.run("tokens", backend.data([ids]))
    assert_same([backend.host(x) for x in ref], fused.run("tokens", backend.data([ids])), backend.host, "detokenizer graph")


def test_nothing_to_fuse_is_left_alone(backend):
    lib = backend.lib
    steps = [P.RegexSplitStep(r"\s+", "remove", lib=lib), P.RegexSplitStep(r"[a-z]+", "isolate", invert=True, lib=lib)]
    assert _kinds(P.fuse(steps)) == ["RegexSplitStep", "RegexSplitStep"]
    assert P.fuse(steps)[0] is steps[0]
