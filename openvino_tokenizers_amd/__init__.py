"""MI355X-native tokenizer hot path of openvino_tokenizers (RegexSplit, BPETokenizer, WordpieceTokenizer,
VocabEncoder, RaggedToDense, VocabDecoder, ByteFallback, FuzeRagged, plus SpecialTokensSplit, UTF8Validate,
Truncate and CombineSegments either side of it) behind the reference's op interface.

Compute happens only in csrc/build/libovtk_amd.so (hand-written HIP for gfx950) through the C ABI of
include/ovtk_amd.h; see DESIGN.md and INTEGRATION.md.
"""
from ._lib import OvtkError, load  # noqa: F401
from .ops import (BPETokenizer, ByteFallback, CombineSegments, FusedDetokenizer, FusedEncodeDense, FusedEncodeTail, FusedSpecialSplitBPE, FusedSplitBPE, FusedSplitWordpiece, FuzeRagged, RaggedToDense,  # noqa: F401
                  RegexSplit, SpecialTokensSplit, StringTensorPack, StringTensorUnpack, TrieTokenizer, Truncate, UTF8Validate, VocabDecoder, VocabEncoder, WordpieceTokenizer)
