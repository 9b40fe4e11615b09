"""MI355X-native tokenizer hot path of openvino_tokenizers (RegexSplit, BPETokenizer, WordpieceTokenizer,
VocabEncoder, RaggedToDense, VocabDecoder, ByteFallback, FuzeRagged) behind the reference's op interface.

Compute happens only in csrc/build/libovtk_amd.so (hand-written HIP for gfx950) through the C ABI of
include/ovtk_amd.h; see DESIGN.md and INTEGRATION.md.
"""
from ._lib import OvtkError, load  # noqa: F401
from .ops import (BPETokenizer, ByteFallback, FusedDetokenizer, FusedSplitBPE, FusedSplitWordpiece, FuzeRagged, RaggedToDense,  # noqa: F401
                  RegexSplit, SpecialTokensSplit, VocabDecoder, VocabEncoder, WordpieceTokenizer)
