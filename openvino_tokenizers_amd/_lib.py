"""ctypes binding of include/ovtk_amd.h.

The product library is `csrc/build/libovtk_amd.so` (HIP, gfx950).  There is no CPU execution
path: if the library is missing or no HIP device is present, loading / handle creation raises.
(The test-suite can point `load(path)` at the SIMT-emulator build of the same sources to check
kernel logic on CPU -- that is test infrastructure and never happens implicitly.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
DEFAULT_LIB = _HERE / "csrc" / "build" / "libovtk_amd.so"

OVTK_OK = 0
E_ARG, E_CAPACITY, E_VOCAB, E_UNSUPPORTED, E_HIP, E_RANGE = -1, -2, -3, -4, -5, -6
MEM_HOST, MEM_DEVICE = 0, 1

i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


class Strings(C.Structure):
    _fields_ = [("begins", C.c_void_p), ("ends", C.c_void_p), ("chars", C.c_void_p), ("n", C.c_int64),
                ("n_chars", C.c_int64)]


class RaggedStrings(C.Structure):
    _fields_ = [("ragged_begins", C.c_void_p), ("ragged_ends", C.c_void_p), ("n_rows", C.c_int64),
                ("strings", Strings)]


class RaggedI32Out(C.Structure):
    _fields_ = [("begins", C.c_void_p), ("ends", C.c_void_p), ("data", C.c_void_p), ("data_capacity", C.c_int64),
                ("n_data", C.c_int64), ("n_rows", C.c_int64)]


class RegexSplitParams(C.Structure):
    _fields_ = [("pattern", C.c_char_p), ("pattern_len", C.c_int64), ("behaviour", C.c_char_p), ("invert", C.c_int),
                ("max_splits", C.c_int), ("device", C.c_int)]


class RaggedStringsOut(C.Structure):
    _fields_ = [("ragged_begins", C.c_void_p), ("ragged_ends", C.c_void_p), ("n_rows", C.c_int64),
                ("begins", C.c_void_p), ("ends", C.c_void_p), ("skips", C.c_void_p), ("capacity", C.c_int64),
                ("n", C.c_int64)]


class BpeParams(C.Structure):
    _fields_ = [("vocab", Strings), ("merges", Strings), ("merges_right", Strings), ("added_tokens", Strings),
                ("added_ids", C.c_void_p), ("unk_token", C.c_char_p), ("unk_token_len", C.c_int64),
                ("fuse_unk", C.c_int), ("suffix_indicator", C.c_char_p), ("suffix_indicator_len", C.c_int64),
                ("end_suffix", C.c_char_p), ("end_suffix_len", C.c_int64), ("byte_fallback", C.c_int),
                ("cache_capacity", C.c_int64), ("device", C.c_int), ("memo_store", C.c_int64), ("memo_learn", C.c_int64)]


class WordpieceParams(C.Structure):
    _fields_ = [("vocab", Strings), ("suffix_indicator", C.c_char_p), ("suffix_indicator_len", C.c_int64),
                ("max_bytes_per_word", C.c_int), ("device", C.c_int), ("memo_store", C.c_int64)]


class VocabEncoderParams(C.Structure):
    _fields_ = [("keys", Strings), ("values", C.c_void_p), ("value_size", C.c_int), ("device", C.c_int)]


class VocabDecoderParams(C.Structure):
    _fields_ = [("vocab", Strings), ("skip_tokens", C.c_void_p), ("n_skip_tokens", C.c_int64), ("device", C.c_int)]


class StringsOut(C.Structure):
    _fields_ = [("begins", C.c_void_p), ("ends", C.c_void_p), ("chars", C.c_void_p), ("chars_capacity", C.c_int64),
                ("n_chars", C.c_int64)]


class RaggedI32(C.Structure):
    _fields_ = [("begins", C.c_void_p), ("ends", C.c_void_p), ("data", C.c_void_p), ("n", C.c_int64),
                ("n_data", C.c_int64)]


class EncodeTailParams(C.Structure):
    _fields_ = [("segs", C.c_void_p), ("n_segs", C.c_int), ("segment_ids", C.c_void_p), ("trunc_a", C.c_int), ("trunc_b", C.c_int),
                ("max_length", C.c_int32), ("trunc_side", C.c_char_p), ("trunc_mode", C.c_char_p), ("target_dim", C.c_int32),
                ("pad_value", C.c_int32), ("type_pad_value", C.c_int32), ("pad_right", C.c_int)]


class DenseParams(C.Structure):
    _fields_ = [("max_length", C.c_int32), ("trunc_left", C.c_int), ("pad_right", C.c_int), ("pad_value", C.c_int32), ("target_dim", C.c_int32),
                ("prefix", C.c_void_p), ("n_prefix", C.c_int), ("suffix", C.c_void_p), ("n_suffix", C.c_int)]


class ShardResult(C.Structure):
    _fields_ = [("n_ids", C.c_int64), ("max_shard_ids", C.c_int64), ("status", C.c_int64), ("reserved", C.c_int64)]


EXPORTS = [
    "ovtk_last_error", "ovtk_abi_version", "ovtk_device_name",
    "ovtk_regex_split_create", "ovtk_regex_split_run", "ovtk_regex_split_destroy",
    "ovtk_special_tokens_split_create", "ovtk_special_tokens_split_run", "ovtk_special_tokens_split_destroy",
    "ovtk_bpe_create", "ovtk_bpe_run", "ovtk_bpe_destroy", "ovtk_bpe_memo_entries", "ovtk_set_memo_store", "ovtk_bpe_store_entries", "ovtk_encode_run", "ovtk_encode_special_run", "ovtk_encode_special_enqueue", "ovtk_encode_dense_enqueue", "ovtk_encode_dense_finish", "ovtk_encode_enqueue", "ovtk_encode_enqueue_host", "ovtk_encode_enqueue_packed", "ovtk_encode_enqueue_wire", "ovtk_encode_finish", "ovtk_set_row_tickets", "ovtk_set_short_path", "ovtk_short_path_stats",
    "ovtk_wordpiece_create", "ovtk_wordpiece_run", "ovtk_wordpiece_encode_run", "ovtk_wordpiece_encode_enqueue", "ovtk_wordpiece_destroy",
    "ovtk_vocab_encoder_create", "ovtk_vocab_encoder_run", "ovtk_vocab_encoder_destroy",
    "ovtk_ragged_to_dense",
    "ovtk_vocab_decoder_create", "ovtk_vocab_decoder_run", "ovtk_vocab_decoder_destroy",
    "ovtk_byte_fallback", "ovtk_fuze_ragged", "ovtk_detokenize_run", "ovtk_detokenize_enqueue", "ovtk_detokenize_finish",
    "ovtk_utf8_validate", "ovtk_truncate", "ovtk_combine_segments", "ovtk_encode_tail_run",
    "ovtk_trie_tokenizer_create", "ovtk_trie_tokenizer_run", "ovtk_trie_tokenizer_destroy",
    "ovtk_string_tensor_packed_bytes", "ovtk_string_tensor_unpack", "ovtk_string_tensor_pack",
    "ovtk_shard_exchange_create", "ovtk_shard_max_rows", "ovtk_shard_wire_bytes", "ovtk_shard_pack", "ovtk_shard_unpack",
    "ovtk_shard_exchange_destroy",
    "ovtk_profile_enable", "ovtk_profile_reset", "ovtk_profile_get", "ovtk_profile_dump",
]


class OvtkError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        super().__init__(f"ovtk error {code}: {msg}")


_cache = {}


def load(path: os.PathLike | str | None = None) -> C.CDLL:
    """Loads the HIP library (default) or an explicitly given build.  Raises if it is missing."""
    product = path is None
    if product and os.environ.get("OVTK_AMD_LIB"):   # another build of the same HIP library (A/B runs of tools/)
        path = os.environ["OVTK_AMD_LIB"]
    p = Path(path) if path is not None else DEFAULT_LIB
    key = str(p)
    if key in _cache:
        return _cache[key]
    if not p.exists():
        raise OvtkError(E_HIP, f"{p} not found: build it with `make -C {_HERE / 'csrc'}` "
                               f"(__graft_entry__.build()); there is no CPU fallback")
    if product:
        # Share the process' HIP runtime with PyTorch (same SONAME libamdhip64.so.7): import torch first.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing, not a requirement
            pass
    lib = C.CDLL(str(p))
    lib.ovtk_last_error.restype = C.c_char_p
    lib.ovtk_device_name.restype = C.c_char_p
    lib.ovtk_profile_dump.restype = C.c_int64
    lib.ovtk_shard_wire_bytes.restype = C.c_int64
    lib.ovtk_string_tensor_packed_bytes.restype = C.c_int64
    lib.ovtk_shard_max_rows.restype = C.c_int64
    _cache[key] = lib
    return lib


def check(lib, rc):
    if rc != OVTK_OK:
        raise OvtkError(rc, lib.ovtk_last_error().decode(errors="replace"))
