"""ctypes front-end of the CPU oracle (oracle/oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never by the product package `openvino_tokenizers_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "build" / "liboracle.so"

i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> Path:
    src_m = max((_HERE / f).stat().st_mtime for f in ("oracle.cpp", "oracle.h"))
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src_m:
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_bpe_tie_events.restype = C.c_int64
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__(f"oracle error {code}: {lib().orc_last_error().decode()}")


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, np.uint8)[:0]
    return a, a.ctypes.data_as(u8p)


def pack_strings(strings):
    """list of bytes/str -> (begins i32, ends i32, chars u8): the reference's decomposed string
    tensor (python/openvino_tokenizers/utils.py:436-458)."""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
    lens = np.fromiter((len(b) for b in bs), dtype=np.int64, count=len(bs))
    ends = np.cumsum(lens).astype(np.int32)
    begins = (ends - lens).astype(np.int32)
    chars = np.frombuffer(b"".join(bs), dtype=np.uint8).copy()
    return begins, ends, chars


def unpack_strings(begins, ends, chars):
    cb = bytes(np.asarray(chars, dtype=np.uint8))
    return [cb[b:e] for b, e in zip(np.asarray(begins).tolist(), np.asarray(ends).tolist())]


# --------------------------------------------------------------------------- RegexSplit
def pcre2_compiles(pattern) -> bool:
    """True when the image's PCRE2 (PCRE2_UTF | PCRE2_UCP) accepts `pattern`."""
    return RegexSplit(pattern, "isolate").compiled()


class RegexSplit:
    def __init__(self, pattern, behaviour="remove", invert=False, max_splits=-1):
        p = pattern.encode("utf-8") if isinstance(pattern, str) else bytes(pattern)
        self._h = C.c_void_p()
        _chk(lib().orc_regex_split_create(p, C.c_int64(len(p)), behaviour.encode(), int(invert), int(max_splits),
                                          C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_regex_split_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def compiled(self) -> bool:
        """Did PCRE2 accept the pattern?  (A rejected one never matches: src/utils.cpp:264-271, 397-399.)"""
        return bool(lib().orc_regex_compiled(self._h))

    def match(self, s: bytes, start: int):
        m = (C.c_int64 * 2)()
        buf, p = _u8(np.frombuffer(s, dtype=np.uint8))
        if lib().orc_regex_match(self._h, p, C.c_int64(len(s)), C.c_int64(start), m):
            return int(m[0]), int(m[1])
        return None

    def __call__(self, rb, re_, begins, ends, chars, skips=None):
        rb, prb = _i32(rb)
        re_, pre = _i32(re_)
        begins, pb = _i32(begins)
        ends, pe = _i32(ends)
        chars, pc = _u8(chars)
        B, N, nch = len(rb), len(begins), len(chars)
        cap = nch + N
        if skips is not None:
            skips, ps = _u8(np.asarray(skips, dtype=np.uint8))
        else:
            ps = None
        orb = np.zeros(max(B, 1), np.int32)
        ore = np.zeros(max(B, 1), np.int32)
        ob = np.zeros(max(cap, 1), np.int32)
        oe = np.zeros(max(cap, 1), np.int32)
        osk = np.zeros(max(cap, 1), np.uint8)
        nrows = C.c_int64()
        nout = C.c_int64()
        _chk(lib().orc_regex_split_run(self._h, prb, pre, C.c_int64(B), pb, pe, C.c_int64(N), pc, C.c_int64(nch), ps,
                                       orb.ctypes.data_as(i32p), ore.ctypes.data_as(i32p), C.byref(nrows),
                                       ob.ctypes.data_as(i32p), oe.ctypes.data_as(i32p),
                                       osk.ctypes.data_as(u8p) if skips is not None else None,
                                       C.c_int64(cap), C.byref(nout)))
        if nout.value < 0:  # empty batch: string tensors alias the inputs
            out = [orb[:1].copy(), ore[:1].copy(), begins, ends, chars]
            if skips is not None:
                out.append(skips)
            return out
        n = nout.value
        out = [orb[:nrows.value].copy(), ore[:nrows.value].copy(), ob[:n].copy(), oe[:n].copy(), chars]
        if skips is not None:
            out.append(osk[:n].copy())
        return out


# --------------------------------------------------------------------------- SpecialTokensSplit
def special_tokens_pattern(tokens):
    """tokens: list of (text, strip_left, strip_right) -> the pattern SpecialTokensSplitStep builds
    (python/openvino_tokenizers/tokenizer_pipeline.py:138-159, utils.py:421-429 quote_meta)."""
    def quote_meta(t):
        return "".join(("\\" if not ch.isalnum() and ch not in ("_", "\u2581", "\uff5c") else "") + ch for ch in t)
    groups = {}
    for text, sl, sr in tokens:
        groups.setdefault((bool(sl), bool(sr)), []).append(text)
    return "|".join(r"(?:\s*)" * sl + "(" + "|".join(quote_meta(t) for t in toks) + ")" + r"(?:\s*)" * sr
                    for (sl, sr), toks in groups.items())


class SpecialTokensSplit:
    def __init__(self, pattern):
        self._re = RegexSplit(pattern, "isolate")

    def __call__(self, rb, re_, begins, ends, chars, skips=None):
        rb, prb = _i32(rb)
        re_, pre = _i32(re_)
        begins, pb = _i32(begins)
        ends, pe = _i32(ends)
        chars, pc = _u8(chars)
        B, cap = len(rb), len(chars) + len(begins) + 1
        ps = None
        if skips is not None:
            skips, ps = _u8(np.asarray(skips, dtype=np.uint8))
        orb, ore = np.zeros(B, np.int32), np.zeros(B, np.int32)
        ob, oe, osk = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.uint8)
        n = C.c_int64()
        _chk(lib().orc_special_tokens_split_run(self._re._h, prb, pre, C.c_int64(B), pb, pe, pc, ps,
                                                orb.ctypes.data_as(i32p), ore.ctypes.data_as(i32p),
                                                ob.ctypes.data_as(i32p), oe.ctypes.data_as(i32p),
                                                osk.ctypes.data_as(u8p), C.c_int64(cap), C.byref(n)))
        return [orb, ore, ob[:n.value].copy(), oe[:n.value].copy(), chars, osk[:n.value].copy()]


# --------------------------------------------------------------------------- BPETokenizer
class BPETokenizer:
    """vocab: list[bytes]; merges: list[(bytes,bytes)] or list[bytes] ("a b" text form);
    added_tokens: dict bytes->id or None."""

    def __init__(self, vocab, merges, added_tokens=None, unk_token="", fuse_unk=False, suffix_indicator="",
                 end_suffix="", byte_fallback=False, cache_capacity=20000):
        vb, ve, vc = pack_strings(vocab)
        self._keep = [vb, ve, vc]
        if len(merges) and isinstance(merges[0], (tuple, list)):
            lb, le, lc = pack_strings([m[0] for m in merges])
            rb_, re_, rc = pack_strings([m[1] for m in merges])
            right = (rb_.ctypes.data_as(i32p), re_.ctypes.data_as(i32p), _u8(rc)[1])
            self._keep += [rb_, re_, rc]
        else:
            lb, le, lc = pack_strings(merges)
            right = (None, None, None)
        self._keep += [lb, le, lc]
        if added_tokens:
            ab, ae, ac = pack_strings(list(added_tokens.keys()))
            ai = np.asarray(list(added_tokens.values()), dtype=np.int32)
            added = (ab.ctypes.data_as(i32p), ae.ctypes.data_as(i32p), _u8(ac)[1], ai.ctypes.data_as(i32p), len(ai))
            self._keep += [ab, ae, ac, ai]
        else:
            added = (None, None, None, None, 0)
        enc = lambda s: s.encode("utf-8") if isinstance(s, str) else bytes(s)
        unk, si, es = enc(unk_token), enc(suffix_indicator), enc(end_suffix)
        self._h = C.c_void_p()
        _chk(lib().orc_bpe_create(vb.ctypes.data_as(i32p), ve.ctypes.data_as(i32p), _u8(vc)[1], C.c_int64(len(vb)),
                                  lb.ctypes.data_as(i32p), le.ctypes.data_as(i32p), _u8(lc)[1],
                                  right[0], right[1], right[2], C.c_int64(len(lb)),
                                  added[0], added[1], added[2], added[3], C.c_int64(added[4]),
                                  unk, C.c_int64(len(unk)), int(fuse_unk), si, C.c_int64(len(si)),
                                  es, C.c_int64(len(es)), int(byte_fallback), C.c_int64(cache_capacity),
                                  C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_bpe_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    @property
    def tie_events(self):
        return int(lib().orc_bpe_tie_events(self._h))

    def clear_cache(self):
        lib().orc_bpe_clear_cache(self._h)

    def __call__(self, rb, re_, begins, ends, chars, cap=None):
        rb, prb = _i32(rb)
        re_, pre = _i32(re_)
        begins, pb = _i32(begins)
        ends, pe = _i32(ends)
        chars, pc = _u8(chars)
        B = len(rb)
        cap = len(chars) if cap is None else cap
        ob = np.zeros(B, np.int32)
        oe = np.zeros(B, np.int32)
        ids = np.zeros(max(cap, 1), np.int32)
        n = C.c_int64()
        _chk(lib().orc_bpe_run(self._h, prb, pre, C.c_int64(B), pb, pe, pc, ob.ctypes.data_as(i32p),
                               oe.ctypes.data_as(i32p), ids.ctypes.data_as(i32p), C.c_int64(cap), C.byref(n)))
        return ob, oe, ids[:n.value].copy()


# --------------------------------------------------------------------------- WordpieceTokenizer
class WordpieceTokenizer:
    def __init__(self, vocab, suffix_indicator="##", max_bytes_per_word=100):
        vb, ve, vc = pack_strings(vocab)
        si = suffix_indicator.encode() if isinstance(suffix_indicator, str) else bytes(suffix_indicator)
        self._h = C.c_void_p()
        _chk(lib().orc_wordpiece_create(vb.ctypes.data_as(i32p), ve.ctypes.data_as(i32p), _u8(vc)[1],
                                        C.c_int64(len(vb)), si, C.c_int64(len(si)), int(max_bytes_per_word),
                                        C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_wordpiece_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def __call__(self, rb, re_, begins, ends, chars, unk_id):
        rb, prb = _i32(rb)
        re_, pre = _i32(re_)
        begins, pb = _i32(begins)
        ends, pe = _i32(ends)
        chars, pc = _u8(chars)
        B, cap = len(rb), max(len(chars), 1)
        ob = np.zeros(B, np.int32)
        oe = np.zeros(B, np.int32)
        ids = np.zeros(cap, np.int32)
        n = C.c_int64()
        _chk(lib().orc_wordpiece_run(self._h, prb, pre, C.c_int64(B), pb, pe, pc, C.c_int32(int(unk_id)),
                                     ob.ctypes.data_as(i32p), oe.ctypes.data_as(i32p), ids.ctypes.data_as(i32p),
                                     C.c_int64(len(chars)), C.byref(n)))
        return ob, oe, ids[:n.value].copy()


# --------------------------------------------------------------------------- TrieTokenizer
class TrieTokenizer:
    def __init__(self, vocab, indices):
        vb, ve, vc = pack_strings(vocab)
        idx, pidx = _i32(indices)
        self._h = C.c_void_p()
        _chk(lib().orc_trie_tokenizer_create(vb.ctypes.data_as(i32p), ve.ctypes.data_as(i32p), _u8(vc)[1], C.c_int64(len(vb)),
                                             pidx, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_trie_tokenizer_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def __call__(self, rb, re_, begins, ends, chars):
        rb, prb = _i32(rb)
        re_, pre = _i32(re_)
        begins, pb = _i32(begins)
        ends, pe = _i32(ends)
        chars, pc = _u8(chars)
        B, cap = len(rb), max(len(chars), 1)
        ob, oe, ids = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(cap, np.int32)
        n = C.c_int64()
        _chk(lib().orc_trie_tokenizer_run(self._h, prb, pre, C.c_int64(B), pb, pe, pc, ob.ctypes.data_as(i32p),
                                          oe.ctypes.data_as(i32p), ids.ctypes.data_as(i32p), C.c_int64(cap), C.byref(n)))
        return ob, oe, ids[:n.value].copy()


# --------------------------------------------------------------------------- VocabEncoder
class VocabEncoder:
    def __init__(self, keys, values):
        values = np.ascontiguousarray(values)
        assert values.dtype in (np.int32, np.int64)
        kb, ke, kc = pack_strings(keys)
        self.dtype = values.dtype
        self._h = C.c_void_p()
        _chk(lib().orc_vocab_encoder_create(kb.ctypes.data_as(i32p), ke.ctypes.data_as(i32p), _u8(kc)[1],
                                            values.ctypes.data_as(C.c_void_p), C.c_int64(len(kb)),
                                            int(values.dtype.itemsize), C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_vocab_encoder_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def __call__(self, begins, ends, chars, default):
        begins, pb = _i32(begins)
        ends, pe = _i32(ends)
        chars, pc = _u8(chars)
        d = np.asarray([default], dtype=self.dtype)
        out = np.zeros(len(begins), self.dtype)
        _chk(lib().orc_vocab_encoder_run(self._h, pb, pe, pc, C.c_int64(len(begins)),
                                         d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out


# --------------------------------------------------------------------------- RaggedToDense
def ragged_to_dense(begins, ends, data, target_dim, default, pad_right=True, pad_max_length=False):
    begins, pb = _i32(begins)
    ends, pe = _i32(ends)
    data = np.ascontiguousarray(data)
    inner = int(np.prod(data.shape[1:])) if data.ndim > 1 else 1
    B, T = len(begins), int(target_dim)
    d = np.asarray([default], dtype=data.dtype)
    out = np.zeros((B, T) + tuple(data.shape[1:]), data.dtype)
    mask = np.zeros((B, T) + tuple(data.shape[1:]), np.uint8)
    _chk(lib().orc_ragged_to_dense(pb, pe, C.c_int64(B), data.ctypes.data_as(C.c_void_p),
                                   C.c_int64(data.shape[0] if data.ndim else 0), int(data.dtype.itemsize),
                                   C.c_int64(inner), C.c_int32(T), d.ctypes.data_as(C.c_void_p), int(pad_right),
                                   int(pad_max_length), out.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(u8p)))
    return out, mask.astype(bool)


# --------------------------------------------------------------------------- UTF8Validate
def utf8_validate(begins, ends, chars, replace_mode):
    (b, pb), (e, pe), (c, pc) = _i32(begins), _i32(ends), _u8(chars)
    cap = 3 * len(c) + 3
    ob, oe, oc = np.zeros(len(b), np.int32), np.zeros(len(b), np.int32), np.zeros(cap, np.uint8)
    n = C.c_int64()
    _chk(lib().orc_utf8_validate(pb, pe, pc, C.c_int64(len(b)),
                                 int(bool(replace_mode)), ob.ctypes.data_as(i32p), oe.ctypes.data_as(i32p),
                                 oc.ctypes.data_as(u8p), C.c_int64(cap), C.byref(n)))
    first = int(b[0]) if len(b) else 0
    return ob, oe, oc[:first + n.value].copy()


# --------------------------------------------------------------------------- Truncate / CombineSegments
def truncate(pairs, max_length, side="right", mode="longest_first"):
    """pairs: [(begins, ends)] or [(b0, e0), (b1, e1)] -> list of truncated (begins, ends)."""
    arrs = [(np.array(b, dtype=np.int32), np.array(e, dtype=np.int32)) for b, e in pairs]
    n = len(arrs[0][0])
    ptr = [a.ctypes.data_as(i32p) for pair in arrs for a in pair] + [None, None]
    _chk(lib().orc_truncate(len(arrs), ptr[0], ptr[1], ptr[2], ptr[3], C.c_int64(n), C.c_int32(int(max_length)),
                            side.encode(), mode.encode() if len(arrs) == 2 else None))
    return arrs


def combine_segments(segments, seg_ids):
    """segments: list of (begins, ends, data) i32 -> (begins, ends, data, ids)."""
    segs = [tuple(np.ascontiguousarray(x, dtype=np.int32) for x in s) for s in segments]
    k = len(segs)
    max_rows = max(len(s[0]) for s in segs)
    cap = sum((max_rows if len(s[0]) == 1 else 1) * len(s[2]) for s in segs) + 1
    PB = (i32p * k)(*[s[0].ctypes.data_as(i32p) for s in segs])
    PE = (i32p * k)(*[s[1].ctypes.data_as(i32p) for s in segs])
    PD = (i32p * k)(*[s[2].ctypes.data_as(i32p) for s in segs])
    NR = (C.c_int64 * k)(*[len(s[0]) for s in segs])
    ids = np.ascontiguousarray(seg_ids, dtype=np.int32)
    ob, oe = np.zeros(max_rows, np.int32), np.zeros(max_rows, np.int32)
    od, oi = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    n = C.c_int64()
    _chk(lib().orc_combine_segments(k, PB, PE, PD, NR, ids.ctypes.data_as(i32p), C.c_int64(max_rows),
                                    ob.ctypes.data_as(i32p), oe.ctypes.data_as(i32p), od.ctypes.data_as(i32p),
                                    oi.ctypes.data_as(i32p), C.c_int64(cap), C.byref(n)))
    return ob, oe, od[:n.value].copy(), oi[:n.value].copy()


# --------------------------------------------------------------------------- detokenize trio
def vocab_decoder(ids, vocab, skip_tokens=()):
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    B, S = ids.shape
    vb, ve, vc = pack_strings(vocab)
    skip = np.asarray(list(skip_tokens), dtype=np.int32)
    Sp = max(S, 1)
    maxlen = int((ve - vb).max()) if len(vb) else 0
    cap = max(B * S * maxlen, 1)
    orb = np.zeros(B, np.int32)
    ore = np.zeros(B, np.int32)
    ob = np.zeros(B * Sp, np.int32)
    oe = np.zeros(B * Sp, np.int32)
    oc = np.zeros(cap, np.uint8)
    n = C.c_int64()
    _chk(lib().orc_vocab_decoder(ids.ctypes.data_as(i32p), C.c_int64(B), C.c_int64(S), vb.ctypes.data_as(i32p),
                                 ve.ctypes.data_as(i32p), _u8(vc)[1], C.c_int64(len(vb)),
                                 skip.ctypes.data_as(i32p), C.c_int64(len(skip)),
                                 orb.ctypes.data_as(i32p), ore.ctypes.data_as(i32p), ob.ctypes.data_as(i32p),
                                 oe.ctypes.data_as(i32p), oc.ctypes.data_as(u8p), C.c_int64(cap), C.byref(n)))
    return orb, ore, ob, oe, oc[:n.value].copy()


def byte_fallback(begins, ends, chars):
    begins, pb = _i32(begins)
    ends, pe = _i32(ends)
    chars, pc = _u8(chars)
    N = len(begins)
    ob = np.zeros(N, np.int32)
    oe = np.zeros(N, np.int32)
    oc = np.zeros(max(len(chars), 1), np.uint8)
    n = C.c_int64()
    _chk(lib().orc_byte_fallback(pb, pe, pc, C.c_int64(N), ob.ctypes.data_as(i32p), oe.ctypes.data_as(i32p),
                                 oc.ctypes.data_as(u8p), C.byref(n)))
    return ob, oe, oc[:n.value].copy()


def fuze(rb, re_, begins, ends):
    rb, prb = _i32(rb)
    re_, pre = _i32(re_)
    begins, pb = _i32(begins)
    ends, pe = _i32(ends)
    B = len(rb)
    ob = np.zeros(B, np.int32)
    oe = np.zeros(B, np.int32)
    _chk(lib().orc_fuze(prb, pre, C.c_int64(B), pb, pe, C.c_int64(len(begins)), ob.ctypes.data_as(i32p),
                        oe.ctypes.data_as(i32p)))
    return ob, oe
