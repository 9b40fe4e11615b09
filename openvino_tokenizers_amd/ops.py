"""Host-side mirror of the reference's custom ops for the tokenizer hot path.

Each class carries the reference op's name, attributes (`visit_attributes` names) and
`evaluate(inputs) -> outputs` contract with the reference's input order,
so a test written against the reference's op reads the same here.  Constant inputs (vocab,
merges, pattern ...) are consumed on the first `evaluate` -- like the reference's lazy
`call_once` initialisation -- and compiled into device tables by the C-ABI library.

Data inputs may be numpy arrays (host memory: staged over PCIe by the library) or torch CUDA
tensors (device memory: zero copies; outputs are torch tensors on the same device and the
kernels run on torch's current stream).  Everything is computed by libovtk_amd.so on the GPU;
there is no CPU implementation behind these classes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_NP = {"i32": np.int32, "u8": np.uint8, "i64": np.int64, "bool": np.uint8}


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "device")


class _Mem:
    """Where the data tensors of one call live + how to allocate outputs next to them."""

    def __init__(self, sample):
        self.torch = _is_torch(sample) and sample.device.type == "cuda"
        if self.torch:
            import torch
            self.t = torch
            self.device = sample.device
            self.mem = L.MEM_DEVICE
            self.stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        else:
            self.mem = L.MEM_HOST
            self.stream = C.c_void_p(0)
        self.keep = []

    def inp(self, x, kind):
        """-> (array, pointer) of a data input in this call's memory space."""
        if self.torch:
            dt = {"i32": self.t.int32, "u8": self.t.uint8, "i64": self.t.int64, "bool": self.t.uint8}[kind]
            if not _is_torch(x):
                x = self.t.as_tensor(np.ascontiguousarray(x, dtype=_NP[kind]), device=self.device)
            if x.dtype == self.t.bool:
                x = x.view(self.t.uint8)
            x = x.to(device=self.device, dtype=dt).contiguous()
            self.keep.append(x)
            return x, C.c_void_p(x.data_ptr())
        if _is_torch(x):
            x = x.cpu().numpy()
        a = np.ascontiguousarray(x, dtype=_NP[kind])
        self.keep.append(a)
        return a, C.c_void_p(a.ctypes.data)

    def alloc(self, n, kind):
        n = max(int(n), 1)
        if self.torch:
            dt = {"i32": self.t.int32, "u8": self.t.uint8, "i64": self.t.int64, "bool": self.t.uint8}[kind]
            x = self.t.empty(n, dtype=dt, device=self.device)
            return x, C.c_void_p(x.data_ptr())
        a = np.empty(n, dtype=_NP[kind])
        return a, C.c_void_p(a.ctypes.data)


def _host(x, dtype):
    if _is_torch(x):
        x = x.cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


def _strings_struct(begins, ends, chars, keep):
    b, e, c = _host(begins, np.int32), _host(ends, np.int32), _host(chars, np.uint8)
    keep += [b, e, c]
    return L.Strings(b.ctypes.data, e.ctypes.data, c.ctypes.data if c.size else None, len(b), len(c))


def _ragged_in(m: _Mem, inputs):
    rb, prb = m.inp(inputs[0], "i32")
    re_, pre = m.inp(inputs[1], "i32")
    b, pb = m.inp(inputs[2], "i32")
    e, pe = m.inp(inputs[3], "i32")
    c, pc = m.inp(inputs[4], "u8")
    rs = L.RaggedStrings(prb, pre, len(rb), L.Strings(pb, pe, pc, len(b), len(c)))
    return rs, (rb, re_, b, e, c)


def _bytes_of(x):
    if isinstance(x, str):
        return x.encode("utf-8")
    if isinstance(x, (bytes, bytearray)):
        return bytes(x)
    return bytes(_host(x, np.uint8))


class _Op:
    def __init__(self, device=0, lib=None):
        self.device = int(device)
        self._lib = lib if lib is not None else L.load()
        self._h = C.c_void_p()

    def _chk(self, rc):
        L.check(self._lib, rc)


class RegexSplit(_Op):
    """Reference: src/regex_split.cpp (evaluate :124-324).  Inputs: ragged_begins, ragged_ends, begins, ends,
    chars, [skips], pattern.  Outputs: ragged_begins, ragged_ends, begins, ends, chars (the input tensor), [skips]."""

    def __init__(self, behaviour="remove", invert=False, max_splits=-1, device=0, lib=None):
        super().__init__(device, lib)
        self.behaviour, self.invert, self.max_splits = behaviour, bool(invert), int(max_splits)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ovtk_regex_split_destroy(self._h)
            self._h = None

    def _ensure(self, pattern):
        if self._h:
            return
        pat = _bytes_of(pattern)
        p = L.RegexSplitParams(pat, len(pat), self.behaviour.encode(), int(self.invert), self.max_splits, self.device)
        self._chk(self._lib.ovtk_regex_split_create(C.byref(p), C.byref(self._h)))

    def _legacy_skips(self, inputs):
        """The 9-input form of old IRs (regex_split.cpp:102, 164-179, 235-238): inputs 6-8 are a string tensor of "skip
        tokens"; a string that EQUALS one of them passes through unsplit.  That is a set-membership flag per string -- the
        VocabEncoder kernel with the skip tokens as keys (value 1, default 0) --, and from there on the op is the 7-input
        form; the flags do not become an output (the legacy form has five)."""
        if not hasattr(self, "_skip_set"):
            n_keys = len(_host(inputs[6], np.int32))
            self._skip_set = VocabEncoder(self.device, self._lib) if n_keys else None
            self._skip_consts = list(inputs[6:9]) + [np.ones(n_keys, np.int32), np.zeros((), np.int32)]
        if self._skip_set is None:
            return None
        (flags,) = self._skip_set.evaluate(list(inputs[2:5]) + self._skip_consts)
        return flags != 0

    def evaluate(self, inputs):
        if len(inputs) == 9:
            skips = self._legacy_skips(inputs)
            if skips is None:
                return self.evaluate(list(inputs[:6]))
            return self.evaluate(list(inputs[:5]) + [skips, inputs[5]])[:5]
        if len(inputs) not in (6, 7):
            raise L.OvtkError(L.E_ARG, f"Incorrect number of inputs passed to RegexSplit: {len(inputs)}; try to reconvert tokenizer "
                                       "with newer version of OpenVINO Tokenizers")
        has_skips = len(inputs) == 7
        self._ensure(inputs[5 + has_skips])
        m = _Mem(inputs[4])
        rs, (rb, re_, b, e, c) = _ragged_in(m, inputs)
        skips, pskips = (m.inp(inputs[5], "bool") if has_skips else (None, None))
        cap = len(c) + len(b)
        orb, porb = m.alloc(len(rb), "i32")
        ore, pore = m.alloc(len(rb), "i32")
        ob, pob = m.alloc(cap, "i32")
        oe, poe = m.alloc(cap, "i32")
        osk, posk = (m.alloc(cap, "bool") if has_skips else (None, None))
        out = L.RaggedStringsOut(porb, pore, 0, pob, poe, posk, cap, 0)
        self._chk(self._lib.ovtk_regex_split_run(self._h, C.byref(rs), pskips, C.byref(out), m.mem, m.stream))
        if out.n < 0:  # regex_split.cpp:129-143: shape {1} ragged dims, string tensors are the inputs
            res = [orb[:1], ore[:1], b, e, c]
            if has_skips:
                res.append(skips)
            return res
        res = [orb[:out.n_rows], ore[:out.n_rows], ob[:out.n], oe[:out.n], c]
        if has_skips:
            res.append(osk[:out.n])
        return res


class SpecialTokensSplit(_Op):
    """Reference: src/special_tokens_split.cpp (evaluate :61-162).  Inputs: ragged_begins, ragged_ends, begins, ends,
    chars, [skips], pattern.  Outputs: ragged_begins, ragged_ends, begins, ends, chars (the input tensor), skips."""

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ovtk_special_tokens_split_destroy(self._h)
            self._h = None

    def _ensure(self, pattern):
        if self._h:
            return
        pat = _bytes_of(pattern)
        self._chk(self._lib.ovtk_special_tokens_split_create(pat, C.c_int64(len(pat)), self.device, C.byref(self._h)))

    def evaluate(self, inputs):
        if len(inputs) not in (6, 7):
            raise L.OvtkError(L.E_ARG, f"Incorrect number of inputs passed to SpecialTokensSplit: {len(inputs)}; try to "
                                       f"reconvert tokenizer with newer version of OpenVINO Tokenizers")
        has_skips = len(inputs) == 7
        self._ensure(inputs[5 + has_skips])
        m = _Mem(inputs[4])
        rs, (rb, re_, b, e, c) = _ragged_in(m, inputs)
        _, pskips = (m.inp(inputs[5], "bool") if has_skips else (None, None))
        cap = len(c) + len(b)  # the reference sizes to n_chars (:88-92); + n for skipped empty strings
        orb, porb = m.alloc(len(rb), "i32")
        ore, pore = m.alloc(len(rb), "i32")
        ob, pob = m.alloc(cap, "i32")
        oe, poe = m.alloc(cap, "i32")
        osk, posk = m.alloc(cap, "bool")
        out = L.RaggedStringsOut(porb, pore, 0, pob, poe, posk, cap, 0)
        self._chk(self._lib.ovtk_special_tokens_split_run(self._h, C.byref(rs), pskips, C.byref(out), m.mem, m.stream))
        return [orb[:out.n_rows], ore[:out.n_rows], ob[:out.n], oe[:out.n], c, osk[:out.n]]


class BPETokenizer(_Op):
    """Reference: src/bpe_tokenizer.cpp (evaluate :47-164).  11/14/15/18 inputs: ragged strings (5), vocab (3),
    merges (3, or 3 + 3 for left/right halves), [added tokens (3) + ids (1)].  Outputs: begins, ends, ids."""

    def __init__(self, unk_token="", fuse_unk=False, suffix_indicator="", end_suffix="", byte_fallback=False,
                 cache_capacity=20000, device=0, lib=None, memo_store=0, memo_learn=0):
        super().__init__(device, lib)
        self.memo_store, self.memo_learn = int(memo_store), int(memo_learn)   # (include/ovtk_amd.h ovtk_bpe_params: not attributes of the reference's op)
        self.unk_token, self.fuse_unk = unk_token, bool(fuse_unk)
        self.suffix_indicator, self.end_suffix = suffix_indicator, end_suffix
        self.byte_fallback, self.cache_capacity = bool(byte_fallback), int(cache_capacity)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ovtk_bpe_destroy(self._h)
            self._h = None

    def _ensure(self, inputs):
        if self._h:
            return
        n = len(inputs)
        if n not in (11, 14, 15, 18):
            raise L.OvtkError(L.E_ARG, "Incorrect number of inputs passed to BPETokenizer")
        keep = []
        vocab = _strings_struct(*inputs[5:8], keep)
        merges = _strings_struct(*inputs[8:11], keep)
        right = _strings_struct(*inputs[11:14], keep) if n in (14, 18) else L.Strings(None, None, None, 0, 0)
        if n in (15, 18):
            added = _strings_struct(*inputs[n - 4:n - 1], keep)
            ids = _host(inputs[n - 1], np.int32)
            keep.append(ids)
            pids = ids.ctypes.data
        else:
            added, pids = L.Strings(None, None, None, 0, 0), None
        unk, si, es = _bytes_of(self.unk_token), _bytes_of(self.suffix_indicator), _bytes_of(self.end_suffix)
        p = L.BpeParams(vocab, merges, right, added, pids, unk, len(unk), int(self.fuse_unk), si, len(si), es, len(es),
                        int(self.byte_fallback), self.cache_capacity, self.device, self.memo_store, self.memo_learn)
        self._chk(self._lib.ovtk_bpe_create(C.byref(p), C.byref(self._h)))

    def evaluate(self, inputs, ids_capacity=None):
        """ids_capacity: size of the ids buffer; default = number of input chars, as the reference sizes it
        (bpe_tokenizer.cpp:135) -- only an end_suffix model can need more."""
        self._ensure(inputs)
        m = _Mem(inputs[4])
        rs, (rb, _, _, _, c) = _ragged_in(m, inputs)
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c) if ids_capacity is None else int(ids_capacity)
        ids, pids = m.alloc(cap, "i32")
        out = L.RaggedI32Out(pob, poe, pids, cap, 0, 0)
        self._chk(self._lib.ovtk_bpe_run(self._h, C.byref(rs), C.byref(out), m.mem, m.stream))
        return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]


class FusedSplitBPE:
    """RegexSplit -> BPETokenizer in one launch (ovtk_encode_run): the same result as chaining the two ops
    (what tokenizer_pipeline.py:1613-1631 builds for byte-level BPE models) without materialising the pieces."""

    def __init__(self, split: RegexSplit, bpe: BPETokenizer):
        self.split, self.bpe = split, bpe

    def evaluate(self, split_inputs, bpe_constant_inputs):
        """split_inputs: the 6/7 inputs of RegexSplit; bpe_constant_inputs: inputs 5.. of BPETokenizer."""
        has_skips = len(split_inputs) == 7
        self.split._ensure(split_inputs[5 + has_skips])
        self.bpe._ensure(list(split_inputs[:5]) + list(bpe_constant_inputs))
        m = _Mem(split_inputs[4])
        rs, (rb, _, _, _, c) = _ragged_in(m, split_inputs)
        _, pskips = (m.inp(split_inputs[5], "bool") if has_skips else (None, None))
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c)
        ids, pids = m.alloc(cap, "i32")
        out = L.RaggedI32Out(pob, poe, pids, cap, 0, 0)
        lib = self.bpe._lib
        L.check(lib, lib.ovtk_encode_run(self.split._h, self.bpe._h, C.byref(rs), pskips, C.byref(out), m.mem, m.stream))
        return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]

    def enqueue(self, split_inputs, bpe_constant_inputs):
        """evaluate() in two halves for CUDA tensors (ovtk_encode_enqueue / ovtk_encode_finish): launches the kernels
        and returns a ticket; `ticket()` waits for them and returns evaluate()'s outputs.  The host is free in between
        (to launch the next batch, or a ShardExchange step)."""
        has_skips = len(split_inputs) == 7
        self.split._ensure(split_inputs[5 + has_skips])
        self.bpe._ensure(list(split_inputs[:5]) + list(bpe_constant_inputs))
        m = _Mem(split_inputs[4])
        if not m.torch:
            raise L.OvtkError(L.E_ARG, "enqueue() needs device-resident (torch CUDA) inputs")
        rs, (rb, _, _, _, c) = _ragged_in(m, split_inputs)
        _, pskips = (m.inp(split_inputs[5], "bool") if has_skips else (None, None))
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c)
        ids, pids = m.alloc(cap, "i32")
        out = L.RaggedI32Out(pob, poe, pids, cap, 0, 0)
        lib = self.bpe._lib
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_enqueue(self.split._h, self.bpe._h, C.byref(rs), pskips, C.byref(out), m.stream, C.byref(pending)))

        def ticket(_keep=m):
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(out)))
            return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]
        return ticket

    def enqueue_host(self, split_inputs, bpe_constant_inputs, outputs=None, stream=None):
        """The two halves for HOST arrays (ovtk_encode_enqueue_host): numpy inputs -- pinned ones (e.g. views of
        torch.empty(..., pin_memory=True)) make the copies asynchronous --, `outputs` = (begins, ends, ids) numpy arrays to
        fill (allocated here when None), `stream` a HIP stream handle (int) or None.  `ticket()` -> [begins, ends, ids]."""
        has_skips = len(split_inputs) == 7
        self.split._ensure(split_inputs[5 + has_skips])
        self.bpe._ensure(list(split_inputs[:5]) + list(bpe_constant_inputs))
        m = _Mem(np.empty(0, np.uint8))
        rs, (rb, _, _, _, c) = _ragged_in(m, split_inputs)
        _, pskips = (m.inp(split_inputs[5], "bool") if has_skips else (None, None))
        if outputs is None:
            outputs = (np.empty(max(len(rb), 1), np.int32), np.empty(max(len(rb), 1), np.int32), np.empty(max(len(c), 1), np.int32))
        ob, oe, ids = outputs
        out = L.RaggedI32Out(ob.ctypes.data, oe.ctypes.data, ids.ctypes.data, len(ids), 0, 0)
        lib = self.bpe._lib
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_enqueue_host(self.split._h, self.bpe._h, C.byref(rs), pskips, C.byref(out),
                                                  C.c_void_p(stream or 0), C.byref(pending)))

        def ticket(_keep=(m, outputs)):
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(out)))
            return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]
        return ticket

    def enqueue_packed(self, packed, pattern, bpe_constant_inputs, outputs=None, stream=None):
        """StringTensorUnpack -> RegexSplit -> BPETokenizer from the PACKED u8 form of a batch of strings, [i32 n][i32 begin_0]
        [i32 end_i x n][bytes] (src/utils.cpp:18-29), in host memory (ovtk_encode_enqueue_packed): one buffer crosses PCIe,
        every string is a row.  `outputs` = (begins, ends, ids) numpy arrays to fill -- pinned ones are written by the kernels
        themselves --, allocated here when None.  `ticket()` -> [begins, ends, ids]."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        if len(packed) < 4:
            raise L.OvtkError(L.E_ARG, "Incorrect packed string tensor format: no batch size in the packed string tensor")
        n = int(packed[:4].view(np.int32)[0])
        if n < 0 or len(packed) < 8 + 4 * n:
            raise L.OvtkError(L.E_ARG, "Incorrect packed string tensor format: the packed string tensor must contain first string "
                                       "offset and end indices")
        self.split._ensure(pattern)
        z = np.zeros(1, np.int32)
        self.bpe._ensure([z, z, z, z, np.zeros(0, np.uint8)] + list(bpe_constant_inputs))
        if outputs is None:
            n_chars = max(len(packed) - 8 - 4 * n, 1)
            outputs = (np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32), np.empty(n_chars, np.int32))
        ob, oe, ids = outputs
        out = L.RaggedI32Out(ob.ctypes.data, oe.ctypes.data, ids.ctypes.data, len(ids), 0, 0)
        lib = self.bpe._lib
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_enqueue_packed(self.split._h, self.bpe._h, C.c_void_p(packed.ctypes.data), C.c_int64(len(packed)),
                                                    C.byref(out), L.MEM_HOST, C.c_void_p(stream or 0), C.byref(pending)))

        def ticket(_keep=(packed, outputs)):
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(out)))
            return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]
        return ticket

    def enqueue_wire(self, split_inputs, bpe_constant_inputs, wire, max_rows, pad_ids, id_bytes, stream=None):
        """The encode straight into a ShardExchange send wire (ovtk_encode_enqueue_wire): CUDA inputs, `wire` a uint8 CUDA
        tensor of ovtk_shard_wire_bytes(pad_ids) bytes with `max_rows` row slots.  compact_kernel writes the header, the
        row ends and the ids narrowed to `id_bytes`, so no ragged int32 tensor and no pack kernel are in between.
        `ticket()` waits and returns the shard's id count (more than pad_ids: the wire was cut, encode again into a
        larger one)."""
        has_skips = len(split_inputs) == 7
        self.split._ensure(split_inputs[5 + has_skips])
        self.bpe._ensure(list(split_inputs[:5]) + list(bpe_constant_inputs))
        m = _Mem(split_inputs[4])
        lib = self.bpe._lib
        if not m.torch and not hasattr(lib, "ovtk_emulator_build"):   # the emulator build's device memory IS host memory (tests)
            raise L.OvtkError(L.E_ARG, "enqueue_wire() needs device-resident (torch CUDA) inputs")
        rs, _ = _ragged_in(m, split_inputs)
        _, pskips = (m.inp(split_inputs[5], "bool") if has_skips else (None, None))
        pending = C.c_void_p()
        st = C.c_void_p(stream) if isinstance(stream, int) else (stream if stream is not None else m.stream)
        L.check(lib, lib.ovtk_encode_enqueue_wire(self.split._h, self.bpe._h, C.byref(rs), pskips, C.c_void_p(wire.data_ptr()),
                                                  C.c_int64(max_rows), C.c_int64(pad_ids), int(id_bytes), st, C.byref(pending)))
        out = L.RaggedI32Out(None, None, None, 0, 0, 0)

        def ticket(_keep=(m, wire)):
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(out)))
            return int(out.n_data)
        return ticket


class FusedSpecialSplitBPE:
    """SpecialTokensSplit -> RegexSplit -> BPETokenizer in one call (ovtk_encode_special_run / _enqueue): the sub-graph
    tokenizer_pipeline.py:1613-1636 builds for a byte-level BPE model, the split strings and skip flags never leaving the device.
    Same result as chaining SpecialTokensSplit.evaluate and FusedSplitBPE.evaluate."""

    def __init__(self, special: SpecialTokensSplit, split: RegexSplit, bpe: BPETokenizer):
        self.special, self.split, self.bpe = special, split, bpe

    def _prep(self, special_inputs, split_pattern, bpe_constant_inputs):
        has_skips = len(special_inputs) == 7
        self.special._ensure(special_inputs[5 + has_skips])
        self.split._ensure(split_pattern)
        self.bpe._ensure(list(special_inputs[:5]) + list(bpe_constant_inputs))
        m = _Mem(special_inputs[4])
        rs, (rb, _, _, _, c) = _ragged_in(m, special_inputs)
        _, pskips = (m.inp(special_inputs[5], "bool") if has_skips else (None, None))
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c)
        ids, pids = m.alloc(cap, "i32")
        return m, rs, pskips, (ob, oe, ids), L.RaggedI32Out(pob, poe, pids, cap, 0, 0)

    def evaluate(self, special_inputs, split_pattern, bpe_constant_inputs):
        """special_inputs: the 6/7 inputs of SpecialTokensSplit; split_pattern: RegexSplit's pattern; bpe_constant_inputs: inputs 5..
        of BPETokenizer."""
        m, rs, pskips, (ob, oe, ids), out = self._prep(special_inputs, split_pattern, bpe_constant_inputs)
        lib = self.bpe._lib
        L.check(lib, lib.ovtk_encode_special_run(self.special._h, self.split._h, self.bpe._h, C.byref(rs), pskips, C.byref(out), m.mem, m.stream))
        return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]

    def enqueue(self, special_inputs, split_pattern, bpe_constant_inputs):
        m, rs, pskips, (ob, oe, ids), out = self._prep(special_inputs, split_pattern, bpe_constant_inputs)
        if not m.torch:
            raise L.OvtkError(L.E_ARG, "enqueue() needs device-resident (torch CUDA) inputs")
        lib = self.bpe._lib
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_encode_special_enqueue(self.special._h, self.split._h, self.bpe._h, C.byref(rs), pskips, C.byref(out), m.stream,
                                                     C.byref(pending)))

        def ticket(_keep=m):
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(out)))
            return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]
        return ticket


class FusedEncodeDense:
    """[SpecialTokensSplit ->] RegexSplit -> BPETokenizer -> Truncate -> CombineSegments (constant ids in front / behind) ->
    RaggedToDense x 2 in one call (ovtk_encode_dense_enqueue / _finish): input_ids [B, T] and attention_mask [B, T] of a converted
    byte-level BPE tokenizer (tokenizer_pipeline.py:1613-1636 + TruncationStep / CombineSegmentsStep / PaddingStep) straight from
    the encode's last pass -- the ragged ids tensor never exists.  Device tensors (torch CUDA; host arrays with the emulator build)."""

    def __init__(self, split: RegexSplit, bpe: BPETokenizer, special: SpecialTokensSplit = None, max_length=2**31 - 1, trunc_side="right",
                 pad_right=True, pad_value=0, prefix=(), suffix=()):
        self.split, self.bpe, self.special = split, bpe, special
        self.max_length, self.trunc_left, self.pad_right, self.pad_value = int(max_length), trunc_side == "left", bool(pad_right), int(pad_value)
        self.prefix, self.suffix = np.asarray(list(prefix), np.int32), np.asarray(list(suffix), np.int32)

    def enqueue(self, ragged_inputs, split_pattern, bpe_constant_inputs, special_pattern=None, target_dim=None, row_capacity=None, stream=None):
        """ragged_inputs: ragged_begins, ragged_ends, begins, ends, chars [, skips].  row_capacity: cells per row the outputs are
        allocated with (default: max_length + the constant ids, or the longest string's bytes if that is less)."""
        has_skips = len(ragged_inputs) == 6
        lib = self.bpe._lib
        if self.special is not None:
            self.special._ensure(special_pattern)
        self.split._ensure(split_pattern)
        self.bpe._ensure(list(ragged_inputs[:5]) + list(bpe_constant_inputs))
        m = _Mem(ragged_inputs[4])
        if not m.torch and not hasattr(lib, "ovtk_emulator_build"):   # the emulator build's device memory IS host memory (tests)
            raise L.OvtkError(L.E_ARG, "FusedEncodeDense needs device-resident (torch CUDA) inputs")
        rs, (rb, _, b, e, c) = _ragged_in(m, ragged_inputs)
        _, pskips = (m.inp(ragged_inputs[5], "bool") if has_skips else (None, None))
        rows = len(rb)
        extra = len(self.prefix) + len(self.suffix)
        if row_capacity is None:
            row_capacity = (int(target_dim) if target_dim is not None else min(self.max_length, max(len(c), 1)) + extra)
        cap = max(rows * int(row_capacity), 1)
        ids, pids = m.alloc(cap, "i32")
        mask, pmask = m.alloc(cap, "bool")
        p = L.DenseParams(C.c_int32(min(self.max_length, 2**31 - 1)), int(self.trunc_left), int(self.pad_right), self.pad_value,
                          -1 if target_dim is None else int(target_dim), self.prefix.ctypes.data_as(C.c_void_p), len(self.prefix),
                          self.suffix.ctypes.data_as(C.c_void_p), len(self.suffix))
        pending = C.c_void_p()
        st = C.c_void_p(stream) if isinstance(stream, int) else (stream if stream is not None else m.stream)
        L.check(lib, lib.ovtk_encode_dense_enqueue(self.special._h if self.special is not None else None, self.split._h, self.bpe._h, C.byref(rs),
                                                   pskips, C.byref(p), pids, pmask, C.c_int64(cap), st, C.byref(pending)))

        def ticket(_keep=(m, p)):
            width, n_ids = C.c_int32(0), C.c_int64(0)
            L.check(lib, lib.ovtk_encode_dense_finish(pending, C.byref(width), C.byref(n_ids)))
            n = rows * int(width.value)
            shape = (rows, int(width.value))
            if m.torch:
                return [ids[:n].reshape(shape), mask[:n].reshape(shape).bool()]
            return [ids[:n].reshape(shape), mask[:n].reshape(shape).astype(bool)]
        return ticket

    def evaluate(self, *args, **kw):
        return self.enqueue(*args, **kw)()


class WordpieceTokenizer(_Op):
    """Reference: src/wordpiece_tokenizer.cpp (evaluate :49-133).  Inputs: ragged strings (5), vocab (3),
    unk_token_id (i32 scalar, read every call).  Outputs: begins, ends, ids."""

    def __init__(self, suffix_indicator="##", max_bytes_per_word=100, device=0, lib=None):
        super().__init__(device, lib)
        self.suffix_indicator, self.max_bytes_per_word = suffix_indicator, int(max_bytes_per_word)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ovtk_wordpiece_destroy(self._h)
            self._h = None

    def _ensure(self, inputs):
        if self._h:
            return
        keep = []
        vocab = _strings_struct(*inputs[5:8], keep)
        si = _bytes_of(self.suffix_indicator)
        p = L.WordpieceParams(vocab, si, len(si), self.max_bytes_per_word, self.device)
        self._chk(self._lib.ovtk_wordpiece_create(C.byref(p), C.byref(self._h)))

    def evaluate(self, inputs, ids_capacity=None):
        if len(inputs) != 9:
            raise L.OvtkError(L.E_ARG, f"Incorrect number of inputs passed to WordpieceTokenizer: {len(inputs)}")
        self._ensure(inputs)
        unk = int(np.asarray(_host(inputs[8], np.int32)).reshape(-1)[0])
        m = _Mem(inputs[4])
        rs, (rb, _, _, _, c) = _ragged_in(m, inputs)
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c) if ids_capacity is None else int(ids_capacity)  # wordpiece_tokenizer.cpp:85
        ids, pids = m.alloc(cap, "i32")
        out = L.RaggedI32Out(pob, poe, pids, cap, 0, 0)
        self._chk(self._lib.ovtk_wordpiece_run(self._h, C.byref(rs), C.c_int32(unk), C.byref(out), m.mem, m.stream))
        return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]


class VocabEncoder(_Op):
    """Reference: src/vocab_encoder.cpp (evaluate_impl :55-94).  Inputs: strings (3), key strings (3), values
    i32/i64[V], default scalar.  Output: values[N]."""

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ovtk_vocab_encoder_destroy(self._h)
            self._h = None

    def _ensure(self, inputs):
        if self._h:
            return
        keep = []
        keys = _strings_struct(*inputs[3:6], keep)
        values = inputs[6].cpu().numpy() if _is_torch(inputs[6]) else np.asarray(inputs[6])
        if values.dtype not in (np.int32, np.int64):  # vocab_encoder.cpp:38-53
            raise L.OvtkError(L.E_ARG, f"VocabEncoder: unsupported element type: {values.dtype}")
        values = np.ascontiguousarray(values)
        self.dtype = values.dtype
        p = L.VocabEncoderParams(keys, values.ctypes.data, values.dtype.itemsize, self.device)
        self._chk(self._lib.ovtk_vocab_encoder_create(C.byref(p), C.byref(self._h)))

    def evaluate(self, inputs):
        if len(inputs) != 8:
            raise L.OvtkError(L.E_ARG, f"Incorrect number of inputs passed to VocabEncoder: {len(inputs)}")
        self._ensure(inputs)
        m = _Mem(inputs[2])
        b, pb = m.inp(inputs[0], "i32")
        e, pe = m.inp(inputs[1], "i32")
        c, pc = m.inp(inputs[2], "u8")
        kind = "i32" if self.dtype == np.int32 else "i64"
        dflt = np.asarray(_host(inputs[7], self.dtype)).reshape(-1)[:1].copy()
        out, pout = m.alloc(len(b), kind)
        s = L.Strings(pb, pe, pc, len(b), len(c))
        self._chk(self._lib.ovtk_vocab_encoder_run(self._h, C.byref(s), C.c_void_p(dflt.ctypes.data), pout, m.mem, m.stream))
        return [out[:len(b)]]


class RaggedToDense(_Op):
    """Reference: src/ragged_to_dense.cpp (evaluate :70-174).  Inputs: begins, ends, data, target size, default,
    [pad_right].  Outputs: dense [B, T, ...], mask bool [B, T, ...].  Stateless."""

    def __init__(self, pad_right=True, pad_max_length=False, device=0, lib=None):
        super().__init__(device, lib)
        self.pad_right, self.pad_max_length = bool(pad_right), bool(pad_max_length)

    def evaluate(self, inputs):
        if len(inputs) not in (5, 6):
            raise L.OvtkError(L.E_ARG, f"Incorrect number of inputs passed to RaggedToDense: {len(inputs)}")
        data = inputs[2]
        m = _Mem(data)
        b, pb = m.inp(inputs[0], "i32")
        e, pe = m.inp(inputs[1], "i32")
        if m.torch:
            d = data.contiguous()
            np_dtype = np.dtype(str(d.dtype).replace("torch.", "").replace("bool", "uint8"))
            pd, shape = C.c_void_p(d.data_ptr()), tuple(d.shape)
        else:
            d = np.ascontiguousarray(data)
            np_dtype, pd, shape = d.dtype, C.c_void_p(d.ctypes.data), d.shape
        m.keep.append(d)
        inner = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        target = int(np.asarray(_host(inputs[3], np.int32)).reshape(-1)[0])  # read as i32 (:82)
        dflt = np.asarray(_host(inputs[4], np_dtype)).reshape(-1)[:1].copy()
        pad_right = self.pad_right if len(inputs) == 5 else bool(np.asarray(_host(inputs[5], np.uint8)).reshape(-1)[0])
        B = len(b)
        n_out = B * target * inner
        if m.torch:
            dense = m.t.empty((B, target) + shape[1:], dtype=d.dtype, device=m.device)
            mask = m.t.empty((B, target) + shape[1:], dtype=m.t.uint8, device=m.device)
            pdense, pmask = C.c_void_p(dense.data_ptr()), C.c_void_p(mask.data_ptr())
        else:
            dense = np.empty((B, target) + shape[1:], np_dtype)
            mask = np.empty((B, target) + shape[1:], np.uint8)
            pdense, pmask = C.c_void_p(dense.ctypes.data), C.c_void_p(mask.ctypes.data)
        if n_out:
            self._chk(self._lib.ovtk_ragged_to_dense(pb, pe, C.c_int64(B), pd, C.c_int64(shape[0] if shape else 0),
                                                     int(np_dtype.itemsize), C.c_int64(inner), C.c_int32(target),
                                                     C.c_void_p(dflt.ctypes.data), int(pad_right), int(self.pad_max_length),
                                                     pdense, pmask, m.mem, self.device, m.stream))
        return [dense, mask.bool() if m.torch else mask.astype(bool)]


class VocabDecoder(_Op):
    """Reference: src/vocab_decoder.cpp (evaluate :23-87).  Inputs: ids i32[B, S], vocab strings (3), [skip_tokens].
    Outputs: ragged_begins, ragged_ends, begins, ends, chars."""

    def __init__(self, skip_tokens=(), device=0, lib=None):
        super().__init__(device, lib)
        self.skip_tokens = [int(t) for t in skip_tokens]

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ovtk_vocab_decoder_destroy(self._h)
            self._h = None

    def _ensure(self, inputs):
        if self._h:
            return
        keep = []
        vocab = _strings_struct(*inputs[1:4], keep)
        lens = keep[1] - keep[0]
        self.max_token_len = int(lens.max()) if len(lens) else 0
        self.mean_token_len = float(lens.mean()) if len(lens) else 0.0
        skip = np.asarray(self.skip_tokens, np.int32)
        p = L.VocabDecoderParams(vocab, skip.ctypes.data if len(skip) else None, len(skip), self.device)
        self._chk(self._lib.ovtk_vocab_decoder_create(C.byref(p), C.byref(self._h)))

    def _prep(self, inputs, chars_capacity):
        if len(inputs) not in (4, 5):
            raise L.OvtkError(L.E_ARG, "Too few inputs passed to VocabDecoder, it means it is not converted properly "
                                       "or it is not used in the supported pattern")
        self._ensure(inputs)
        m = _Mem(inputs[0])
        ids, pids = m.inp(inputs[0], "i32")
        B, S = int(ids.shape[0]), int(ids.shape[1])
        if len(inputs) == 5:
            skip = np.ascontiguousarray(_host(inputs[4], np.int32)).reshape(-1)
            pskip, nskip = C.c_void_p(skip.ctypes.data if len(skip) else 0), len(skip)
            if nskip == 0:  # an empty input 4 still overrides the attribute: nothing is skipped
                skip = np.zeros(1, np.int32)
                pskip = C.c_void_p(skip.ctypes.data)
            m.keep.append(skip)
        else:
            pskip, nskip = None, 0
        cap = B * S * self.max_token_len if chars_capacity is None else int(chars_capacity)
        return m, pids, B, S, pskip, nskip, cap

    def evaluate(self, inputs, chars_capacity=None):
        m, pids, B, S, pskip, nskip, cap = self._prep(inputs, chars_capacity)
        sp = max(S, 1)
        orb, porb = m.alloc(B, "i32")
        ore, pore = m.alloc(B, "i32")
        ob, pob = m.alloc(B * sp, "i32")
        oe, poe = m.alloc(B * sp, "i32")
        oc, poc = m.alloc(cap, "u8")
        out = L.StringsOut(pob, poe, poc, cap, 0)
        self._chk(self._lib.ovtk_vocab_decoder_run(self._h, pids, C.c_int64(B), C.c_int64(S), pskip, C.c_int64(nskip),
                                                   porb, pore, C.byref(out), m.mem, m.stream))
        return [orb[:B], ore[:B], ob[:B * sp], oe[:B * sp], oc[:out.n_chars]]


class ByteFallback(_Op):
    """Reference: src/byte_fallback.cpp (evaluate :16-50).  Strings (3) -> strings (3).  Stateless."""

    def evaluate(self, inputs):
        m = _Mem(inputs[2])
        b, pb = m.inp(inputs[0], "i32")
        e, pe = m.inp(inputs[1], "i32")
        c, pc = m.inp(inputs[2], "u8")
        ob, pob = m.alloc(len(b), "i32")
        oe, poe = m.alloc(len(b), "i32")
        oc, poc = m.alloc(len(c), "u8")  # byte_fallback.cpp:24
        s = L.Strings(pb, pe, pc, len(b), len(c))
        out = L.StringsOut(pob, poe, poc, len(c), 0)
        self._chk(self._lib.ovtk_byte_fallback(C.byref(s), C.byref(out), m.mem, self.device, m.stream))
        return [ob[:len(b)], oe[:len(b)], oc[:out.n_chars]]


class TrieTokenizer(_Op):
    """Reference: src/trie_tokenizer.cpp (evaluate :23-81).  Inputs: ragged strings (5), vocab (3), indices i32.
    Outputs: begins, ends, ids."""

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ovtk_trie_tokenizer_destroy(self._h)
            self._h = None

    def _ensure(self, inputs):
        if self._h:
            return
        keep = []
        vocab = _strings_struct(inputs[5], inputs[6], inputs[7], keep)
        idx = _host(inputs[8], np.int32).reshape(-1)
        if len(idx) != vocab.n:
            raise L.OvtkError(L.E_ARG, "Vocab size must be equal to Indices size")   # trie_tokenizer.cpp:38
        self._chk(self._lib.ovtk_trie_tokenizer_create(C.byref(vocab), idx.ctypes.data_as(C.c_void_p), self.device, C.byref(self._h)))

    def evaluate(self, inputs):
        self._ensure(inputs)
        m = _Mem(inputs[4])
        rs, (rb, _, _, _, c) = _ragged_in(m, inputs)
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c)   # trie_tokenizer.cpp:60
        ids, pids = m.alloc(cap, "i32")
        out = L.RaggedI32Out(pob, poe, pids, cap, 0, 0)
        self._chk(self._lib.ovtk_trie_tokenizer_run(self._h, C.byref(rs), C.byref(out), m.mem, m.stream))
        return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]


def _device_alloc(n, kind, device):
    """Device-side output of the staging ops: a torch CUDA tensor, or -- no GPU in the process: the emulator build,
    whose "device" memory is host memory -- a numpy array."""
    n = max(int(n), 1)
    try:
        import torch
        if torch.cuda.is_available():
            dt = {"i32": torch.int32, "u8": torch.uint8}[kind]
            t = torch.empty(n, dtype=dt, device=torch.device("cuda", device))
            return t, C.c_void_p(t.data_ptr()), C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    except ImportError:  # pragma: no cover
        pass
    a = np.empty(n, dtype=_NP[kind])
    return a, C.c_void_p(a.ctypes.data), C.c_void_p(0)


class StringTensorUnpack(_Op):
    """Reference: src/string_tensor_unpack.cpp (evaluate :45-78, the packed-u8 branch; the element::string branch is a
    host container and stays with the caller).  Input: packed u8 [i32 n][i32 begin_0][i32 end_i x n][bytes] in HOST
    (numpy) or device (torch CUDA) memory.  Outputs: begins, ends, chars in DEVICE memory -- the H2D staging step."""

    def __init__(self, mode="begins_ends", device=0, lib=None):
        super().__init__(device, lib)
        if mode != "begins_ends":
            raise L.OvtkError(L.E_ARG, f"StringTensorUnpack supporst only 'begins_ends' mode, but get {mode}")

    def evaluate(self, inputs):
        packed = inputs[0]
        on_dev = _is_torch(packed) and packed.device.type == "cuda"
        if on_dev:
            packed = packed.contiguous()
            ptr, nbytes, mem = C.c_void_p(packed.data_ptr()), packed.numel(), L.MEM_DEVICE
        else:
            packed = _host(packed, np.uint8)
            ptr, nbytes, mem = C.c_void_p(packed.ctypes.data), packed.size, L.MEM_HOST
        rows = max((nbytes - 8) // 4, 0)
        ob, pob, stream = _device_alloc(rows, "i32", self.device)
        oe, poe, _ = _device_alloc(rows, "i32", self.device)
        oc, poc, _ = _device_alloc(nbytes, "u8", self.device)
        out = L.StringsOut(pob, poe, poc, nbytes, 0)
        n = C.c_int64(0)
        self._chk(self._lib.ovtk_string_tensor_unpack(ptr, C.c_int64(nbytes), mem, C.byref(out), C.c_int64(rows), C.byref(n),
                                                      self.device, stream))
        return [ob[:n.value], oe[:n.value], oc[:out.n_chars]]


class StringTensorPack(_Op):
    """Inverse of StringTensorUnpack (OpenVINO's opset15 StringTensorPack followed by the packed-u8 serialisation of
    python/openvino_tokenizers/utils.py): begins, ends, chars in DEVICE memory -> one packed u8 buffer, returned in
    host memory (`to_host=True`, the D2H staging step) or left on the device."""

    def evaluate(self, inputs, to_host=True):
        m = _Mem(inputs[2])
        if not m.torch:
            try:  # host arrays are uploaded first; without a GPU (emulator build) host memory IS device memory
                import torch
                if torch.cuda.is_available():
                    dev = torch.device("cuda", self.device)
                    kinds = (np.int32, np.int32, np.uint8)
                    return self.evaluate([torch.as_tensor(_host(x, k), device=dev) for x, k in zip(inputs, kinds)], to_host)
            except ImportError:  # pragma: no cover
                pass
        b, pb = m.inp(inputs[0], "i32")
        e, pe = m.inp(inputs[1], "i32")
        c, pc = m.inp(inputs[2], "u8")
        s = L.Strings(pb, pe, pc, len(b), len(c))
        total = int((e - b).clamp(min=0).sum()) if m.torch else int(np.maximum(e.astype(np.int64) - b, 0).sum())
        cap = int(self._lib.ovtk_string_tensor_packed_bytes(C.c_int64(len(b)), C.c_int64(total)))
        if to_host or not m.torch:
            buf = np.empty(cap, np.uint8)
            ptr, mem = C.c_void_p(buf.ctypes.data), L.MEM_HOST
        else:
            buf = m.t.empty(cap, dtype=m.t.uint8, device=m.device)
            ptr, mem = C.c_void_p(buf.data_ptr()), L.MEM_DEVICE
        n = C.c_int64(0)
        self._chk(self._lib.ovtk_string_tensor_pack(C.byref(s), ptr, C.c_int64(cap), mem, C.byref(n), self.device, m.stream))
        return [buf[:n.value]]


class UTF8Validate(_Op):
    """Reference: src/utf8_validate.cpp (evaluate :18-143).  Strings (3) -> strings (3); attribute replace_mode."""

    def __init__(self, replace_mode=False, device=0, lib=None):
        super().__init__(device, lib)
        self.replace_mode = bool(replace_mode)

    def evaluate(self, inputs):
        m = _Mem(inputs[2])
        b, pb = m.inp(inputs[0], "i32")
        e, pe = m.inp(inputs[1], "i32")
        c, pc = m.inp(inputs[2], "u8")
        ob, pob = m.alloc(len(b), "i32")
        oe, poe = m.alloc(len(b), "i32")
        cap = 3 * len(c)  # utf8_validate.cpp:31-33
        oc, poc = m.alloc(cap, "u8")
        s = L.Strings(pb, pe, pc, len(b), len(c))
        out = L.StringsOut(pob, poe, poc, cap, 0)
        self._chk(self._lib.ovtk_utf8_validate(C.byref(s), int(self.replace_mode), C.byref(out), m.mem, self.device, m.stream))
        return [ob[:len(b)], oe[:len(b)], oc[:out.n_chars]]


class FuzeRagged(_Op):
    """Reference: src/fuze.cpp (evaluate :20-40).  ragged_begins, ragged_ends, begins, ends -> begins, ends."""

    def evaluate(self, inputs):
        m = _Mem(inputs[2])
        rb, prb = m.inp(inputs[0], "i32")
        re_, pre = m.inp(inputs[1], "i32")
        b, pb = m.inp(inputs[2], "i32")
        e, pe = m.inp(inputs[3], "i32")
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        self._chk(self._lib.ovtk_fuze_ragged(prb, pre, C.c_int64(len(rb)), pb, pe, C.c_int64(len(b)), pob, poe, m.mem,
                                             self.device, m.stream))
        return [ob[:len(rb)], oe[:len(rb)]]


class Truncate(_Op):
    """Reference: src/truncate.cpp (evaluate :37-150).  Inputs: (begins, ends, data) x 1 or 2, max_length,
    trunc_side, [trunc_mode].  Outputs: (begins, ends, data) per input; data is the input tensor."""

    def evaluate(self, inputs):
        k = 1 if len(inputs) < 8 else 2
        max_length = int(np.asarray(_host(inputs[3 * k], np.int32)).reshape(-1)[0])
        side = _bytes_of(inputs[3 * k + 1])
        mode = _bytes_of(inputs[3 * k + 2]) if len(inputs) > 3 * k + 2 else b"longest_first"
        m = _Mem(inputs[0])
        n = len(inputs[0])
        ins, outs = [], []
        for j in range(2):
            if j < k:
                ins += [m.inp(inputs[3 * j], "i32")[1], m.inp(inputs[3 * j + 1], "i32")[1]]
                outs += [m.alloc(n, "i32"), m.alloc(n, "i32")]
            else:
                ins += [None, None]
                outs += [(None, None), (None, None)]
        self._chk(self._lib.ovtk_truncate(k, ins[0], ins[1], ins[2], ins[3], C.c_int64(n), C.c_int32(max_length), side,
                                          mode, outs[0][1], outs[1][1], outs[2][1], outs[3][1], m.mem, self.device,
                                          m.stream))
        res = []
        for j in range(k):
            res += [outs[2 * j][0][:n], outs[2 * j + 1][0][:n], inputs[3 * j + 2]]
        return res


class CombineSegments(_Op):
    """Reference: src/combine_segments.cpp (evaluate :36-134).  Inputs: (begins, ends, data) x k, segment_ids[k].
    Outputs: begins, ends, data, begins, ends, segment ids (the two ragged outputs share their offsets)."""

    def evaluate(self, inputs, capacity=None):
        k = (len(inputs) - 1) // 3
        ids = _host(inputs[-1], np.int32).reshape(-1)
        if len(ids) != k or (len(inputs) - 1) % 3:
            raise L.OvtkError(L.E_ARG, "CombineSegments: expected 3*k + 1 inputs with k segment ids")
        m = _Mem(inputs[2])
        segs = (L.RaggedI32 * k)()
        rows = 0
        bound = 0
        for j in range(k):
            b, pb = m.inp(np.atleast_1d(inputs[3 * j]) if not _is_torch(inputs[3 * j]) else inputs[3 * j].reshape(-1), "i32")
            e, pe = m.inp(np.atleast_1d(inputs[3 * j + 1]) if not _is_torch(inputs[3 * j + 1]) else inputs[3 * j + 1].reshape(-1), "i32")
            d, pd = m.inp(inputs[3 * j + 2], "i32")
            segs[j] = L.RaggedI32(pb, pe, pd, len(b), len(d))
            rows = max(rows, len(b))
        for j in range(k):
            bound += int(segs[j].n_data) * (rows if segs[j].n == 1 else 1)
        cap = int(capacity) if capacity is not None else bound
        ob, pob = m.alloc(rows, "i32")
        oe, poe = m.alloc(rows, "i32")
        od, pod = m.alloc(cap, "i32")
        oi, poi = m.alloc(cap, "i32")
        n_out = C.c_int64(0)
        self._chk(self._lib.ovtk_combine_segments(segs, k, ids.ctypes.data_as(C.c_void_p), pob, poe, pod, poi,
                                                  C.c_int64(cap), C.byref(n_out), m.mem, self.device, m.stream))
        return [ob[:rows], oe[:rows], od[:n_out.value], ob[:rows], oe[:rows], oi[:n_out.value]]


class FusedEncodeTail(_Op):
    """Truncate -> CombineSegments -> RaggedToDense (input_ids + mask) -> RaggedToDense (token_type_ids) in one kernel
    (ovtk_encode_tail_run): what tokenizer_pipeline.py's TruncationStep, CombineSegmentsStep and PaddingStep build, without
    the combined ragged tensor.  segments: [(begins, ends, data)] i32; truncated: indices of the one or two segments
    Truncate applies to; target_dim None = the longest combined row (the PaddingStep's ReduceMax)."""

    def __init__(self, max_length=2**31 - 1, trunc_side="right", trunc_mode="longest_first", pad_right=True, device=0, lib=None):
        super().__init__(device, lib)
        self.max_length, self.trunc_side, self.trunc_mode, self.pad_right = int(max_length), trunc_side, trunc_mode, bool(pad_right)

    def evaluate(self, segments, segment_ids, truncated=(), pad_value=0, type_pad_value=0, target_dim=None):
        k = len(segments)
        ids = _host(segment_ids, np.int32).reshape(-1)
        m = _Mem(next((s[2] for s in segments if _is_torch(s[2])), segments[0][2]))
        segs = (L.RaggedI32 * k)()
        rows = 0
        for j, (b, e, d) in enumerate(segments):
            b = b.reshape(-1) if _is_torch(b) else np.atleast_1d(b)
            e = e.reshape(-1) if _is_torch(e) else np.atleast_1d(e)
            bb, pb = m.inp(b, "i32")
            ee, pe = m.inp(e, "i32")
            dd, pd = m.inp(d, "i32")
            segs[j] = L.RaggedI32(pb, pe, pd, len(bb), len(dd))
            rows = max(rows, len(bb))
        tr = list(truncated) + [-1, -1]
        p = L.EncodeTailParams(C.cast(segs, C.c_void_p), k, ids.ctypes.data_as(C.c_void_p), int(tr[0]), int(tr[1]),
                               C.c_int32(min(self.max_length, 2**31 - 1)), _bytes_of(self.trunc_side), _bytes_of(self.trunc_mode),
                               -1 if target_dim is None else int(target_dim), int(pad_value), int(type_pad_value), int(self.pad_right))
        T = C.c_int32(0)
        if target_dim is None:   # measure first: the width decides the output shapes
            one, pone = m.alloc(1, "i32")
            rc = self._lib.ovtk_encode_tail_run(C.byref(p), pone, None, None, C.c_int64(0), C.byref(T), m.mem, self.device, m.stream)
            if rc not in (L.OVTK_OK, L.E_CAPACITY):
                self._chk(rc)
            p.target_dim = T.value
        width = int(p.target_dim)
        n = rows * width
        out_ids, pids = m.alloc(n, "i32")
        mask, pmask = m.alloc(n, "bool")
        types, ptypes = m.alloc(n, "i32")
        self._chk(self._lib.ovtk_encode_tail_run(C.byref(p), pids, pmask, ptypes, C.c_int64(max(n, 1)), C.byref(T), m.mem, self.device,
                                                 m.stream))
        shape = (rows, width)
        if m.torch:
            return [out_ids[:n].reshape(shape), mask[:n].reshape(shape).bool(), types[:n].reshape(shape)]
        return [out_ids[:n].reshape(shape), mask[:n].reshape(shape).astype(bool), types[:n].reshape(shape)]


class FusedDetokenizer:
    """VocabDecoder -> [ByteFallback] -> FuzeRagged in one pass (ovtk_detokenize_run): the same begins/ends/chars as
    chaining the three ops (tokenizer_pipeline.py:1321-1371) without the per-token offsets going through HBM."""

    def __init__(self, decoder: VocabDecoder, byte_fallback=False):
        self.decoder, self.byte_fallback = decoder, bool(byte_fallback)
        self._side = {}   # (device, n) -> the HIP streams evaluate_chunked spreads its chunks over

    def _side_streams(self, m, n):
        """The same streams for every call: torch's caching allocator keeps freed blocks PER STREAM, and a chunk's output is
        2 GB -- with fresh streams per call every pass allocated its chunks anew (hipMalloc, and hipFree once the card was
        full: config 5 then took 400-740 ms per pass instead of 15)."""
        key = (str(m.device), int(n))
        if key not in self._side:
            self._side[key] = [m.t.cuda.Stream(m.device) for _ in range(int(n))]
        return self._side[key]

    def evaluate(self, inputs, chars_capacity=None):
        d = self.decoder
        m, pids, B, S, pskip, nskip, cap = d._prep(inputs, chars_capacity)
        ob, pob = m.alloc(B, "i32")
        oe, poe = m.alloc(B, "i32")
        oc, poc = m.alloc(cap, "u8")
        out = L.StringsOut(pob, poe, poc, cap, 0)
        d._chk(d._lib.ovtk_detokenize_run(d._h, pids, C.c_int64(B), C.c_int64(S), pskip, C.c_int64(nskip),
                                          int(self.byte_fallback), C.byref(out), m.mem, m.stream))
        return [ob[:B], oe[:B], oc[:out.n_chars]]

    def enqueue(self, inputs, chars_capacity=None):
        """evaluate() in two halves for CUDA tensors (ovtk_detokenize_enqueue / ovtk_detokenize_finish): launches the
        passes and returns a ticket; `ticket()` waits for them and returns evaluate()'s outputs."""
        d = self.decoder
        m, pids, B, S, pskip, nskip, cap = d._prep(inputs, chars_capacity)
        if not m.torch:
            raise L.OvtkError(L.E_ARG, "enqueue() needs device-resident (torch CUDA) inputs")
        ob, pob = m.alloc(B, "i32")
        oe, poe = m.alloc(B, "i32")
        oc, poc = m.alloc(cap, "u8")
        out = L.StringsOut(pob, poe, poc, cap, 0)
        pending = C.c_void_p()
        d._chk(d._lib.ovtk_detokenize_enqueue(d._h, pids, C.c_int64(B), C.c_int64(S), pskip, C.c_int64(nskip),
                                              int(self.byte_fallback), C.byref(out), m.stream, C.byref(pending)))

        def ticket(_keep=(m, inputs)):
            d._chk(d._lib.ovtk_detokenize_finish(pending, C.byref(out)))
            return [ob[:B], oe[:B], oc[:out.n_chars]]
        return ticket


    def evaluate_chunked(self, inputs, chunk_chars=(1 << 31) - 2, bytes_per_id=None, streams=3, depth=2, sink=None):
        """ids [B, S] of ANY size -> one string per row, in row chunks (BASELINE config 5: 1 M x 2 048 ids are 8-9 GB of text).
        The reference counts chars in int32 (src/vocab_decoder.cpp:62-63,69,80), so one VocabDecoder call cannot hold more
        than 2^31 bytes: the batch is cut into chunks of consecutive rows whose output stays below `chunk_chars`, each
        chunk is one ovtk_detokenize_enqueue on one of `streams` HIP streams, `depth` chunks ahead of the one being
        completed (host arrays: one blocking ovtk_detokenize_run per chunk).  Rows per chunk come from an estimate of
        the output bytes per id (`bytes_per_id`, default: the vocabulary's mean token length; refined with every
        completed chunk); a chunk that overflows its buffer all the same is reported by the library with the size it
        needs (OVTK_E_CAPACITY) and is cut again.  Returns the chunks in row order as (row_begin, row_end, begins, ends,
        chars) -- begins / ends int32 offsets into that chunk's own chars -- or, with `sink`, hands each one to
        sink(row_begin, row_end, begins, ends, chars) as it completes and returns the number of chunks (the tensors may be
        dropped or reused by the caller; nothing else keeps them)."""
        d = self.decoder
        d._ensure(inputs)
        ids = inputs[0]
        B, S = int(ids.shape[0]), int(ids.shape[1])
        tail = list(inputs[1:])
        chunk_chars = int(min(chunk_chars, (1 << 31) - 2))
        est = float(bytes_per_id) if bytes_per_id else max(d.mean_token_len, 0.25)
        row_cap = max(1, ((1 << 31) - 2) // max(S, 1))   # batch * seq_len must fit int32 as well (vocab_decoder.cpp:45-46)
        m0 = _Mem(ids)
        side = self._side_streams(m0, max(int(streams), 1)) if m0.torch else []
        done, inflight, n_launched = [], [], 0
        self.chunk_log = []   # (row_begin, row_end, capacity, bytes or None when it overflowed): how the batch was cut
        lo, pending_rows = 0, []   # pending_rows: re-cut chunks (row ranges) that go before the rest of the batch

        forced_cap = {}   # (row_begin, row_end) -> capacity to launch it with (a single row that overflowed its estimate)

        def next_range():
            nonlocal lo
            if pending_rows:
                return pending_rows.pop(0)
            if lo >= B:
                return None
            rows = int(chunk_chars / (max(S, 1) * est * 1.08)) if S else B
            rows = max(1, min(rows, row_cap, B - lo))
            r = (lo, lo + rows)
            lo += rows
            return r

        def launch(r):
            nonlocal n_launched
            a, b = r
            cap = int(min(chunk_chars, (b - a) * S * est * 1.12 + 4096))
            cap = forced_cap.pop((a, b), cap)
            part = [ids[a:b]] + tail
            if m0.torch:
                st = side[n_launched % len(side)]
                st.wait_stream(m0.t.cuda.current_stream(m0.device))   # the ids were produced on the caller's stream
                with m0.t.cuda.stream(st):
                    m, pids, nb, _, pskip, nskip, _ = d._prep(part, cap)
                    ob, pob = m.alloc(nb, "i32")
                    oe, poe = m.alloc(nb, "i32")
                    oc, poc = m.alloc(cap, "u8")
                    out = L.StringsOut(pob, poe, poc, cap, 0)
                    pend = C.c_void_p()
                    d._chk(d._lib.ovtk_detokenize_enqueue(d._h, pids, C.c_int64(nb), C.c_int64(S), pskip, C.c_int64(nskip),
                                                          int(self.byte_fallback), C.byref(out), m.stream, C.byref(pend)))
                n_launched += 1
                return (r, cap, m, ob, oe, oc, out, pend, st)
            m, pids, nb, _, pskip, nskip, _ = d._prep(part, cap)
            ob, pob = m.alloc(nb, "i32")
            oe, poe = m.alloc(nb, "i32")
            oc, poc = m.alloc(cap, "u8")
            out = L.StringsOut(pob, poe, poc, cap, 0)
            rc = d._lib.ovtk_detokenize_run(d._h, pids, C.c_int64(nb), C.c_int64(S), pskip, C.c_int64(nskip),
                                            int(self.byte_fallback), C.byref(out), m.mem, m.stream)
            n_launched += 1
            return (r, cap, m, ob, oe, oc, out, rc, None)

        def complete(item):
            nonlocal est
            (a, b), cap, m, ob, oe, oc, out, pend, st = item
            rc = d._lib.ovtk_detokenize_finish(pend, C.byref(out)) if st is not None else pend
            if rc == L.E_CAPACITY:
                need = int(out.n_chars)
                self.chunk_log.append((a, b, cap, None))
                if b - a == 1:
                    if cap >= chunk_chars or need >= (1 << 31) - 1:
                        d._chk(rc)   # one row alone is beyond the chunk size (or beyond int32 offsets): nothing to cut
                    forced_cap[(a, b)] = chunk_chars   # once more, with all the room a chunk may have
                    pending_rows[:0] = [(a, b)]
                    return
                # cut again: as many pieces as the reported need asks for (at least two); the estimate learns from it
                if 0 < need < (1 << 31) - 1:
                    est = max(est, need / max((b - a) * S, 1))
                parts = max(2, -(-need // max(int(chunk_chars / 1.08), 1))) if need < (1 << 31) - 1 else 2
                parts = min(parts, b - a) if b - a > 1 else 1
                step = -(-(b - a) // parts)
                pending_rows[:0] = [(x, min(x + step, b)) for x in range(a, b, step)]
                return
            d._chk(rc)
            n = int(out.n_chars)
            self.chunk_log.append((a, b, cap, n))
            if (b - a) * S:
                est = 0.5 * est + 0.5 * max(n / ((b - a) * S), 0.25)
            res = (a, b, ob[:b - a], oe[:b - a], oc[:n])
            if st is not None:
                m0.t.cuda.current_stream(m0.device).wait_stream(st)   # (finish() waited on the host; this orders later device work)
            if sink is not None:
                sink(*res)
                done.append(None)
            else:
                done.append(res)

        # Chunks complete in launch order, so `done` is in row order as long as a re-cut chunk is relaunched before anything
        # behind it completes: on an overflow everything in flight behind it is completed first only AFTER its pieces --
        # simplest: drain the pipeline, relaunch the pieces, go on.
        def drop(items):   # calls in flight whose results are not wanted: finished all the same (the library owns them until then)
            for it in items:
                if it[8] is not None:
                    d._lib.ovtk_detokenize_finish(it[7], C.byref(it[6]))
            items.clear()

        try:
            while True:
                r = next_range()
                if r is None and not inflight:
                    break
                if r is not None:
                    inflight.append(launch(r))
                while inflight and (len(inflight) > depth or r is None or not m0.torch):
                    item = inflight.pop(0)
                    before = len(pending_rows)
                    complete(item)
                    if len(pending_rows) > before and inflight:   # overflow: what was launched behind it is redone after its pieces
                        redo = [it[0] for it in inflight]
                        drop(inflight)
                        pending_rows.extend(redo)
                        pending_rows.sort()
                        break
        finally:
            drop(inflight)   # (an error in the middle of the pipeline: nothing stays queued behind the caller's back)
        return len(done) if sink is not None else done


class FusedSplitWordpiece:
    """RegexSplit(\\s+, remove) -> RegexSplit(BERT delimiters, isolate) -> WordpieceTokenizer in one pass
    (ovtk_wordpiece_encode_run): the same result as chaining the three ops (what tokenizer_pipeline.py:392-435 and
    :641-659 build for BERT models) without materialising the words."""

    def __init__(self, whitespace: RegexSplit, delimiters: RegexSplit, wordpiece: WordpieceTokenizer):
        self.whitespace, self.delimiters, self.wordpiece = whitespace, delimiters, wordpiece

    def evaluate(self, ragged_inputs, whitespace_pattern, delimiters_pattern, wordpiece_constant_inputs):
        """ragged_inputs: inputs 0-4 of the first RegexSplit; wordpiece_constant_inputs: inputs 5-8 of WordpieceTokenizer."""
        self.whitespace._ensure(whitespace_pattern)
        self.delimiters._ensure(delimiters_pattern)
        wp = self.wordpiece
        wp._ensure(list(ragged_inputs[:5]) + list(wordpiece_constant_inputs))
        unk = int(np.asarray(_host(wordpiece_constant_inputs[3], np.int32)).reshape(-1)[0])
        m = _Mem(ragged_inputs[4])
        rs, (rb, _, _, _, c) = _ragged_in(m, ragged_inputs)
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c)
        ids, pids = m.alloc(cap, "i32")
        out = L.RaggedI32Out(pob, poe, pids, cap, 0, 0)
        L.check(wp._lib, wp._lib.ovtk_wordpiece_encode_run(wp._h, self.whitespace._h, self.delimiters._h, C.byref(rs),
                                                           C.c_int32(unk), C.byref(out), m.mem, m.stream))
        return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]

    def enqueue(self, ragged_inputs, whitespace_pattern, delimiters_pattern, wordpiece_constant_inputs):
        """evaluate() in two halves for CUDA tensors (ovtk_wordpiece_encode_enqueue / ovtk_encode_finish): returns a
        ticket; `ticket()` waits for the kernels and returns evaluate()'s outputs."""
        self.whitespace._ensure(whitespace_pattern)
        self.delimiters._ensure(delimiters_pattern)
        wp = self.wordpiece
        wp._ensure(list(ragged_inputs[:5]) + list(wordpiece_constant_inputs))
        unk = int(np.asarray(_host(wordpiece_constant_inputs[3], np.int32)).reshape(-1)[0])
        m = _Mem(ragged_inputs[4])
        if not m.torch:
            raise L.OvtkError(L.E_ARG, "enqueue() needs device-resident (torch CUDA) inputs")
        rs, (rb, _, _, _, c) = _ragged_in(m, ragged_inputs)
        ob, pob = m.alloc(len(rb), "i32")
        oe, poe = m.alloc(len(rb), "i32")
        cap = len(c)
        ids, pids = m.alloc(cap, "i32")
        out = L.RaggedI32Out(pob, poe, pids, cap, 0, 0)
        lib = wp._lib
        pending = C.c_void_p()
        L.check(lib, lib.ovtk_wordpiece_encode_enqueue(wp._h, self.whitespace._h, self.delimiters._h, C.byref(rs), C.c_int32(unk),
                                                       C.byref(out), m.stream, C.byref(pending)))

        def ticket(_keep=m):
            L.check(lib, lib.ovtk_encode_finish(pending, C.byref(out)))
            return [ob[:out.n_rows], oe[:out.n_rows], ids[:out.n_data]]
        return ticket
