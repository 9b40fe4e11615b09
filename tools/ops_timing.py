"""Times the ops of SURVEY 8(f) and the ops around the hot path that bench.py has no configuration for, one op = one library call,
on a config-2-shaped batch (65 536 rows x ~512 bytes of zipf text, a GPT-2 special token in one row in a hundred; the ids of that
batch for the ops behind the tokenizer), inputs and outputs in HBM:

    UTF8Validate (both modes), SpecialTokensSplit (the op: count pass, scan, write pass), RegexSplit (the op), StringTensorPack /
    StringTensorUnpack, TrieTokenizer, Truncate, CombineSegments, RaggedToDense (in bench.py too: --config r2d), the fused tail
    (ovtk_encode_tail_run: Truncate -> CombineSegments -> RaggedToDense x 2 in one call).

Per op one JSON line: wall time per call (Python + the library's host side + the kernels; the call returns when its results are
complete, so host waits are inside), the algorithmic bytes the op has to move (inputs read once + outputs written once) and the rate
they make, and whether a prefix of the result equals the oracle's.  The kernels' own durations: run this script under
`rocprofv3 --kernel-trace --stats` (profiles/README.md); VERDICT r04 weak 6 asked for both.

    python tools/ops_timing.py [--rows 65536] [--bytes 512] [--reps 10] [--check-rows 256]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--bytes", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--check-rows", type=int, default=256)
    ap.add_argument("--emu", action="store_true", help="the SIMT-emulator build on host arrays (a smoke test of this script; the times mean nothing)")
    args = ap.parse_args()
    from openvino_tokenizers_amd import _lib as L
    from openvino_tokenizers_amd.ops import (BPETokenizer, CombineSegments, FusedEncodeTail, FusedSplitBPE, RaggedToDense, RegexSplit,
                                             SpecialTokensSplit, StringTensorPack, StringTensorUnpack, TrieTokenizer, Truncate, UTF8Validate)
    from oracle import oracle as O
    from tools.harness import BpeTok, pack_strings
    from tools.workloads import TextModel, ragged_rows

    if args.emu:
        lib = L.load(Path(__file__).resolve().parent.parent / "tests" / "emu" / "build" / "libovtk_emu.so")

        def sync():
            pass
    else:
        import torch
        lib = L.load()
        dev = torch.device("cuda", 0)
        sync = torch.cuda.synchronize
    tok = BpeTok.load("gpt2")
    n = args.rows
    b, e, c = TextModel(1234, "zipf").batch(n, args.bytes, seed=1000)
    rng = np.random.default_rng(1)
    special = np.frombuffer(b"<|endoftext|>", np.uint8)
    for i in np.flatnonzero(rng.random(n) < 0.01):
        if e[i] - b[i] > 2 * len(special):
            at = int(b[i]) + int(rng.integers(0, e[i] - b[i] - len(special)))
            c[at:at + len(special)] = special
    rb, re_ = ragged_rows(n)
    host = [rb, re_, b, e, c]
    d = host if args.emu else [torch.as_tensor(x, device=dev) for x in host]
    k = min(args.check_rows, n)
    head = [rb[:k], re_[:k], b[:k], e[:k], c[: int(e[k - 1])]]

    def to_np(x):
        return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

    def same(ref, got, upto=None):
        for r, g in zip(ref, got):
            g = to_np(g)
            r = np.asarray(r)
            m = len(r) if upto is None else min(len(r), upto)
            if not np.array_equal(r.reshape(-1)[:m], g.reshape(-1)[:m]):
                return False
        return True

    def timed(name, fn, algo_bytes, check=None, note=""):
        for _ in range(2):
            out = fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = fn()
        sync()
        dt = (time.perf_counter() - t0) / args.reps
        ok = None if check is None else bool(check(out))
        print(json.dumps({"op": name, "ms_per_call": round(dt * 1e3, 4), "algorithmic_MB": round(algo_bytes / 1e6, 2),
                          "GB_per_s": round(algo_bytes / dt / 1e9, 1), "rows": n, "prefix_equals_oracle": ok, "note": note}), flush=True)
        return out

    n_c = len(c)
    # ---- UTF8Validate (src/utf8_validate.cpp:18-143): strings in, strings out
    for mode in (False, True):
        op = UTF8Validate(replace_mode=mode, lib=lib)
        ref = O.utf8_validate(b[:k], e[:k], c, mode)
        timed(f"UTF8Validate(replace_mode={mode})", lambda op=op: op.evaluate(d[2:5]), 2 * n_c + 16 * n,
              lambda out, ref=ref: same(ref[:2], [to_np(out[0])[:k], to_np(out[1])[:k]]), "valid ASCII text: every byte is copied")
    # ---- SpecialTokensSplit, the op (src/special_tokens_split.cpp:61-162)
    sp_text = r"(\<\|endoftext\|\>)"   # what SpecialTokensSplitStep builds for this token (tokenizer_pipeline.py:138-159, quote_meta)
    sp_pat = np.frombuffer(sp_text.encode(), np.uint8)
    sp = SpecialTokensSplit(lib=lib)
    ref_sp = O.SpecialTokensSplit(sp_text)(*head)
    sp_out = timed("SpecialTokensSplit", lambda: sp.evaluate(d + [sp_pat]), n_c + 8 * n + 9 * n + 8 * n,
                   lambda out: same(ref_sp[:2], [to_np(out[0])[:k], to_np(out[1])[:k]]) and same([ref_sp[2], ref_sp[3]], [to_np(out[2]), to_np(out[3])], upto=len(ref_sp[2])),
                   "one row in a hundred holds the token; the op as the graph would run it alone (inside the fused encode: special_sparse_kernel, bench.py --config pipeline)")
    # ---- RegexSplit, the op (src/regex_split.cpp:124-324), GPT-2's pattern
    pat = tok.pattern_u8()
    rs = RegexSplit("isolate", lib=lib)
    ref_rs = O.RegexSplit(tok.pattern, "isolate")(*head)
    rs_out = timed("RegexSplit(isolate, GPT-2 pattern)", lambda: rs.evaluate(d + [pat]), n_c + 16 * n + 8 * (n_c // 4),
                   lambda out: same(ref_rs[:2], [to_np(out[0])[:k], to_np(out[1])[:k]]) and same([ref_rs[2], ref_rs[3]], [to_np(out[2]), to_np(out[3])], upto=len(ref_rs[2])),
                   "the op alone: pieces as begins / ends (inside the fused encode the pieces never exist)")
    # ---- BPETokenizer, the op (src/bpe_tokenizer.cpp:47-164): the pieces of the RegexSplit above in, ragged ids out -- what a graph
    # that is not fused runs behind RegexSplit
    bpe_op = BPETokenizer(**tok.attrs, lib=lib)
    ref_bpe = tok.oracle()(*ref_rs[:5])
    n_pieces = len(rs_out[2])
    timed("BPETokenizer (pre-split pieces)", lambda: bpe_op.evaluate(list(rs_out[:5]) + tok.consts), n_c + 8 * n_pieces + 16 * n + 4 * (n_c // 4),
          lambda out: same(ref_bpe[:2], [to_np(out[0])[:k], to_np(out[1])[:k]]) and same([ref_bpe[2]], [to_np(out[2])], upto=len(ref_bpe[2])),
          f"{n_pieces} pieces: lookup_kernel<kPieces> -> merge_kernel -> compact_kernel")
    del rs_out, sp_out
    # ---- StringTensorPack / Unpack (src/string_tensor_pack.cpp:40-86, string_tensor_unpack.cpp:45-78)
    pk = StringTensorPack(lib=lib)
    packed = timed("StringTensorPack", lambda: pk.evaluate(d[2:5], to_host=args.emu), 2 * n_c + 12 * n, None, "device to device")
    un = StringTensorUnpack(lib=lib)
    timed("StringTensorUnpack", lambda: un.evaluate([packed[0]]), 2 * n_c + 12 * n,
          lambda out: same([b[:k], e[:k]], [to_np(out[0])[:k], to_np(out[1])[:k]]), "packed u8 tensor on the device -> begins / ends / chars")
    # ---- TrieTokenizer (src/trie_tokenizer.cpp:23-81): the vocabulary's strings as the trie, longest match, unknown bytes skipped
    vb, ve, vc = pack_strings(tok.vocab)
    idx = np.arange(len(tok.vocab), dtype=np.int32)
    tr = TrieTokenizer(lib=lib)
    ref_tr = O.TrieTokenizer(tok.vocab, idx)(*head)
    tr_out = timed("TrieTokenizer", lambda: tr.evaluate(d + [vb, ve, vc, idx]), n_c + 16 * n + 4 * (n_c // 3),
                   lambda out: same(ref_tr[:2], [to_np(out[0])[:k], to_np(out[1])[:k]]) and same([ref_tr[2]], [to_np(out[2])], upto=len(ref_tr[2])),
                   f"V = {len(tok.vocab)} (the GPT-2-shaped vocabulary's strings), a lane per 64-byte segment + a lane per row that stitches them")
    del tr_out
    # ---- the ids of the batch (the fused encode), then the ops behind the tokenizer
    fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**tok.attrs, lib=lib))
    ib, ie, ids = fused.evaluate(d + [pat], tok.consts)
    n_t = int(len(ids))
    ibh, ieh, idh = to_np(ib), to_np(ie), to_np(ids)
    max_len = 64
    tc = Truncate(lib=lib)
    ref_t = O.truncate([(ibh[:k], ieh[:k])], max_len, "right", "longest_first")
    t_out = timed("Truncate(max_length=64, right)", lambda: tc.evaluate([ib, ie, ids, np.array([max_len], np.int32), b"right", b"longest_first"]),
                  16 * n, lambda out: same(ref_t[0][:2], [to_np(out[0])[:k], to_np(out[1])[:k]]), "offsets only: the data tensor passes through")
    bos = (np.array([0], np.int32), np.array([1], np.int32), np.array([50256], np.int32))
    bos_d = bos if args.emu else tuple(torch.as_tensor(x, device=dev) for x in bos)   # (constants of the graph: resident like everything else)
    cs = CombineSegments(lib=lib)
    seg_ids = np.array([0, 0, 0], np.int32)
    ref_c = O.combine_segments([bos, (ref_t[0][0], ref_t[0][1], idh), bos], seg_ids)
    kept = int(np.minimum(ieh - ibh, max_len).sum())
    c_out = timed("CombineSegments(bos + truncated ids + eos)", lambda: cs.evaluate(list(bos_d) + [t_out[0], t_out[1], ids] + list(bos_d) + [seg_ids]),
                  4 * kept + 2 * 4 * (kept + 2 * n) + 16 * n,
                  lambda out: same(ref_c[:2], [to_np(out[0])[:k], to_np(out[1])[:k]]) and same([ref_c[2]], [to_np(out[2])], upto=len(ref_c[2])),
                  "ids + segment ids out")
    r2d = RaggedToDense(pad_right=True, lib=lib)
    T = max_len + 2
    ref_d = O.ragged_to_dense(ref_c[0], ref_c[1], ref_c[2], T, 50256, True)
    timed("RaggedToDense(pad right)", lambda: r2d.evaluate([c_out[0], c_out[1], c_out[2], np.array([T], np.int32), np.array([50256], np.int32)]),
          4 * (kept + 2 * n) + 5 * n * T + 8 * n, lambda out: same([ref_d[0]], [to_np(out[0])[:k]]), f"T = {T}: input_ids i32 + mask u8")
    tail = FusedEncodeTail(max_length=max_len, lib=lib)
    timed("ovtk_encode_tail_run (Truncate -> CombineSegments -> RaggedToDense x 2)",
          lambda: tail.evaluate([bos_d, (ib, ie, ids), bos_d], seg_ids, truncated=(1,), pad_value=50256, target_dim=T),
          4 * kept + 9 * n * T + 8 * n, lambda out: same([ref_d[0]], [to_np(out[0])[:k]]),
          "one call, one kernel behind the width measurement; input_ids + attention_mask + token_type_ids")
    print(json.dumps({"batch": {"rows": n, "text_bytes": n_c, "ids": n_t, "ids_kept_by_truncate": kept}}))


if __name__ == "__main__":
    main()
