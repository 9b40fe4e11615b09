"""Quick GPU sanity run (used through gpurun while developing): parity vs oracle + a rough timing."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import RegexSplit, BPETokenizer, FusedSplitBPE
from tools.make_tokenizers import load_tokenizer
from tools.workloads import TextModel, ragged_rows
from oracle import oracle as O

lib = L.load()
print("device:", lib.ovtk_device_name())
name = sys.argv[1] if len(sys.argv) > 1 else "gpt2_small"
t = load_tokenizer(name)
pack = O.pack_strings
vb, ve, vc = pack(t["vocab"]); lb, le, lc = pack([m[0] for m in t["merges"]]); rb_, re_, rc = pack([m[1] for m in t["merges"]])
ab, ae, ac = pack(list(t["added"].keys())); aid = np.array(list(t["added"].values()), np.int32)
consts = [vb, ve, vc, lb, le, lc, rb_, re_, rc, ab, ae, ac, aid]
pat = np.frombuffer(t["pattern"].encode(), np.uint8)
m = TextModel(7, "zipf")
ors = O.RegexSplit(t["pattern"], "isolate"); obpe = O.BPETokenizer(t["vocab"], t["merges"], t["added"], **t["attrs"])
f = FusedSplitBPE(RegexSplit("isolate"), BPETokenizer(**t["attrs"]))
rs = RegexSplit("isolate"); bp = BPETokenizer(**t["attrs"])
for (B, T) in [(32, 128), (2048, 512)]:
    b, e, c = m.batch(B, T)
    rb, re2 = ragged_rows(B)
    sp = ors(rb, re2, b, e, c); ref = obpe(*sp[:5])
    got = f.evaluate([rb, re2, b, e, c, pat], consts)
    print(B, T, "fused host ok:", all(np.array_equal(x, y) for x, y in zip(ref, got)))
    gsp = rs.evaluate([rb, re2, b, e, c, pat])
    print(B, T, "split ok:", all(np.array_equal(x, y) for x, y in zip(sp[:4], gsp[:4])))
    g2 = bp.evaluate(list(gsp[:5]) + consts)
    print(B, T, "pieces ok:", all(np.array_equal(x, y) for x, y in zip(ref, g2)))
    dev = [torch.as_tensor(x, device="cuda") for x in (rb, re2, b, e, c)]
    gd = f.evaluate(dev + [pat], consts)
    print(B, T, "fused device ok:", all(np.array_equal(x, y.cpu().numpy()) for x, y in zip(ref, gd)))
# timing at config 2 scale
B, T = 65536, 512
b, e, c = m.batch(B, T)
rb, re2 = ragged_rows(B)
dev = [torch.as_tensor(x, device="cuda") for x in (rb, re2, b, e, c)]
lib.ovtk_profile_enable(1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    gd = f.evaluate(dev + [pat], consts)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"cfg2 fused: {dt*1e3:.2f} ms  -> {len(c)/dt/1e6:.1f} MB/s, tokens {len(gd[2])}")
buf = (__import__('ctypes').c_char * 4096)()
lib.ovtk_profile_dump(buf, 4096); print(buf.value.decode())
t0 = time.time(); sp = ors(rb[:2048], re2[:2048], b[:2048], e[:2048], c); ref = obpe(*sp[:5]); dt = time.time() - t0
print("oracle 2048 rows:", dt, "s", e[2047]/dt/1e6, "MB/s")
ids = gd[2].cpu().numpy(); gb = gd[0].cpu().numpy(); ge = gd[1].cpu().numpy()
print("prefix parity on 2048 rows:", np.array_equal(ref[2], ids[:ge[2047]]))
