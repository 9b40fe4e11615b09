import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import RegexSplit
from tools.workloads import TextModel, ragged_rows
lib = L.load()
n = 65536
b, e, c = TextModel(1234, "zipf").batch(n, 512, seed=1000)
rb, re_ = ragged_rows(n)
d = [torch.as_tensor(np.ascontiguousarray(a), device="cuda") for a in (rb, re_, b, e, c)]
for pat in [r"\w+|[^\w\s]+", r"'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]++[\r\n]*|\s*[\r\n]|\s+(?!\S)|\s+(?=xyz)|\s+"]:
    op = RegexSplit("isolate", lib=lib)
    p8 = np.frombuffer(pat.encode(), np.uint8)
    op.evaluate(d + [p8]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): out = op.evaluate(d + [p8])
    torch.cuda.synchronize()
    print("compiled-pattern RegexSplit op", pat[:30], round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms per config-2 batch,", int(out[2].numel()), "pieces")
