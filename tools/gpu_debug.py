import sys
import numpy as np
sys.path.insert(0, ".")
from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.ops import RegexSplit, BPETokenizer, FusedSplitBPE
from tools.make_tokenizers import load_tokenizer, GPT2_PATTERN
from tools.workloads import ragged_rows
from oracle import oracle as O
lib = L.load(sys.argv[1] if len(sys.argv) > 1 else None)
pat = np.frombuffer(GPT2_PATTERN.encode(), np.uint8)
ors = O.RegexSplit(GPT2_PATTERN, "isolate")
rs = RegexSplit("isolate", lib=lib)
for strings in (["hello world"], ["Hello world! it's 123  ok\n"], ["a"], ["ab cd", "x y z"]):
    b, e, c = O.pack_strings(strings)
    rb, re2 = ragged_rows(len(b))
    ref = ors(rb, re2, b, e, c)
    got = rs.evaluate([rb, re2, b, e, c, pat])
    print(strings)
    for i in range(4):
        print("  ref", ref[i].tolist(), "\n  got", np.asarray(got[i]).tolist())
