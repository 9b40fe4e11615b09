"""GPU soak of the row-shard exchange in a one-rank RCCL group: six batches round after round through ShardExchange.submit
(pack kernel) and through lease_wire / submit_wire (the encode writes the wire), every global tensor handed back against
the blocking encode of its batch.    python tools/soak_exchange.py [rounds]"""
import os, sys, socket
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, torch.distributed as dist
from openvino_tokenizers_amd import _lib as L
from openvino_tokenizers_amd.distributed import ShardExchange
from openvino_tokenizers_amd.ops import BPETokenizer, FusedSplitBPE, RegexSplit
from tools.harness import BpeTok
from tools.workloads import TextModel, ragged_rows
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
lib = L.load()
tok = BpeTok.load("gpt2")
fused = FusedSplitBPE(RegexSplit("isolate", lib=lib), BPETokenizer(**tok.attrs, lib=lib))
pat = tok.pattern_u8()
n = 6000
batches, refs = [], []
for i in range(6):
    b, e, c = TextModel(100 + i, "zipf").batch(n, 250 + 30 * i)
    rb, re_ = ragged_rows(n)
    data = [torch.as_tensor(np.array(x), device=dev) for x in (rb, re_, b, e, c)]
    refs.append([t.cpu().numpy().copy() for t in fused.evaluate(data + [pat], tok.consts)])
    batches.append(data)
rounds = int(sys.argv[1])
for mode in ("pack", "wire"):
    ex = ShardExchange(n, len(tok.vocab), dev, lib=lib, stream=torch.cuda.Stream(dev), headroom=1.05)
    if mode == "wire":
        ex.agree_pad(max(len(r[2]) for r in refs))
    bad = 0
    order = []
    def check(done):
        global bad
        if done is None:
            return
        j = order.pop(0)
        got = [t.cpu().numpy() for t in done]
        if not all(np.array_equal(a, g) for a, g in zip(refs[j], got)):
            bad += 1
            print("MISMATCH", mode, j, flush=True)
    for r in range(rounds):
        for k, data in enumerate(batches):
            if mode == "pack":
                res = fused.enqueue(data + [pat], tok.consts)()
                order.append(k)
                check(ex.submit(*res))
            else:
                w = ex.lease_wire()
                def enc(wire, data=data):
                    return fused.enqueue_wire(data + [pat], tok.consts, wire.t, ex.max_rows, wire.pad, ex.id_bytes)()
                enc(w)
                order.append(k)
                check(ex.submit_wire(w, enc))
    for d in ex.flush():
        check(d)
    print(mode, "batches", rounds * len(batches), "bad", bad, "regathers", ex.regathers)
    ex.close()
dist.destroy_process_group()
