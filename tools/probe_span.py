"""Diagnostic: per-wave clocks of lookup_span_kernel.  Needs the -DOVTK_PROBE build (tools/probe_merge.py says how).
Per wave (wall_clock64, 10 ns ticks): 0 start, 1 out of the lookup loop, 2 misses looked up in the store (the short path), 3 records written;
slot 8: where the wave ran (HW_ID, XCC), slot 9: the bytes of its rows.  (Stamps 4-7 belonged to the one-launch form with a look-back that round 6
measured and dropped: profiles/r06/a_one_pass_*.)"""
import ctypes as C, sys, argparse
from pathlib import Path
from types import SimpleNamespace
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvino_tokenizers_amd import _lib as L
import bench
lib = L.load(ROOT / "tools" / "build" / "libovtk_probe.so")
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="2"); ap.add_argument("--short-path", type=int, default=1)
a = ap.parse_args()
args = SimpleNamespace(config=a.config, tokenizer="gpt2", text="zipf", rows=65536 if a.config != "4" else 131072, bytes=512, batches=4, no_memo=False, pattern=None, cache_capacity=None, memo_learn=None)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.check(lib, lib.ovtk_set_short_path(a.short_path))
wl = bench.make_workload(args, lib, dev, 0)
for i in range(16):
    wl.step(i)
torch.cuda.synchronize()
out = np.zeros((8192, 12), np.uint64)
names = ["start", "loop done", "resolved", "records", "published", "prefix known", "copied", "done"]
for i in range(4):
    lib.ovtk_debug_probe(None, 1)
    wl.step(i)
    torch.cuda.synchronize()
    lib.ovtk_debug_probe(out.ctypes.data_as(C.POINTER(C.c_ulonglong)), 0)
    ts = out.astype(np.int64)
    live = ts[:, 0] > 0
    t0 = ts[live, 0].min()
    print(f"batch {i}: waves {live.sum()}")
    for k, nm in enumerate(names):
        m = live & (ts[:, k] > 0)
        if not m.any():
            print(f"  {k} {nm}: -")
            continue
        v = (ts[m, k] - t0) / 100.0
        print(f"  {k} {nm:13s} n={m.sum():5d} min {v.min():6.1f} p10 {np.percentile(v, 10):6.1f} p50 {np.median(v):6.1f} p90 {np.percentile(v, 90):6.1f} max {v.max():6.1f} us")
    m = live & (ts[:, 5] > 0) & (ts[:, 4] > 0)
    if m.any():
        d = (ts[m, 5] - ts[m, 4]) / 100.0
        print(f"  look-back wait: p50 {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} max {d.max():.1f}")
        d = (ts[m, 6] - ts[m, 5]) / 100.0
        print(f"  copy: p50 {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} max {d.max():.1f}")
        d = (ts[m, 2] - ts[m, 1]) / 100.0
        print(f"  resolve: p50 {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} max {d.max():.1f}")
# the chain: a wave's prefix cannot be known before every wave in front of it has published
pub = (ts[:, 4] - t0) / 100.0
known = (ts[:, 5] - t0) / 100.0
liv = np.flatnonzero(live)
runmax = np.maximum.accumulate(pub[liv])
print("running max of 'published' at waves 1/8 .. 8/8:", [round(float(runmax[min(len(liv) - 1, k * len(liv) // 8 - 1)]), 1) for k in range(1, 9)])
lag = known[liv][1:] - np.maximum(runmax[:-1], pub[liv][1:])
ok = ts[liv, 5][1:] > 0
print("prefix known - max(published of everyone in front, own published): p50 %.1f p90 %.1f max %.1f" % (np.median(lag[ok]), np.percentile(lag[ok], 90), lag[ok].max()))
res = (ts[:, 2] - ts[:, 1]) / 100.0
slow = liv[np.argsort(res[liv])[-8:]]
print("slowest resolves (wave, us, loop done at):", [(int(wv), round(float(res[wv]), 1), round(float((ts[wv, 1] - t0) / 100.0), 1)) for wv in slow])
pubd = (ts[:, 4] - ts[:, 3]) / 100.0
slow = liv[np.argsort(pubd[liv])[-8:]]
print("slowest publish (leaders' look-back) (wave, us):", [(int(wv), round(float(pubd[wv]), 1)) for wv in slow])
T = (ts[:, :8] - t0) / 100.0
late = liv[pub[liv] > runmax.min() + 20]
first_late = [int(wv) for wv in liv if pub[wv] > 70][:6]
print("first waves that published after 70 us:", first_late)
for wv in first_late + [int(liv[len(liv) // 2 + 5]), int(liv[-3])]:
    print(f"  wave {wv} (group {wv // 64}, in group {wv % 64}):", " ".join(f"{nm}={T[wv, k]:.1f}" for k, nm in enumerate(names)))
# per group: when its last wave reported (records stamp of the slowest), when its leader published
grp_last = {}
for wv in liv:
    g = int(wv) // 64
    grp_last[g] = max(grp_last.get(g, 0.0), float(T[wv, 3]))
gl = [grp_last[g] for g in sorted(grp_last)]
print("per group, 'records' of its slowest wave:", [round(x, 1) for x in gl])
# where the waves ran, and how long their lookup loops took by place
hw = ts[:, 8]
xcc = (hw >> 32) & 0xF
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
place = xcc * 1000 + se * 100 + sh * 50 + cu
dur = (ts[:, 1] - ts[:, 0]) / 100.0
live = ts[:, 0] > 0
w_idx0 = np.arange(len(dur))
print("bytes per wave: min %d p50 %d max %d" % (ts[live, 9].min(), np.median(ts[live, 9]), ts[live, 9].max()))
print("corr(loop time, bytes) = %.3f" % np.corrcoef(dur[live], ts[live, 9])[0, 1])
import collections
per_cu = collections.Counter(place[live].tolist())
cnts = np.array(list(per_cu.values()))
print("CUs used %d; waves per CU: min %d p50 %d max %d; histogram %s" % (len(per_cu), cnts.min(), np.median(cnts), cnts.max(), dict(collections.Counter(cnts.tolist()))))
for n in sorted(set(cnts.tolist())):
    sel = np.array([per_cu[p] == n for p in place.tolist()]) & live
    print(f"  waves on CUs that hold {n} waves: {sel.sum()}; loop time p50 {np.median(dur[sel]):.1f} p90 {np.percentile(dur[sel], 90):.1f} max {dur[sel].max():.1f}")
per_simd = collections.Counter((place[live] * 4 + simd[live]).tolist())
sc = np.array([per_simd[p] for p in (place * 4 + simd).tolist()])
for n in sorted(set(sc[live].tolist())):
    sel = (sc == n) & live
    print(f"  waves on SIMDs that hold {n} waves: {sel.sum()}; loop time p50 {np.median(dur[sel]):.1f} p90 {np.percentile(dur[sel], 90):.1f} max {dur[sel].max():.1f}")
slowest = np.argsort(np.where(live, dur, 0))[-24:]
print("slowest lookup loops (wave, wave in block, SIMD, CU place, us):", [(int(wv), int(wv) % 4, int(simd[wv]), int(place[wv]), round(float(dur[wv]), 1)) for wv in slowest])
for k in range(4):
    sel = live & (w_idx0 % 4 == k)
    print(f"  wave-in-block {k}: loop p50 {np.median(dur[sel]):.1f} p90 {np.percentile(dur[sel], 90):.1f} p99 {np.percentile(dur[sel], 99):.1f} max {dur[sel].max():.1f}; SIMDs {sorted(set(simd[sel].tolist()))}")
for x in range(8):
    sel = (xcc == x) & live
    if sel.any():
        print(f"  XCC {x}: waves {sel.sum()} loop p50 {np.median(dur[sel]):.1f} p90 {np.percentile(dur[sel], 90):.1f} max {dur[sel].max():.1f}")
w_idx = np.arange(len(dur))
for q in range(8):
    sel = live & (w_idx * 8 // max(1, live.sum()) == q)
    if sel.any():
        print(f"  waves {q}/8 of the grid: loop p50 {np.median(dur[sel]):.1f} max {dur[sel].max():.1f}")
t, x = C.c_int64(), C.c_int64()
lib.ovtk_short_path_stats(C.byref(t), C.byref(x))
print("one pass tried / exact", t.value, x.value)
