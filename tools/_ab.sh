mkdir -p gpurun_out/r03/ab
for rep in 1 2; do for v in nt nont; do for st in 0 131072 262144; do
  L=""; [ $v = nont ] && L="--lib tools/build/libovtk_nont.so"
  python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-extras --memo-store $st $L > gpurun_out/r03/ab/c2_${v}_${st}_$rep.json 2>/dev/null
done; done; done
for v in nt nont; do for st in 131072 262144; do
  L=""; [ $v = nont ] && L="--lib tools/build/libovtk_nont.so"
  python bench.py --config 4 --steps 30 --warmup 10 --no-cpu-baseline --no-extras --memo-store $st $L > gpurun_out/r03/ab/c4_${v}_${st}_1.json 2>/dev/null
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r03/ab/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], r["one_stream_kernel_ms"], r["one_stream_kernel_sum_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
