mkdir -p gpurun_out/r03/ab2
for rep in 1 2; do for st in 131072 262144 1048576; do
  python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-extras --memo-store $st > gpurun_out/r03/ab2/c2_${st}_$rep.json 2>/dev/null
done; done
for st in 131072 262144 1048576; do
  python bench.py --config 4 --steps 30 --warmup 10 --no-cpu-baseline --no-extras --memo-store $st > gpurun_out/r03/ab2/c4_${st}_1.json 2>/dev/null
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_e.json 2> gpurun_out/r03/bench_e.err
python -m pytest tests/test_string_tensor.py tests/test_bpe_parity.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" > gpurun_out/r03/gpu_tests_5.log
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r03/ab2/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], r["one_stream_kernel_ms"], r["one_stream_kernel_sum_ms_per_step"], d["config"]["piece_memo"]["store"]["entries"])
    except Exception as e: print(f, "ERR", e)
d=json.load(open("gpurun_out/r03/bench_e.json")); print(d["value"], d["ms_per_step"], d["end_to_end"])
PY
cat gpurun_out/r03/gpu_tests_5.log
