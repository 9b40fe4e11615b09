mkdir -p gpurun_out/r03/ab4
for rep in 1 2; do
  python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-extras > gpurun_out/r03/ab4/c2_ahead_$rep.json 2>/dev/null
  OVTK_LOOKUP_STRIDED=1 python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-extras > gpurun_out/r03/ab4/c2_strided_$rep.json 2>/dev/null
done
python bench.py --config 4 --steps 30 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r03/ab4/c4_1.json 2>/dev/null
python bench.py --config 3 --steps 30 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r03/ab4/c3_1.json 2>/dev/null
python -m pytest tests/test_bpe_parity.py tests/test_full_size.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -5
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $R/gpurun_out/pmc_q_$c; (cd $R && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_q_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-alone-leg > /dev/null 2>&1); done
cd $R
python - <<PY
import json, glob, pandas as pd
for f in sorted(glob.glob("gpurun_out/r03/ab4/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], r["one_stream_kernel_ms"], r["one_stream_kernel_sum_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_q_%s/**/*counter_collection.csv"%c, recursive=True)
    t=pd.read_csv(f[0]); t=t[t.Kernel_Name.str.contains("ovtk")]
    t=t.sort_values("Dispatch_Id").groupby(t.Kernel_Name.str.slice(11,40)).tail(4)
    print(c, t.groupby(t.Kernel_Name.str.slice(11,40)).Counter_Value.mean().round(0).to_dict())
PY
find gpurun_out/pmc_q_FETCH_SIZE gpurun_out/pmc_q_WRITE_SIZE -type f ! -name "*counter_collection.csv" -delete
