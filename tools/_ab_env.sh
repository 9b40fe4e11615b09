#!/bin/bash
# A/B of an environment knob in one gpurun call: VAR=name VALS="a b c" [ARGS="--config 2"] bash tools/_ab_env.sh
TAG=${TAG:-h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in $VALS; do
  for st in ${STREAMS:-3 1}; do
    env $VAR=$v python bench.py --steps ${STEPS:-50} --warmup 10 --no-extras --no-cpu-baseline --streams $st $ARGS > $OUT/bench_${VAR}_${v}_s$st.json 2> $OUT/bench_${VAR}_${v}_s$st.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${VAR}_${v}_s$st.json").read().strip().splitlines()[-1])
    op=d["config"].get("short_path")
    print("$VAR=$v streams $st", d["value"], d["ms_per_step"], "alone", d["roofline"]["one_stream_kernel_ms"], "overlapped", d["kernel_ms"], "short tried/exact", op["calls_tried"], op["calls_that_needed_no_other_kernel"], "parity", d.get("parity_prefix_bit_exact"))
except Exception as e:
    print("$VAR=$v streams $st FAILED", e); print(open("$OUT/bench_${VAR}_${v}_s$st.err").read()[-600:])
PY
  done
done
