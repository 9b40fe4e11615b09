#!/bin/bash
# bench.py --short-path 0 / 1 on several configurations, one gpurun call
TAG=${TAG:-j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for cfg in ${CFGS:-2 3 4 pipeline}; do
  for m in ${MODES:-0 1}; do
    python bench.py --config $cfg --steps 50 --warmup 10 --no-extras --no-cpu-baseline --short-path $m > $OUT/bench_c${cfg}_sp$m.json 2> $OUT/bench_c${cfg}_sp$m.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_c${cfg}_sp$m.json").read().strip().splitlines()[-1])
    op=d["config"].get("short_path")
    print("config $cfg short $m", d["value"], d["unit"], d["ms_per_step"], "steady", d.get("value_steady"), "alone", d["roofline"]["one_stream_kernel_ms"], "short tried/exact", op["calls_tried"], op["calls_that_needed_no_other_kernel"])
except Exception as e:
    print("config $cfg short $m FAILED", e); print(open("$OUT/bench_c${cfg}_sp$m.err").read()[-800:])
PY
  done
done
