#!/bin/bash
# A/B of the one-pass form in one gpurun call: tests, then bench config 2 with --short-path 0 / 1 (twice each), one-stream and three streams.
TAG=${TAG:-a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_short_path.py -m gpu -x -q > $OUT/t_short_path.log 2>&1; tail -3 $OUT/t_short_path.log
for m in 0 1 0 1; do
  for st in 3 1; do
    python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline --short-path $m --streams $st > $OUT/bench_c2_op${m}_s$st.json 2> $OUT/bench_c2_op${m}_s$st.err
    python - <<PY
import json
d=json.loads(open("$OUT/bench_c2_op${m}_s$st.json").read().strip().splitlines()[-1])
op=d["config"].get("short_path")
print("mode $m streams $st", d["value"], d["ms_per_step"], "alone", d["roofline"]["one_stream_kernel_ms"], "overlapped", d["kernel_ms"], "tried/exact", op["calls_tried"], op["calls_that_needed_no_other_kernel"], "traffic", d["roofline"].get("traffic_ratio"))
PY
  done
done
