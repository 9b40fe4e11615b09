#!/bin/bash
# the default bench line of config 2 with its stress legs, short path 1 and 0, one gpurun call
TAG=${TAG:-n}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for m in 1 0; do
  python bench.py --no-cpu-baseline --short-path $m > $OUT/bench_c2_full_sp$m.json 2> $OUT/bench_c2_full_sp$m.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_c2_full_sp$m.json").read().strip().splitlines()[-1])
print("short $m", d["value"], d["ms_per_step"], "steady", d["value_steady"])
for k,v in (d.get("stress") or {}).items(): print("   ", k, v["value"], v["ms_per_step"])
PY
done
