for st in 2 3 4; do for dp in 2 3; do
python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-alone-leg --streams $st --depth $dp > /tmp/b.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("streams $st depth $dp", d["value"], d["ms_per_step"], "steady", d["value_steady"], d["steady"]["ms_per_step"], d["steady"]["completion_interval_ms"])
PY
done; done
