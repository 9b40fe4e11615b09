# stream / depth sweep of the timed loop in one gpurun call (measurement only): gpurun_out/r06/x_streams_depth.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06; O=gpurun_out/r06/x_streams_depth.txt; : > $O
for sd in "3 2" "2 1" "2 2" "3 3" "4 2" "4 3" "4 4" "6 4" "3 2"; do
  set -- $sd
  for st in 20 200; do
    python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --streams $1 --depth $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 depth $2 steps $st', d['ms_per_step'], 'steady', d['steady']['ms_per_step'])" >> $O
  done
done
cat $O
