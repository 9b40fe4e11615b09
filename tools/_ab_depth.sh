cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06; O=gpurun_out/r06/x_depth_ab.txt; : > $O
for rep in 1 2 3; do for sd in "3 2" "3 3" "4 4"; do set -- $sd
  python bench.py --steps 20 --warmup 5 --no-extras --streams $1 --depth $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep streams $1 depth $2 steps 20 (cpu baseline leg in front)', d['ms_per_step'], 'steady', d['steady']['ms_per_step'])" >> $O
done; done; cat $O
