# The round's measurement call (one gpurun call): TAG=<letter> bash tools/final_collect.sh -> gpurun_out/r06/<letter>_*
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06; T=${TAG:-e}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; tail -2 $O/${T}_gpu_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --short-path 0 > $O/${T}_bench_c2_four_launches.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_c2_driver_form.json 2>/dev/null; timeout 600 python bench.py --steps 20 --warmup 5 --short-path 0 > $O/${T}_bench_c2_driver_form_four_launches.json 2>/dev/null
for c in 2 3 4 5 pipeline r2d vocab_encoder 1; do timeout 600 python bench.py --config $c > $O/${T}_bench_c$c.json 2> $O/${T}_bench_c$c.err; done
for p in qwen2 cl100k o200k deepseek-v3; do timeout 300 python bench.py --config 4 --pattern $p --no-cpu-baseline --no-extras > $O/${T}_bench_c4_$p.json 2>/dev/null; done
timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/${T}_bench_c2_200steps.json 2>/dev/null
for r in 1 2 3; do timeout 300 python bench.py --config 2 --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/${T}_bench_rep$r.json 2>/dev/null; done
bash tools/collect_profiles.sh > $O/${T}_collect.log 2>&1
CONFIGS="2 3 4" bash tools/collect_inst_counters.sh > $O/${T}_inst.log 2>&1; cp gpurun_out/inst_counters.csv $O/${T}_instruction_counters.csv
timeout 300 python tools/ops_timing.py > $O/${T}_ops_timing.jsonl 2> $O/${T}_ops_timing.err; cut -c1-150 $O/${T}_ops_timing.jsonl
rm -rf gpurun_out/prof_ops; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ops -- python tools/ops_timing.py --reps 20 > gpurun_out/prof_ops.log 2>&1
cp $(find gpurun_out/prof_ops -name '*kernel_stats.csv' | head -1) $O/${T}_ops_kernel_stats.csv; find gpurun_out/prof_ops -type f ! -name '*kernel_stats.csv' -delete
timeout 900 python tools/soak.py 3000 > $O/${T}_soak.log 2>&1; tail -3 $O/${T}_soak.log
timeout 600 python tools/soak.py 1000 "" 20000 2 > $O/${T}_soak_short2.log 2>&1; tail -2 $O/${T}_soak_short2.log
timeout 900 python tools/fuzz_span.py 0 40 > $O/${T}_fuzz_span.log 2>&1; tail -2 $O/${T}_fuzz_span.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/%s_bench_*.json' % __import__('os').environ.get('TAG','e'))):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['ms_per_step'], round(d['value']), r.get('kernel'), r.get('kernel_ms'), r.get('frac'), d.get('parity_prefix_bit_exact'))
    except Exception as e: print(f, 'ERR', e)
PY
