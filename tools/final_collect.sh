cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/e_gpu_tests.log 2>&1; tail -2 $O/e_gpu_tests.log
for c in 2 3 4 5 pipeline r2d vocab_encoder 1; do timeout 600 python bench.py --config $c > $O/e_bench_c$c.json 2> $O/e_bench_c$c.err; done
for p in qwen2 cl100k o200k deepseek-v3; do timeout 300 python bench.py --config 4 --pattern $p --no-cpu-baseline --no-extras > $O/e_bench_c4_$p.json 2>/dev/null; done
timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/e_bench_c2_200steps.json 2>/dev/null
for r in 1 2 3; do timeout 300 python bench.py --config 2 --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/e_bench_rep$r.json 2>/dev/null; done
bash tools/collect_profiles.sh > $O/e_collect.log 2>&1
CONFIGS="2 3 4" bash tools/collect_inst_counters.sh > $O/e_inst.log 2>&1; cp gpurun_out/inst_counters.csv $O/e_instruction_counters.csv
timeout 900 python tools/soak.py 3000 > $O/e_soak.log 2>&1; tail -3 $O/e_soak.log
timeout 900 python tools/fuzz_span.py 0 40 > $O/e_fuzz_span.log 2>&1; tail -2 $O/e_fuzz_span.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/e_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['ms_per_step'], round(d['value']), r.get('kernel'), r.get('kernel_ms'), r.get('frac'), d.get('parity_prefix_bit_exact'))
    except Exception as e: print(f, 'ERR', e)
PY
