run() { timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['kernel_ms'], (d['roofline'].get('alone') or {}).get('kernel_ms'))"; }
run
run --streams 1
run --config 4
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
