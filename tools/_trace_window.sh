# kernel trace of the driver's 20-step window (gpurun): when the span / compact kernels of the timed region start and end
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06; rm -rf gpurun_out/trace_w
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_w -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1
python - <<'PY' > gpurun_out/r06/z_window_trace.txt
import csv, glob
f = glob.glob('gpurun_out/trace_w/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'ovtk' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
spans = [r for r in rows if 'lookup_span_kernel' in r['Kernel_Name']]
# priming 8 + warm-up 5, then the 20 timed launches
t0 = int(spans[13]['Start_Timestamp'])
first, last = int(spans[13]['Start_Timestamp']), int(spans[32]['End_Timestamp'])
print('span launches 13..32 (the timed region), times in us relative to the first one\'s start; then every ovtk kernel in that window')
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s >= first - 200000 and s <= last + 100000:
        print('%-28s start %8.1f end %8.1f dur %6.1f' % (r['Kernel_Name'].split('(')[0].replace('void ovtk::', '')[:28], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
find gpurun_out/trace_w -type f -delete; head -70 gpurun_out/r06/z_window_trace.txt
