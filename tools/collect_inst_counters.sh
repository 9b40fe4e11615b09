#!/bin/bash
# Runs on the MI355X box (through gpurun): instruction and activity counters of the encode kernels, one stream, for the
# VALU-issue analysis in DESIGN.md section 6.  Writes gpurun_out/inst_counters.csv (kernel, config, counter, mean per launch).
#   gpurun --timeout 900 -- 'bash tools/collect_inst_counters.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
echo "config,kernel,counter,mean_per_launch,launches" > "$OUT/inst_counters.csv"
for cfg in ${CONFIGS:-2 3 4}; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    rm -rf "$OUT/pmc_inst"
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_inst" -- \
        python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-alone-leg --streams 1 > "$OUT/pmc_inst.log" 2>&1
    python - "$cfg" "$OUT" <<'PY'
import csv, glob, collections, sys
cfg, out = sys.argv[1], sys.argv[2]
fs = glob.glob(out + '/pmc_inst/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'ovtk' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].replace('void ', '').split('(')[0].replace('ovtk::', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out + '/inst_counters.csv', 'a') as fh:
    for (k, c), v in sorted(acc.items()):
        v = v[-4:]   # the steady state: the last four launches (bench.py's priming pass comes first)
        fh.write(f'{cfg},"{k}",{c},{sum(v) / len(v):.0f},{len(v)}\n')
PY
  done
done
rm -rf "$OUT/pmc_inst"
cat "$OUT/inst_counters.csv" | head -60
