#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE passes of bench.py
# for every single-GPU configuration; raw output under gpurun_out/, summarised by tools/summarize_profiles.py.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
for cfg in ${CONFIGS:-2 3 4 5 r2d vocab_encoder pipeline}; do
  rm -rf "$OUT/prof_c$cfg" "$OUT/pmc_fetch_c$cfg" "$OUT/pmc_write_c$cfg"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c$cfg" -- \
      python bench.py --config $cfg --no-cpu-baseline --no-extras --no-alone-leg > "$OUT/prof_c$cfg.log" 2>&1
  # the same kernels with the chip to themselves (what bench.py reports as roofline.alone): one stream
  rm -rf "$OUT/prof1_c$cfg"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof1_c$cfg" -- \
      python bench.py --config $cfg --no-cpu-baseline --no-extras --no-alone-leg --streams 1 > "$OUT/prof1_c$cfg.log" 2>&1
  find "$OUT/prof1_c$cfg" -type f ! -name '*kernel_stats.csv' -delete
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c$cfg" -- \
      python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-alone-leg > "$OUT/pmc_fetch_c$cfg.log" 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c$cfg" -- \
      python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-alone-leg > "$OUT/pmc_write_c$cfg.log" 2>&1
  # keep the merge-back small: the stats and counter tables only
  find "$OUT/prof_c$cfg" "$OUT/pmc_fetch_c$cfg" "$OUT/pmc_write_c$cfg" -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete
done
grep -h '^{"metric"' "$OUT"/prof_c*.log | cut -c1-200
