# instruction / activity counters of TrieTokenizer's kernels (gpurun): gpurun_out/r06/n_trie_counters.csv
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT/r06
echo "kernel,counter,mean_per_launch,launches" > $OUT/r06/n_trie_counters.csv
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/pmc_t
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_t -- python tools/ops_timing.py --reps 4 > $OUT/pmc_t.log 2>&1
  python - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/pmc_t/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rie' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].replace('void ', '').split('(')[0].replace('ovtk::', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out + '/r06/n_trie_counters.csv', 'a') as fh:
    for (k, c), v in sorted(acc.items()):
        fh.write(f'"{k}",{c},{sum(v) / len(v):.0f},{len(v)}\n')
PY
done
find $OUT/pmc_t -type f -delete; cat $OUT/r06/n_trie_counters.csv
