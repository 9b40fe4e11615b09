cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r04/pcs
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace -d gpurun_out/r04/pcs/st -o st --output-format csv -- python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-extras --sync > gpurun_out/r04/pcs/st.log 2>&1; echo "stochastic rc=$?"; tail -5 gpurun_out/r04/pcs/st.log
ls -la gpurun_out/r04/pcs/st 2>/dev/null | head
if ! ls gpurun_out/r04/pcs/st/*pc_sampling* >/dev/null 2>&1; then
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace -d gpurun_out/r04/pcs/ht -o ht --output-format csv -- python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-extras --sync > gpurun_out/r04/pcs/ht.log 2>&1; echo "host_trap rc=$?"; tail -5 gpurun_out/r04/pcs/ht.log
ls -la gpurun_out/r04/pcs/ht | head
fi
# keep the merge small: aggregate by instruction
python - <<'PY'
import glob, csv, collections, os
for f in glob.glob('gpurun_out/r04/pcs/*/*pc_sampling*.csv'):
    print(f, os.path.getsize(f))
    cnt = collections.Counter()
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        for row in rd:
            cnt[(row.get('Instruction',''), row.get('Instruction_Comment',''))] += 1
    print(cols)
    with open(f.replace('.csv', '_agg.txt'), 'w') as out:
        for (ins, com), n in cnt.most_common():
            out.write(f"{n}\t{ins}\t{com}\n")
    os.remove(f)
PY
du -sh gpurun_out/r04/pcs
