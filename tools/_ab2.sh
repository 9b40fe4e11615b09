cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for v in prev main; do
  if [ $v = main ]; then lib=$PWD/openvino_tokenizers_amd/csrc/build/libovtk_amd.so; else lib=$PWD/openvino_tokenizers_amd/csrc/build/$v/libovtk_amd.so; fi
  OVTK_AMD_LIB=$lib timeout 300 python bench.py --config 2 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r05/tmpR.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05/tmpR.json').read().strip().splitlines()[-1])
s=d.get('stress') or {}
print('$v', d['ms_per_step'], d['roofline']['one_stream_kernel_ms'], {k:(round(v['value']/1000,1) if isinstance(v,dict) else v) for k,v in s.items()})
PY
done
