# A/B in one gpurun call: build/prev (the commit before) against build/ (this tree); config from $CFG, extras with $EXTRAS=1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do
for v in ${VARIANTS:-prev main}; do
  if [ $v = main ]; then lib=$PWD/openvino_tokenizers_amd/csrc/build/libovtk_amd.so; else lib=$PWD/openvino_tokenizers_amd/csrc/build/$v/libovtk_amd.so; fi
  extras="--no-extras"; [ "${EXTRAS:-0}" = 1 ] && [ $rep = 1 ] && extras=""
  OVTK_AMD_LIB=$lib timeout 600 python bench.py --config ${CFG:-2} --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline $extras > gpurun_out/r05/tmpR.json 2>gpurun_out/r05/tmpR.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05/tmpR.json').read().strip().splitlines()[-1])
s=d.get('stress') or {}
print('$v', d['ms_per_step'], d['roofline'].get('one_stream_kernel_ms'), d['config'].get('piece_memo'), {k:(round(v['value']/1000,1) if isinstance(v,dict) else v) for k,v in s.items()})
PY
done; done
