# Where lookup_span_kernel's time goes, by ablation: copies of the library whose span kernel leaves out 1 = the lookup rounds, 2 = and the
# scan, 3 = and the piece list (-DOVTK_SPAN_ABLATE=n: garbage out, timing only; never the product build), timed by bench.py's one-stream
# leg in ONE call next to the product build.  The switches are not in the product source: they are a patch
# (profiles/r05/experiments/span_ablate.patch) applied to a COPY of csrc/ under csrc/build/ablsrc.  Run on the GPU box:  gpurun --timeout 900 -- 'bash tools/span_ablate.sh'
# (A version with shader-clock timers per section was tried first: the compiler moves the ALU work across the clock reads.)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r04
rm -rf openvino_tokenizers_amd/csrc/build/ablsrc
mkdir -p openvino_tokenizers_amd/csrc/build/ablsrc/openvino_tokenizers_amd/csrc openvino_tokenizers_amd/csrc/build/ablsrc/include
cp openvino_tokenizers_amd/csrc/*.hpp openvino_tokenizers_amd/csrc/*.cpp openvino_tokenizers_amd/csrc/*.inc openvino_tokenizers_amd/csrc/build/ablsrc/openvino_tokenizers_amd/csrc/
cp include/*.h openvino_tokenizers_amd/csrc/build/ablsrc/include/
(cd openvino_tokenizers_amd/csrc/build/ablsrc && patch -p0 < "$OLDPWD/profiles/r05/experiments/span_ablate.patch")
for v in 1 2 3; do
  mkdir -p openvino_tokenizers_amd/csrc/build/abl$v
  if [ ! -f openvino_tokenizers_amd/csrc/build/abl$v/libovtk_amd.so ]; then
    (cd openvino_tokenizers_amd/csrc/build/ablsrc/openvino_tokenizers_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wno-unused-function -DOVTK_SPAN_ABLATE=$v -shared \
       -o ../../../abl$v/libovtk_amd.so api_encode.cpp api_ops.cpp tables.cpp runtime.cpp regex_compile.cpp)
  fi
done
for v in 0 1 2 3; do
  if [ $v = 0 ]; then lib=$PWD/openvino_tokenizers_amd/csrc/build/libovtk_amd.so; else lib=$PWD/openvino_tokenizers_amd/csrc/build/abl$v/libovtk_amd.so; fi
  OVTK_AMD_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r04/abl$v.json 2>/dev/null || true
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r04/abl$v.json').read().strip().splitlines()[-1])
    print('ablation $v: step', d['ms_per_step'], 'ms; kernels alone', d['roofline'].get('one_stream_kernel_ms'))
except Exception as e:
    print('ablation $v failed', e)
PY
done
