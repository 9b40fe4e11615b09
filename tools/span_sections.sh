# A copy of the library with lookup_span_kernel's section timers, and their shares on config 2 (run on the GPU box).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p openvino_tokenizers_amd/csrc/build/timers gpurun_out/r04
if [ ! -f openvino_tokenizers_amd/csrc/build/timers/libovtk_amd.so ]; then
  (cd openvino_tokenizers_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wno-unused-function -DOVTK_SPAN_TIMERS -shared \
     -o build/timers/libovtk_amd.so api_encode.cpp api_ops.cpp tables.cpp runtime.cpp regex_compile.cpp)
fi
OVTK_AMD_LIB=$PWD/openvino_tokenizers_amd/csrc/build/timers/libovtk_amd.so python tools/span_sections.py | tee gpurun_out/r04/span_sections.txt
