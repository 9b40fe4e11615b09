#!/bin/bash
# What an LDS-resident table would cost in occupancy (DESIGN.md 6, round 3 "tables staged in LDS"): the encode kernels with a dummy
# __shared__ array of the size such a table needs, everything else unchanged -- the time they then take is a floor for any version
# that really uses the table (the probes it would save are measured separately: tools/probe_merge.py for merge_kernel's initial
# pair lookups).  Build here (hipcc cross-compiles), run on the GPU box:
#   bash tools/lds_table_ablation.sh build;   gpurun -- 'bash tools/lds_table_ablation.sh run'
set -u
cd "$(dirname "$0")/.."
SRC="api_encode.cpp api_ops.cpp tables.cpp runtime.cpp regex_compile.cpp"
VARIANTS="merge16:-DOVTK_MERGE_PAD_LDS=16384 merge36:-DOVTK_MERGE_PAD_LDS=36864 lookup7:-DOVTK_LOOKUP_PAD_LDS=7168"
if [ "${1:-run}" = build ]; then
  mkdir -p tools/build
  for v in $VARIANTS; do
    (cd openvino_tokenizers_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wno-unused-function ${v#*:} -shared \
       -o ../../tools/build/libovtk_${v%%:*}.so $SRC)
  done
  exit 0
fi
mkdir -p gpurun_out/r03/lds
for rep in 1 2; do
  python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-extras > gpurun_out/r03/lds/base_$rep.json 2>/dev/null
  for v in $VARIANTS; do
    python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-extras --lib tools/build/libovtk_${v%%:*}.so > gpurun_out/r03/lds/${v%%:*}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r03/lds/*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], "ms_per_step", d["ms_per_step"], "one-stream kernels", r["one_stream_kernel_ms"])
PY
