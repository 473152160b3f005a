#!/bin/bash
# Build a variant of the library with extra compile flags for ONE translation unit and link it beside the shipped objects:
#   tools/build_variant.sh w12 gemm_big -DFACT_EXPERIMENTAL_W12      -> tools/bin/libfact_w12.so
#   tools/build_variant.sh pm2 gemm_big -DBIG_DMA_IN_MFMA=2
# tools/bin/ is git-ignored but travels to the GPU box; A/B against the tree's library with tools/attic/ab_libs.sh, or load it
# with FACT_LIB=tools/bin/libfact_w12.so (mint_amd/_lib.py) for tools/bench_r2.py.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; unit=$2; shift 2
mkdir -p $R/tools/bin /tmp/fact_variant
[ -f $R/mint_amd/lib/obj/gemm.o ] || $R/mint_amd/csrc/build.sh
# (the variant is a test / bench build: engine.hip / probe.hip with the debug surface, mint_amd/csrc/build.sh)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-result -DFACT_DEBUG_ABI "$@" -c $R/mint_amd/csrc/$unit.hip -o /tmp/fact_variant/$unit.$name.o
objs=""
for f in gemm gemm_big rowops attention engine_dbg probe_dbg; do
  if [ ${f%_dbg} = $unit ]; then objs="$objs /tmp/fact_variant/$unit.$name.o"; else objs="$objs $R/mint_amd/lib/obj/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$R/mint_amd/csrc/exports.map -o $R/tools/bin/libfact_$name.so $objs
echo "built tools/bin/libfact_$name.so"
