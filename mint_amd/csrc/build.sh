#!/bin/bash
# Build the FACT engine for gfx950 (MI355X). hipcc cross-compiles without a GPU present.
#   ../lib/libfact_hip.so      production library: -fvisibility=hidden, exports exactly include/fact_hip.h
#   ../lib/libfact_hip_dbg.so  the same objects with engine.hip / probe.hip compiled -DFACT_DEBUG_ABI: additionally exports
#                              include/fact_hip_debug.h (single-op entry points, probes, recorder, A/B knobs) - what the
#                              parity tests and bench.py load (mint_amd/_lib.py: FACT_DEBUG_ABI=1)
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" "$OUT/obj"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-result $FACT_EXTRA_FLAGS"
# the flag set is part of the object cache key: a default build after a variant build (FACT_EXTRA_FLAGS=-D...) must not
# link the variant's objects into the shipped library, nor the other way round
if [ "$(cat "$OUT/obj/.flags" 2>/dev/null)" != "$FLAGS" ]; then
  rm -f "$OUT"/obj/*.o
  echo "$FLAGS" > "$OUT/obj/.flags"
fi
pids=()
stale() {  # $1 = object, $2 = source
  [ ! -f "$1" ] || [ "$2" -nt "$1" ] || [ -n "$(find . ../../include -name '*.h' -newer "$1" 2>/dev/null)" ]
}
for f in gemm gemm_big rowops attention engine probe; do
  if stale "$OUT/obj/$f.o" "$f.hip"; then
    $HIPCC $FLAGS -c "$f.hip" -o "$OUT/obj/$f.o" &
    pids+=($!)
  fi
done
for f in engine probe; do
  if stale "$OUT/obj/${f}_dbg.o" "$f.hip"; then
    $HIPCC $FLAGS -DFACT_DEBUG_ABI -c "$f.hip" -o "$OUT/obj/${f}_dbg.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
COMMON="$OUT/obj/gemm.o $OUT/obj/gemm_big.o $OUT/obj/rowops.o $OUT/obj/attention.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o "$OUT/libfact_hip.so" $COMMON "$OUT/obj/engine.o" "$OUT/obj/probe.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o "$OUT/libfact_hip_dbg.so" $COMMON "$OUT/obj/engine_dbg.o" "$OUT/obj/probe_dbg.o"
echo "built $OUT/libfact_hip.so $OUT/libfact_hip_dbg.so"
