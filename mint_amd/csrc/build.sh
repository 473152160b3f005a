#!/bin/bash
# Build libfact_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU present.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" "$OUT/obj"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $FACT_EXTRA_FLAGS"
# the flag set is part of the object cache key: a default build after a variant build (FACT_EXTRA_FLAGS=-D...) must not
# link the variant's objects into the shipped library, nor the other way round
if [ "$(cat "$OUT/obj/.flags" 2>/dev/null)" != "$FLAGS" ]; then
  rm -f "$OUT"/obj/*.o
  echo "$FLAGS" > "$OUT/obj/.flags"
fi
pids=()
for f in gemm gemm_big rowops attention engine probe; do
  if [ ! -f "$OUT/obj/$f.o" ] || [ "$f.hip" -nt "$OUT/obj/$f.o" ] || [ -n "$(find . ../../include -name '*.h' -newer "$OUT/obj/$f.o" 2>/dev/null)" ]; then
    $HIPCC $FLAGS -c "$f.hip" -o "$OUT/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfact_hip.so" "$OUT"/obj/gemm.o "$OUT"/obj/gemm_big.o "$OUT"/obj/rowops.o "$OUT"/obj/attention.o "$OUT"/obj/engine.o "$OUT"/obj/probe.o
echo "built $OUT/libfact_hip.so"
