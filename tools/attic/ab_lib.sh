#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <other.so> [bench flags]   (alternates other / current, 3 rounds)
R=$(pwd); other=$1; shift
cp $R/mint_amd/lib/libfact_hip.so /tmp/cur.so
for i in 1 2 3; do
  for which in other cur; do
    if [ $which = other ]; then cp $other $R/mint_amd/lib/libfact_hip.so; else cp /tmp/cur.so $R/mint_amd/lib/libfact_hip.so; fi
    ms=$(cd $R && timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
    echo "$which : $ms ms"
  done
done
cp /tmp/cur.so $R/mint_amd/lib/libfact_hip.so
