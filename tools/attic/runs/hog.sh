#!/bin/bash
cd "$(dirname "$0")/../.."
for q in "" 5 6 7 8; do
  for opts in "" "aux_stream=0" "aux_stream=0 side_stream=0"; do
    echo -n "GPU_MAX_HW_QUEUES=${q:-default} [$opts] "
    if [ -z "$q" ]; then HOG_NS=0,8 python tools/cu_hog_probe.py $opts 2>&1 | grep cu_hog | sed 's/cu_hog_probe[^:]*://'
    else GPU_MAX_HW_QUEUES=$q HOG_NS=0,8 python tools/cu_hog_probe.py $opts 2>&1 | grep cu_hog | sed 's/cu_hog_probe[^:]*://'; fi
  done
done
