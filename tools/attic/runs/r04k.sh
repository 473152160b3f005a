#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04k; mkdir -p $O
CLASSES="adam+shadows wgrad_group ffn1_dgrad attention_bwd" ROUNDS=2 tools/runs/abk.sh "" "--fuse-optimizer 0" "--opt adam_hold=0" "--opt aux_stream=0" | tee $O/ab.txt
