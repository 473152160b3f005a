#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04h; mkdir -p $O
CLASSES="gelu'_dgrad ffn1_dgrad qkv_dgrad attention_bwd ln_bwd_dx wgrad_group bias/ln_param_grads" ROUNDS=2 tools/runs/abk.sh "" "--opt lite_stream=2" "--opt lite_stream=1" | tee $O/ab.txt
