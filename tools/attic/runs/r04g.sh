#!/bin/bash
# which co-runner stretches the dgrad chain?  timing-only ablations (debug option "skip": results are wrong while set)
cd "$(dirname "$0")/../.."
O=gpurun_out/r04g; mkdir -p $O
CLASSES="gelu'_dgrad ffn1_dgrad qkv_dgrad out_proj_dgrad+heads attention_bwd ln_bwd_dx wgrad_group bias/ln_param_grads" ROUNDS=1 tools/runs/abk.sh "" "--opt skip=2" "--opt skip=1" "--opt skip=3" "--opt wgrad_parts=1" | tee $O/ab.txt
