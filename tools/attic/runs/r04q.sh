#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04q; mkdir -p $O
CLASSES="ffn1_dgrad qkv_dgrad ln_bwd_dx wgrad_group bias/ln_param_grads" ROUNDS=2 tools/runs/abk.sh "" "--opt tile192=1 --opt tn_loop=2" "--opt ln_cs=1 --opt tile192=1" "--opt ln_cs=1 --opt tile192=1 --opt tn_loop=2" "--opt tn_loop=2 --opt ln_cs=1" | tee $O/ab.txt
