#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04f; mkdir -p $O
CLASSES="ln_bwd_dx bias/ln_param_grads wgrad_group attention_bwd" ROUNDS=2 tools/runs/abk.sh "" "--opt ln_split=0" "--opt tn_loop=2" "--opt ln_split=0 --opt tn_loop=2" | tee $O/ab.txt
