#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "layernorm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -3
CLASSES="gelu'_dgrad ffn1_dgrad qkv_dgrad attention_bwd ln_bwd_dx wgrad_group bias/ln_param_grads" ROUNDS=2 tools/runs/abk.sh "--opt ln_cs=0" "--opt ln_cs=1" "--opt ln_cs=2" | tee $O/ab.txt
