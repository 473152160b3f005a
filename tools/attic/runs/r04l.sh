#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "partials" 2>&1 | tail -3
