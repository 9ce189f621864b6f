#!/bin/bash
# pre-split weight planes: unit tests, then whole-step parity / determinism with them on, then the bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_env_ab.sh "RSCOTR_WPLANES=1" "RSCOTR_WPLANES=0"
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "presplit" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_determinism_gpu.py -q -m gpu 2>&1 | tail -8
