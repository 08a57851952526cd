#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s21
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_planes.py -x -q -m gpu -k "bf16_feature_storage or forward_only or q32_feature" > $O/diag.log 2>&1; echo "rc=$?"; tail -12 $O/diag.log
