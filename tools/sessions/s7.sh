#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s7
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_planes.py tests/test_pieces_entry.py tests/test_features.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
