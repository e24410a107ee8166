#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04last}
rm -rf $O; mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "specialised_waves" > $O/gpu_ws_tests.log 2>&1 < /dev/null; tail -n 3 $O/gpu_ws_tests.log | cut -c1-200
