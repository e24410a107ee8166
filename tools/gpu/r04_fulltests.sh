#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04ft}
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1 < /dev/null
tail -n 8 $O/gpu_tests.log | cut -c1-300
