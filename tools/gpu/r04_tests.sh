#!/bin/bash
# round 4: selected GPU tests (K = pytest -k expression), then optionally the whole suite (ALL=1)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04h}
mkdir -p $O
if [ -n "$K" ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$K" > $O/gpu_sel.log 2>&1; tail -25 $O/gpu_sel.log; fi
if [ -n "$ALL" ]; then timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log; fi
