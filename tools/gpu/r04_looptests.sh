#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04lt}
rm -rf $O; mkdir -p $O
for sb in 96 0; do
  STRIVE_CNN_SMALL_BATCH=$sb timeout 300 python -m pytest tests/test_loops.py -m gpu -q -s -k "refine_loop_uniform_raster_tight" > $O/gpu_refine_tight_sb$sb.log 2>&1 < /dev/null
  echo "== sb $sb"; grep -n "iterations with a different\|loop refine\|passed\|failed\|^E  .*Assertion" $O/gpu_refine_tight_sb$sb.log | cut -c1-400 | head -10
done
