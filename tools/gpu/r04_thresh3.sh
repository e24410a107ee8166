#!/bin/bash
# round 4: the shipped thresholds (chain <= 96 samples, tail one sample per workgroup <= 256) against the throughput chain
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04w}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "map_cnn or embed or rollout_golden" > $O/gpu_cnn_tests.log 2>&1 < /dev/null; tail -n 3 $O/gpu_cnn_tests.log
for cfg in "1 8" "6 16" "12 16" "16 16" "32 16"; do
  set -- $cfg
  STRIVE_CNN_TAIL_S=4 STRIVE_CNN_SMALL_BATCH=0 timeout 120 $B --scenes $1 --agents $2 --steps 40 --warmup 5 > $O/bench_$1x$2_r03chain.json 2> $O/bench_$1x$2_r03chain.err < /dev/null
  timeout 120 $B --scenes $1 --agents $2 --steps 40 --warmup 5 > $O/bench_$1x$2_default.json 2> $O/bench_$1x$2_default.err < /dev/null
done
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null | awk '!s[$0]++'
