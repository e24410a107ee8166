#!/bin/bash
# round 4: where the small-batch CNN chain stops paying (threshold CNN_SMALL_BATCH), and conv2 on conv_bf6_kernel at one scene
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04u}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for cfg in "3 16" "4 16" "6 16" "8 16"; do
  set -- $cfg
  for sb in 0 1024; do
    STRIVE_CNN_SMALL_BATCH=$sb timeout 120 $B --scenes $1 --agents $2 --steps 60 --warmup 10 > $O/bench_$1x$2_sb$sb.json 2> $O/bench_$1x$2_sb$sb.err < /dev/null
  done
done
STRIVE_CONV_WS=0 timeout 120 $B --scenes 1 --agents 8 --steps 200 --warmup 20 > $O/bench_1x8_ws0.json 2> $O/bench_1x8_ws0.err < /dev/null
timeout 120 $B --scenes 1 --agents 8 --steps 200 --warmup 20 > $O/bench_1x8_ws1.json 2> $O/bench_1x8_ws1.err < /dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null | awk '!s[$0]++'
