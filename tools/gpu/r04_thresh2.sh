#!/bin/bash
# round 4: small-batch CNN chain pieces at larger batches (12 x 16, 16 x 16, 32 x 16 agents)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04v}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for cfg in "8 16" "16 16" "32 16"; do
  set -- $cfg
  STRIVE_CNN_SMALL_BATCH=0 timeout 120 $B --scenes $1 --agents $2 --steps 30 --warmup 5 > $O/bench_$1x$2_sb0.json 2> $O/bench_$1x$2_sb0.err < /dev/null
  for ts in 1 2 4; do
    STRIVE_CNN_TAIL_S=$ts STRIVE_CNN_SMALL_BATCH=1024 timeout 120 $B --scenes $1 --agents $2 --steps 30 --warmup 5 > $O/bench_$1x$2_sb1024_tail$ts.json 2> $O/bench_$1x$2_sb1024_tail$ts.err < /dev/null
  done
done
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null | awk '!s[$0]++'
