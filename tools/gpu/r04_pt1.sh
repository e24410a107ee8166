#!/bin/bash
# round 4: conv3 / conv4 of the small-batch chain with one pixel tile per wave (STRIVE_CNN_SMALL_PT1): bench + kernel durations
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04t}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for v in 0 1; do
  if [ $v = 1 ]; then export STRIVE_CNN_SMALL_PT1=1; else unset STRIVE_CNN_SMALL_PT1; fi
  timeout 120 $B --scenes 1 --agents 8 --steps 200 --warmup 20 > $O/bench_1x8_pt$v.json 2> $O/bench_1x8_pt$v.err < /dev/null
  timeout 120 $B --scenes 2 --agents 12 --steps 100 --warmup 10 > $O/bench_2x12_pt$v.json 2> $O/bench_2x12_pt$v.err < /dev/null
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt$v -- env STRIVE_HIP_GRAPH=0 $B --scenes 1 --agents 8 --steps 20 --warmup 3 > $O/kt$v.log 2>&1 < /dev/null
  DB=$(find $O/kt$v -name "*.db" 2>/dev/null | head -n 1)
  if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 2> $O/kt_sum$v.err < /dev/null | head -n 40 > $O/1x8_kernel_stats_pt$v.txt; grep "conv_bf6" $O/1x8_kernel_stats_pt$v.txt | cut -c1-150; fi
done
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null
find $O -type f -size +1M -delete
