#!/bin/bash
# round 4: conv2 of the small-batch chain on conv_bf6_kernel with one pixel tile per wave (STRIVE_CNN_SMALL_CONV2) vs conv_ws_kernel
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04x}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for v in 0 1; do
  if [ $v = 1 ]; then export STRIVE_CNN_SMALL_CONV2=1; else unset STRIVE_CNN_SMALL_CONV2; fi
  timeout 120 $B --scenes 1 --agents 8 --steps 200 --warmup 20 > $O/bench_1x8_c$v.json 2> $O/bench_1x8_c$v.err < /dev/null
  timeout 120 $B --scenes 4 --agents 16 --steps 60 --warmup 10 > $O/bench_4x16_c$v.json 2> $O/bench_4x16_c$v.err < /dev/null
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt$v -- env STRIVE_HIP_GRAPH=0 $B --scenes 1 --agents 8 --steps 20 --warmup 3 > $O/kt$v.log 2>&1 < /dev/null
  DB=$(find $O/kt$v -name "*.db" 2>/dev/null | head -n 1)
  if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 2> $O/kt_sum$v.err < /dev/null | head -n 40 > $O/1x8_kernel_stats_c$v.txt; grep "conv_\|cnn_tail" $O/1x8_kernel_stats_c$v.txt | cut -c1-150; fi
done
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null | awk '!s[$0]++'
find $O -type f -size +1M -delete
