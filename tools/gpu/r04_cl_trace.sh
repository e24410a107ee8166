#!/bin/bash
# round 4: kernel trace of the closed-loop adversarial closure at one scene x 8 agents
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04cl}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- $B --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 10 --warmup 3 > $O/kt.log 2>&1 < /dev/null
DB=$(find $O/kt -name "*.db" 2>/dev/null | head -n 1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 2> $O/kt_sum.err < /dev/null | head -n 45 > $O/cl1x8_kernel_stats.txt; head -n 30 $O/cl1x8_kernel_stats.txt | cut -c1-150; fi
find $O -type f -size +1M -delete
