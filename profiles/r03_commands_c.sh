# r03: the device planner alone (tools/planner_bench.py) under rocprofv3, then the closed-loop closure
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03c
rm -rf $O; mkdir -p $O
python tools/planner_bench.py 10 > $O/planner_alone.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python tools/planner_bench.py 5 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB | head -8 > $O/planner_kernel_stats.txt 2>&1
(timeout 600 python -m pytest tests/test_planner.py tests/test_loops.py -m gpu -q -s -k "planner or closed_loop" 2>&1 | grep -v "^  File\|^Extension" | tail -12) > $O/gpu_tests.log
B="python bench.py --no-cpu-baseline --no-roofline"
$B --workload adv --planner hardcode --steps 10 --warmup 3 > $O/bench_line_adv_hardcode.json 2>> $O/bench.err
$B --workload adv --steps 10 --warmup 3 > $O/bench_line_adv_ego.json 2>> $O/bench.err
find $O -type f -size +1M -delete
