set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03b
rm -rf $O; mkdir -p $O
(timeout 600 python -m pytest tests/test_planner.py tests/test_loops.py tests/test_third_party_pins.py -m gpu -q -s -k "planner or closed_loop or iou" 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $O/gpu_tests.log
B="python bench.py --no-cpu-baseline --no-roofline"
$B --workload adv --planner hardcode --steps 10 --warmup 3 > $O/bench_line_adv_hardcode.json 2>> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_hard -- $B --workload adv --planner hardcode --steps 6 --warmup 2 > $O/kt_hard.log 2>&1
DB=$(find $O/kt_hard -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/adv_hardcode_kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
