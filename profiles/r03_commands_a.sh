# r03, first evidence pass (run on the GPU box through gpurun): GPU test-suite, bench lines of the new workloads, kernel trace of
# the closed-loop adversarial closure (planner kernels)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03a
rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^  File\|^Extension" | tail -80) > $O/gpu_tests.log
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_hard -- $B --workload adv --planner hardcode --steps 6 --warmup 2 > $O/kt_hard.log 2>&1
DB=$(find $O/kt_hard -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/adv_hardcode_kernel_stats.txt 2>&1
$B --workload adv --planner hardcode --steps 10 --warmup 3 > $O/bench_line_adv_hardcode.json 2>> $O/bench.err
find $O -type f -size +1M -delete
