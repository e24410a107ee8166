# r03: full GPU suite + headline bench + train bench at HEAD (after the training-backward work)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03k2
rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/gpu_tests.log
python bench.py > $O/bench_line_default.json 2>> $O/bench.err
B="python bench.py --no-cpu-baseline --no-roofline --workload train"
$B --steps 10 --warmup 3 > $O/bench_line_train.json 2>> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 3 --warmup 1 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB | head -60 > $O/train_kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
