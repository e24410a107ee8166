# r03: training step with the LDS-staged weight-gradient kernel vs the implicit-GEMM form (STRIVE_WGRAD_IGEMM=1), same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03aa
rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_training.py tests/test_gpu_parity.py -m gpu -q -s 2>&1 | tail -6) > $O/train_tests.log
B="python bench.py --no-cpu-baseline --no-roofline --workload train"
STRIVE_WGRAD_ATOMICS=1 $B --steps 10 --warmup 3 > $O/bench_line_train_igemm.json 2>> $O/bench.err
$B --steps 10 --warmup 3 > $O/bench_line_train_tile.json 2>> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 3 --warmup 1 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB | head -48 > $O/train_kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
