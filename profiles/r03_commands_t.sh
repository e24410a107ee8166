# r03 (HEAD): training-step line + its parity tests
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03x3
rm -rf $O; mkdir -p $O
(timeout 600 python -m pytest tests/test_training.py -m gpu -q -s 2>&1 | tail -5) > $O/train_tests.log
B="python bench.py --no-cpu-baseline --no-roofline --workload train"
$B --steps 10 --warmup 3 > $O/bench_line_train.json 2>> $O/bench.err
$B --steps 10 --warmup 3 > $O/bench_line_train_2.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-roofline > $O/bench_line_default.json 2>> $O/bench.err
