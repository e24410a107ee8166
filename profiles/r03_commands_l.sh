# r03: A/B of the two-stream CNN split (STRIVE_CNN_SPLIT=1), same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03l2
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
$B > $O/bench_base_1.json 2>> $O/bench.err
STRIVE_CNN_SPLIT=1 $B > $O/bench_split_1.json 2>> $O/bench.err
$B > $O/bench_base_2.json 2>> $O/bench.err
STRIVE_CNN_SPLIT=1 $B > $O/bench_split_2.json 2>> $O/bench.err
STRIVE_CNN_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "map_cnn or rollout_smooth or reproducible" 2>&1 | tail -3 > $O/split_tests.log
