# r03: A/B of conv2 with specialised producer / consumer waves (STRIVE_CONV_WS=1, default) vs conv_bf6_kernel (=0), same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03u2
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for rep in 1 2; do
  STRIVE_CONV_WS=0 $B > $O/bench_ws0_$rep.json 2>> $O/bench.err
  STRIVE_CONV_WS=1 $B > $O/bench_ws1_$rep.json 2>> $O/bench.err
done
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $O/ws_tests.log
timeout 300 python tools/cnn_stress.py > $O/cnn_stress.log 2>&1
