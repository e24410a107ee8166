# r03: A/B of conv2 with 4-row tiles at 4 workgroups per CU (pt1) vs 8-row tiles at 3 (pt2), same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03s2
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline"
for rep in 1 2; do
  cp strive_amd/libstrive_hip_pt2.so strive_amd/libstrive_hip.so
  $B > $O/bench_pt2_$rep.json 2>> $O/bench.err
  cp strive_amd/libstrive_hip_pt1.so strive_amd/libstrive_hip.so
  $B > $O/bench_pt1_$rep.json 2>> $O/bench.err
done
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -4) > $O/pt1_tests.log
