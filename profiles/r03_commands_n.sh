# r03: A/B of the weight-fragment ring of conv_bf6_kernel (3 buffers + barrier per step vs 5 buffers + barrier per two steps)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03o2
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline"
for rep in 1 2; do
  cp strive_amd/libstrive_hip_ring3.so strive_amd/libstrive_hip.so
  $B > $O/bench_ring3_$rep.json 2>> $O/bench.err
  cp strive_amd/libstrive_hip_ring5.so strive_amd/libstrive_hip.so
  $B > $O/bench_ring5_$rep.json 2>> $O/bench.err
done
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -4) > $O/ring5_tests.log
timeout 300 python tools/cnn_stress.py > $O/cnn_stress.log 2>&1
