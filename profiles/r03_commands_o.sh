# r03: A/B of conv1 with both weight pieces in one 32x32x16 instruction (new) vs the 16x16x32 form (old), same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03r2
rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -4) > $O/conv1new_tests.log
B="python bench.py --no-cpu-baseline"
for rep in 1 2; do
  cp strive_amd/libstrive_hip_conv1old.so strive_amd/libstrive_hip.so
  $B > $O/bench_old_$rep.json 2>> $O/bench.err
  cp strive_amd/libstrive_hip_conv1new.so strive_amd/libstrive_hip.so
  $B > $O/bench_new_$rep.json 2>> $O/bench.err
done
python tools/conv1_probe.py 512 > $O/conv1_probe_new.txt 2>&1
timeout 300 python tools/cnn_stress.py > $O/cnn_stress.log 2>&1
