# r03: A/B of the staging split with v_cvt_pk_f16_f32 (new) vs scalar conversions (old), same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03x8
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline"
for rep in 1 2; do
  cp strive_amd/libstrive_hip_old.so strive_amd/libstrive_hip.so
  $B > $O/bench_old_$rep.json 2>> $O/bench.err
  cp strive_amd/libstrive_hip_new.so strive_amd/libstrive_hip.so
  $B > $O/bench_new_$rep.json 2>> $O/bench.err
done
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3) > $O/tests_new.log
