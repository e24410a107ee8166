# r05 call G: forward edge chunks on K workgroups per scene (A/B, three alternations), stepwise-sweep GPU test
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05g
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for i in 1 2 3; do
STRIVE_SCENE_FWD_K=0 $B > $O/bench_fwdk0_$i.json 2>> $O/bench.err < /dev/null
STRIVE_SCENE_FWD_K=4 $B > $O/bench_fwdk4_$i.json 2>> $O/bench.err < /dev/null
done
STRIVE_SCENE_FWD_K=2 $B > $O/bench_fwdk2_1.json 2>> $O/bench.err < /dev/null
STRIVE_SCENE_SPLIT=12 STRIVE_SCENE_FWD_K=3 $B --agents 12 > $O/bench_32x12_split12_fwdk3.json 2>> $O/bench.err < /dev/null
$B --agents 12 > $O/bench_32x12_default.json 2>> $O/bench.err < /dev/null
STRIVE_SCENE_SPLIT=12 STRIVE_SCENE_FWD_K=3 $B --agents 14 > $O/bench_32x14_split12_fwdk3.json 2>> $O/bench.err < /dev/null
$B --agents 14 > $O/bench_32x14_default.json 2>> $O/bench.err < /dev/null
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -s -k "stepwise or smooth_map_tight or per_scene or shared_forward" 2>&1 < /dev/null | tail -8) > $O/gpu_tests_g.log
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['ms_per_step'], d['value'])"; done
tail -5 $O/gpu_tests_g.log; tail -3 $O/bench.err
