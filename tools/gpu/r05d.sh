# r05 call D: stepwise reverse sweep (K workgroups per scene) A/B at the headline and at 32 x 12; fused tail with 2 samples per workgroup
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05d
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for k in 0 4 2 3; do
STRIVE_SWEEP_STEP=$k $B > $O/bench_32x16_sweep_step$k.json 2>> $O/bench.err < /dev/null
done
$B > $O/bench_32x16_default.json 2>> $O/bench.err < /dev/null
for k in 0 3; do
STRIVE_SWEEP_STEP=$k $B --agents 12 > $O/bench_32x12_sweep_step$k.json 2>> $O/bench.err < /dev/null
STRIVE_SWEEP_STEP=$k $B --agents 14 > $O/bench_32x14_sweep_step$k.json 2>> $O/bench.err < /dev/null
done
STRIVE_SWEEP_STEP=2 $B --agents 10 > $O/bench_32x10_sweep_step2.json 2>> $O/bench.err < /dev/null
STRIVE_SWEEP_STEP=0 $B --agents 10 > $O/bench_32x10_sweep_step0.json 2>> $O/bench.err < /dev/null
STRIVE_CNN_TAIL_S=2 $B > $O/bench_32x16_tail_s2.json 2>> $O/bench.err < /dev/null
STRIVE_CNN_TAIL_S=1 $B > $O/bench_32x16_tail_s1.json 2>> $O/bench.err < /dev/null
STRIVE_HIP_GRAPH=0 $B > $O/bench_32x16_default_eager.json 2>> $O/bench.err < /dev/null
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_loops.py -m gpu -q -x -k "smooth_map_tight or per_scene or attacker or refine_closure_at or graph_replay" 2>&1 < /dev/null | tail -8) > $O/gpu_tests_d.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --no-cpu-baseline --no-roofline --steps 7 --warmup 2 > $O/kt.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/kt -name "*.db" | head -1) | head -40 > $O/kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['ms_per_step'], d['value'])"; done
tail -4 $O/gpu_tests_d.log
