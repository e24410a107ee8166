# r05 call E: training step after the tape-based sweep kernels; fused-tail S A/B (3 alternations); attacker / stepwise tests
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05e
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
$B --workload train --steps 10 --warmup 3 > $O/bench_line_train.json 2> $O/bench.err < /dev/null
for i in 1 2 3; do
STRIVE_CNN_TAIL_S=4 $B --steps 20 --warmup 5 > $O/bench_tail_s4_$i.json 2>> $O/bench.err < /dev/null
STRIVE_CNN_TAIL_S=2 $B --steps 20 --warmup 5 > $O/bench_tail_s2_$i.json 2>> $O/bench.err < /dev/null
done
(timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_training.py tests/test_loops.py -m gpu -q -s -k "attacker or training_step or lagged or sol or refine_loop_uniform" 2>&1 < /dev/null | tail -30) > $O/gpu_tests_e.log
timeout 400 rocprofv3 --kernel-trace --stats -d $O/ktt -- $B --workload train --steps 3 --warmup 1 > $O/ktt.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/ktt -name "*.db" | head -1) | head -50 > $O/train_kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['ms_per_step'], d['value'])"; done
tail -12 $O/gpu_tests_e.log
