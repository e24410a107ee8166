# r05 call F: two-rollout iterations replayed as a single-stream HIP graph vs eager two-stream, at 512 agents and at 1 x 16 / 4 x 16
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05f
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for g in 0 1; do
STRIVE_HIP_GRAPH=$g $B --workload adv --steps 20 --warmup 6 > $O/adv512_graph$g.json 2>> $O/bench.err < /dev/null
STRIVE_HIP_GRAPH=$g $B --workload adv --planner hardcode --steps 20 --warmup 6 > $O/advcl512_graph$g.json 2>> $O/bench.err < /dev/null
STRIVE_HIP_GRAPH=$g $B --workload adv --total-agents 128 --steps 20 --warmup 6 > $O/adv128_graph$g.json 2>> $O/bench.err < /dev/null
STRIVE_HIP_GRAPH=$g $B --workload adv --total-agents 16 --scenes 1 --steps 30 --warmup 6 > $O/adv16_graph$g.json 2>> $O/bench.err < /dev/null
done
(timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "attacker" 2>&1 < /dev/null | tail -12) > $O/gpu_tests_f.log
for f in $O/*graph*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step'], d.get('host_cpu_ms_per_step'), d['config']['hip_graph'], d.get('scenes_dropped',{}) and d['scenes_dropped'].get('count'))"; done
tail -5 $O/gpu_tests_f.log; tail -3 $O/bench.err
