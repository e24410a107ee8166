#!/bin/bash
# round 4: closed loop -- planner on a side stream under the adversarial loss and its sweep (STRIVE_PLANNER_OVERLAP=0 | 1)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04co}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 400 python -m pytest tests -m gpu -x -q -k "closed_loop or g6h or planner or scenario_json or loop or shared" > $O/gpu_cl_tests.log 2>&1 < /dev/null; tail -n 4 $O/gpu_cl_tests.log
for ov in ; do
  STRIVE_PLANNER_OVERLAP=$ov timeout 120 $B --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 20 --warmup 6 > $O/bench_cl1x8_ov$ov.json 2> $O/bench_cl1x8_ov$ov.err < /dev/null
  STRIVE_PLANNER_OVERLAP=$ov timeout 120 $B --workload adv --planner hardcode --total-agents 20 --scenes 1 --steps 20 --warmup 6 > $O/bench_cl1x20_ov$ov.json 2> $O/bench_cl1x20_ov$ov.err < /dev/null
  STRIVE_PLANNER_OVERLAP=$ov timeout 200 $B --workload adv --planner hardcode --steps 10 --warmup 2 > $O/bench_cl512_ov$ov.json 2> $O/bench_cl512_ov$ov.err < /dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get('O', 'gpurun_out/r04co') + '/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for f in $O/*.err; do if [ -s $f ]; then echo "== $f"; tail -n 4 $f; fi; done
