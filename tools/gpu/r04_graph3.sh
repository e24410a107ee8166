#!/bin/bash
# round 4: HIP-graph replay of the CLOSED-LOOP adversarial iteration (device planner inside the capture)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04z}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for gm in 0 1; do
  STRIVE_HIP_GRAPH=$gm timeout 120 $B --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 20 --warmup 6 > $O/bench_cl1x8_graph$gm.json 2> $O/bench_cl1x8_graph$gm.err < /dev/null
  STRIVE_HIP_GRAPH=$gm timeout 120 $B --workload adv --planner hardcode --total-agents 20 --scenes 1 --steps 20 --warmup 6 > $O/bench_cl1x20_graph$gm.json 2> $O/bench_cl1x20_graph$gm.err < /dev/null
  STRIVE_HIP_GRAPH=$gm timeout 120 $B --workload adv --planner hardcode --total-agents 48 --scenes 4 --steps 20 --warmup 6 > $O/bench_cl4x12_graph$gm.json 2> $O/bench_cl4x12_graph$gm.err < /dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get('O', 'gpurun_out/r04z') + '/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['config'].get('hip_graph'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for f in $O/*.err; do if [ -s $f ]; then echo "== $f"; tail -n 4 $f; fi; done
