#!/bin/bash
# round 4: HIP-graph replay of the adversarial iteration with the SHARED forward rollout (one forward, two sweeps on one stream)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04y}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for gm in 0 1; do
  STRIVE_HIP_GRAPH=$gm timeout 120 $B --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 20 --warmup 6 > $O/bench_cl1x8_graph$gm.json 2> $O/bench_cl1x8_graph$gm.err < /dev/null
  STRIVE_HIP_GRAPH=$gm timeout 120 $B --workload adv --total-agents 8 --scenes 1 --steps 40 --warmup 6 > $O/bench_adv1x8_graph$gm.json 2> $O/bench_adv1x8_graph$gm.err < /dev/null
  STRIVE_HIP_GRAPH=$gm timeout 120 $B --workload adv --total-agents 16 --scenes 2 --steps 40 --warmup 6 > $O/bench_adv16_graph$gm.json 2> $O/bench_adv16_graph$gm.err < /dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04y/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['config'].get('hip_graph'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for f in $O/*.err; do if [ -s $f ]; then echo "== $f"; tail -n 4 $f; fi; done
