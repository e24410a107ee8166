#!/bin/bash
# round 4: up to which batch size does the replayed graph of the refine iteration pay (AUTO_MAX_AGENTS)?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04g4}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for cfg in "6 16" "8 16" "12 16" "16 16" "32 16"; do
  set -- $cfg
  for gm in 0 1; do
    STRIVE_HIP_GRAPH=$gm timeout 120 $B --scenes $1 --agents $2 --steps 40 --warmup 6 > $O/bench_$1x$2_graph$gm.json 2> $O/bench_$1x$2_graph$gm.err < /dev/null
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get('O', 'gpurun_out/r04g4') + '/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['config'].get('hip_graph'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
