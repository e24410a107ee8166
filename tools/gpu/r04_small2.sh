#!/bin/bash
# round 4: fused tail with 1 / 2 / 4 samples per workgroup at the one-scene point, the new GPU tests, the training line's roofline
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04s}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 240 python -m pytest tests/test_gpu_parity.py tests/test_training.py -m gpu -x -q -k "small_batch_chain or measurement_hook or map_cnn" > $O/gpu_new_tests.log 2>&1 < /dev/null; tail -n 3 $O/gpu_new_tests.log
for ts in 1 2 4; do
  STRIVE_CNN_TAIL_S=$ts timeout 120 $B --scenes 1 --agents 8 --steps 200 --warmup 20 > $O/bench_1x8_tail$ts.json 2> $O/bench_1x8_tail$ts.err < /dev/null
done
timeout 120 $B --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 20 --warmup 6 > $O/bench_cl1x8.json 2> $O/bench_cl1x8.err < /dev/null
timeout 200 python bench.py --no-cpu-baseline --workload train --steps 10 --warmup 3 > $O/bench_line_train.json 2> $O/bench_train.err < /dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null
python - <<PY < /dev/null
import json
try:
    d = json.loads(open('$O/bench_line_train.json').read().strip().splitlines()[-1])
    print('train roofline:', json.dumps(d.get('roofline'))[:900])
except Exception as e:
    print('train line unreadable', e)
PY
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt8 -- env STRIVE_HIP_GRAPH=0 $B --scenes 1 --agents 8 --steps 20 --warmup 3 > $O/kt8.log 2>&1 < /dev/null
DB=$(find $O/kt8 -name "*.db" 2>/dev/null | head -n 1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 2> $O/kt8_sum.err < /dev/null | head -n 40 > $O/1x8_kernel_stats.txt; head -n 12 $O/1x8_kernel_stats.txt | cut -c1-150; fi
find $O -type f -size +1M -delete
for f in $O/*.err; do if [ -s $f ]; then echo "== $f"; tail -n 2 $f; fi; done
