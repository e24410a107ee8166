#!/bin/bash
# round 4: forward split of large scenes (STRIVE_SCENE_SPLIT = agents per scene from which the edge rows leave the scene kernel)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04p}
mkdir -p $O
for sp in 12 0; do
  STRIVE_SCENE_SPLIT=$sp timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_32x16_split$sp.json 2> $O/bench_32x16_split$sp.err
  STRIVE_SCENE_SPLIT=$sp timeout 200 python bench.py --steps 20 --warmup 3 --scenes 32 --agents 12 --no-cpu-baseline --no-roofline > $O/bench_32x12_split$sp.json 2> $O/bench_32x12_split$sp.err
done
STRIVE_SCENE_SPLIT=8 timeout 200 python bench.py --steps 20 --warmup 3 --scenes 32 --agents 8 --no-cpu-baseline --no-roofline > $O/bench_32x8_split8.json 2> $O/bench_32x8_split8.err
STRIVE_SCENE_SPLIT=0 timeout 200 python bench.py --steps 20 --warmup 3 --scenes 32 --agents 8 --no-cpu-baseline --no-roofline > $O/bench_32x8_split0.json 2> $O/bench_32x8_split0.err
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json | awk '!s[$0]++'
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or full_size" > $O/gpu_rollout_tests.log 2>&1; tail -3 $O/gpu_rollout_tests.log
