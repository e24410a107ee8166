#!/bin/bash
# round 4: GPU parity tests + A/B of the scene-resident rollout kernels (STRIVE_SCENE_KERNELS=1/0), 32 x 16 and 1 x 8 agents
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04a
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
for sk in 1 0; do
  STRIVE_SCENE_KERNELS=$sk timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_32x16_sk$sk.json 2> $O/bench_32x16_sk$sk.err
  STRIVE_SCENE_KERNELS=$sk timeout 200 python bench.py --steps 20 --warmup 3 --scenes 1 --agents 8 --no-cpu-baseline --no-roofline > $O/bench_1x8_sk$sk.json 2> $O/bench_1x8_sk$sk.err
done
grep -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*' $O/bench_*.json
