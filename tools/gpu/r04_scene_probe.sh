#!/bin/bash
# round 4: phase ticks of the scene-resident kernels (+ bench of both rollout paths when AB=1)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04b}
mkdir -p $O
timeout 120 python tools/scene_phase_probe.py 32 16 16 > $O/phase_32x16.txt 2>&1
timeout 120 python tools/scene_phase_probe.py 1 8 16 > $O/phase_1x8.txt 2>&1
grep -v amdgpu.ids $O/phase_32x16.txt $O/phase_1x8.txt
for sk in 1 ${AB:+0}; do
  STRIVE_SCENE_KERNELS=$sk timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_32x16_sk$sk.json 2> $O/bench_32x16_sk$sk.err
  STRIVE_SCENE_KERNELS=$sk timeout 200 python bench.py --steps 20 --warmup 3 --scenes 1 --agents 8 --no-cpu-baseline --no-roofline > $O/bench_1x8_sk$sk.json 2> $O/bench_1x8_sk$sk.err
done
grep -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*' $O/bench_*.json
if [ -n "$TESTS" ]; then timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or full_size" > $O/gpu_rollout_tests.log 2>&1; tail -3 $O/gpu_rollout_tests.log; fi
