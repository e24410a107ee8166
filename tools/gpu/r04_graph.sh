#!/bin/bash
# round 4: HIP-graph replay of the closure at the shipped operating point (one scene per batch) vs eager; full GPU suite
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04g}
mkdir -p $O
for gm in 1 0; do
  STRIVE_HIP_GRAPH=$gm timeout 200 python bench.py --steps 20 --warmup 5 --workload adv --total-agents 16 --scenes 2 --no-cpu-baseline --no-roofline > $O/bench_adv16_graph$gm.json 2> $O/bench_adv16_graph$gm.err
done
grep -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*\|"hip_graph": [a-z]*\|"rollout_kernels": "[a-z-]*"\|"final_loss": [0-9.e+-]*' $O/bench_*.json
timeout 600 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
