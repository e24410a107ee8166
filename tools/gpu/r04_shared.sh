#!/bin/bash
# round 4: one forward rollout for the complementary-detach pair (STRIVE_SHARED_ROLLOUT=1/0): tests + adversarial bench lines
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04n}
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "shared_forward or loops_uniform" > $O/gpu_sel.log 2>&1; tail -6 $O/gpu_sel.log
for sh in 1 0; do
  STRIVE_SHARED_ROLLOUT=$sh timeout 300 python bench.py --steps 20 --warmup 4 --workload adv --no-cpu-baseline --no-roofline > $O/bench_adv_shared$sh.json 2> $O/bench_adv_shared$sh.err
  STRIVE_SHARED_ROLLOUT=$sh timeout 300 python bench.py --workload adv --planner hardcode --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > $O/bench_advcl_shared$sh.json 2> $O/bench_advcl_shared$sh.err
  STRIVE_SHARED_ROLLOUT=$sh timeout 300 python bench.py --steps 20 --warmup 6 --workload adv --total-agents 16 --scenes 2 --no-cpu-baseline --no-roofline > $O/bench_adv16_shared$sh.json 2> $O/bench_adv16_shared$sh.err
  STRIVE_SHARED_ROLLOUT=$sh timeout 300 python bench.py --steps 20 --warmup 6 --workload adv --planner hardcode --total-agents 8 --scenes 1 --no-cpu-baseline --no-roofline > $O/bench_advcl8_shared$sh.json 2> $O/bench_advcl8_shared$sh.err
done
for f in $O/*.err; do grep -v amdgpu.ids $f | tail -n 2; done
grep -o '"ms_per_step": [0-9.]*\|"hip_graph": [a-z]*\|"value": [0-9.]*' $O/bench_*.json
