#!/bin/bash
# round 4: the training step -- host synchronisations per step, bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04i}
mkdir -p $O
timeout 300 python tools/sync_audit.py train 2>&1 | grep -v amdgpu.ids | tee $O/sync_audit_train.txt | head -40
timeout 300 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err
grep -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*' $O/bench_train.json
if [ -n "$TESTS" ]; then timeout 600 python -m pytest tests/test_training.py -m gpu -q > $O/train_tests.log 2>&1; tail -5 $O/train_tests.log; fi
