#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04inv}
rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_loops.py -m gpu -q -k "map_cnn or per_scene or conv2 or refine_loop_uniform or reproducible or full_size" > $O/gpu_inv_tests.log 2>&1 < /dev/null
tail -n 6 $O/gpu_inv_tests.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 120 $B --steps 30 --warmup 5 > $O/bench_32x16.json 2> $O/bench_32x16.err < /dev/null
timeout 120 $B --scenes 1 --agents 8 --steps 200 --warmup 20 > $O/bench_1x8.json 2> $O/bench_1x8.err < /dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null | awk '!s[$0]++'
