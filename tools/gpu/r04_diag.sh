#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04f}
mkdir -p $O
timeout 300 python tools/grad_accuracy.py 16 2>&1 | grep -v amdgpu.ids | tee $O/grad_accuracy.txt
STRIVE_SCENE_KERNELS=0 timeout 300 python -m pytest tests/test_loops.py -m gpu -x -q -k "uniform_raster_tight" 2>&1 | tail -5 | tee $O/loops_sk0.log
STRIVE_SCENE_KERNELS=1 timeout 300 python -m pytest tests/test_loops.py -m gpu -q -k "uniform_raster_tight" -s 2>&1 | grep "loop u\|passed\|failed\|Error" | tee $O/loops_sk1.log
