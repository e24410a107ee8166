#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04pp}
rm -rf $O; mkdir -p $O
timeout 200 python tools/planner_phase_probe.py 8 > $O/planner_phase_1x8.txt 2> $O/planner_phase_1x8.err < /dev/null
cat $O/planner_phase_1x8.txt; tail -n 5 $O/planner_phase_1x8.err
