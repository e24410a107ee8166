# r03 (HEAD): closed-loop closure with the concurrency switch; planner / loop GPU tests
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03x7
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --workload adv"
STRIVE_CONV_WS=0 $B --planner hardcode > $O/advhc_ws0.json 2>> $O/bench.err
$B --planner hardcode > $O/advhc_head.json 2>> $O/bench.err
$B > $O/adv_head.json 2>> $O/bench.err
(timeout 900 python -m pytest tests/test_planner.py tests/test_loops.py -m gpu -q 2>&1 | tail -4) > $O/tests.log
