# r03: closed-loop adversarial closure: conv2 as conv_bf6_kernel (ws0) vs conv_ws_kernel leaving 0 / 16 / 32 CUs to the planner, same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03x6
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --workload adv --planner hardcode"
for rep in 1 2; do
  STRIVE_CONV_WS=0 $B > $O/advhc_ws0_$rep.json 2>> $O/bench.err
  STRIVE_PLANNER_CUS=0 $B > $O/advhc_ws1_r0_$rep.json 2>> $O/bench.err
  STRIVE_PLANNER_CUS=16 $B > $O/advhc_ws1_r16_$rep.json 2>> $O/bench.err
  STRIVE_PLANNER_CUS=32 $B > $O/advhc_ws1_r32_$rep.json 2>> $O/bench.err
done
