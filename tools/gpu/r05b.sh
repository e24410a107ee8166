# r05 call B: do co-resident conv workgroups run in lock step (dephase probe); headline with / without graph replay; lagged abs-max
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05b
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 600 python tools/conv_dephase_probe.py 512 > $O/conv_dephase_probe.txt 2>&1
$B --steps 20 --warmup 5 > $O/bench_line_graph_auto.json 2> $O/bench.err < /dev/null
STRIVE_HIP_GRAPH=0 $B --steps 20 --warmup 5 > $O/bench_line_graph_off.json 2>> $O/bench.err < /dev/null
(timeout 600 python -m pytest tests/test_training.py tests/test_loops.py -m gpu -q -x -k "lagged or graph_replay or training_step" 2>&1 < /dev/null | tail -15) > $O/gpu_tests_b.log
cat $O/conv_dephase_probe.txt | head -80
