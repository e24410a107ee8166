# Round 5, second half (the training step): the lines and traces behind DESIGN.md 4.11's training table, taken at the round's head.
# Run with:  gpurun --timeout 1800 -- 'bash profiles/r05_commands_training.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05t_final; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
python bench.py > $O/bench_line_driver_default.json 2> $O/bench.err                                     # the driver's command
python bench.py --no-cpu-baseline --workload train --steps 20 --warmup 4 > $O/bench_line_train.json 2>> $O/bench.err      # (with its roofline record)
timeout 400 rocprofv3 --kernel-trace --stats -d $O/ktt -- $B --workload train --steps 6 --warmup 3 > $O/ktt.log 2>&1 < /dev/null
DB=$(find $O/ktt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB | head -70 > $O/train_kernel_stats.txt 2>&1
python profiles/gap_report.py $DB 5 rollout_init_kernel 3 | head -30 > $O/train_gaps.txt 2>&1
find $O -type f -size +1M -delete
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -8) > $O/gpu_tests.log
tail -c 300 $O/bench_line_driver_default.json; tail -c 200 $O/bench_line_train.json | head -c 200; echo; head -3 $O/train_gaps.txt; tail -2 $O/smoke.log; tail -3 $O/gpu_tests.log; tail -3 $O/bench.err
