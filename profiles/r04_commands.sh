# r04: rocprofv3 evidence of the driver's bench command -- kernel trace + separate PMC passes -> r04_traffic.json -> the bench
# lines of every workload; kernel traces of the one-scene operating point, the adversarial closure and the training step; GPU suite
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04final}
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 7 --warmup 2 > $O/kt.log 2>&1 < /dev/null
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt 2>&1
python profiles/gap_report.py $DB 5 rollout_init_kernel 2 > $O/gaps.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 2 --warmup 1 > $O/fetch.log 2>&1 < /dev/null
python profiles/summarize_pmc.py $(find $O/fetch -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 2 --warmup 1 > $O/write.log 2>&1 < /dev/null
python profiles/summarize_pmc.py $(find $O/write -name "*counter_collection.csv" | head -1) > $O/pmc_write_size.txt 2>&1
python profiles/make_traffic.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt $O/kernel_stats.txt > $O/traffic.json 2>> $O/fetch.log
find $O -type f -size +1M -delete
if [ -s $O/traffic.json ]; then cp $O/traffic.json profiles/r04_traffic.json; fi          # (this run's own counters feed bench.py's `traffic` fields below)
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
$B --workload adv --steps 20 --warmup 4 > $O/bench_line_adv.json 2>> $O/bench.err < /dev/null
$B --workload adv --planner hardcode > $O/bench_line_adv_hardcode.json 2>> $O/bench.err < /dev/null
$B --scenes 1 --agents 8 --steps 40 --warmup 5 > $O/bench_line_1x8.json 2>> $O/bench.err < /dev/null
STRIVE_HIP_GRAPH=0 STRIVE_SCENE_KERNELS=0 STRIVE_CNN_SMALL_BATCH=0 STRIVE_CNN_TAIL_S=4 $B --scenes 1 --agents 8 --steps 40 --warmup 5 > $O/bench_line_1x8_r03_path.json 2>> $O/bench.err
$B --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 20 --warmup 6 > $O/bench_line_closed_loop_1x8.json 2>> $O/bench.err < /dev/null
python bench.py --no-cpu-baseline --workload train --steps 10 --warmup 3 > $O/bench_line_train.json 2>> $O/bench.err      # (with its roofline record)
$B --workload sample --steps 5 > $O/bench_line_sample.json 2>> $O/bench.err < /dev/null
$B --workload sharded4096 --steps 5 > $O/bench_line_sharded4096_1gpu.json 2>> $O/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt8 -- env STRIVE_HIP_GRAPH=0 $B --scenes 1 --agents 8 --steps 20 --warmup 3 > $O/kt8.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/kt8 -name "*.db" | head -1) | head -60 > $O/1x8_kernel_stats.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kta -- $B --workload adv --steps 5 --warmup 2 > $O/kta.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/kta -name "*.db" | head -1) | head -60 > $O/adv_kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/ktt -- $B --workload train --steps 3 --warmup 1 > $O/ktt.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/ktt -name "*.db" | head -1) | head -60 > $O/train_kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
(timeout 900 python -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -6) > $O/gpu_tests.log
tail -c 300 $O/bench_line.json
