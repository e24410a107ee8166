# r06: rocprofv3 evidence of the driver's bench command -- kernel trace + separate PMC passes (FETCH_SIZE, WRITE_SIZE -> r06_traffic.json;
# LDS / matrix-pipe counters -> r06_util.json) -> the bench lines of every workload; kernel traces of the one-scene operating point, the
# adversarial closure, the closed loop and the training step; the GPU suite.  Run with:  gpurun --timeout 2400 -- 'bash profiles/r06_commands.sh'
# (The convolution A/B runs of the round: tools/conv_variants_probe.py, tools/gpu/r06_decomp.sh -> r06_conv_variants_first_probe.txt,
#  r06_conv_ws_decomposition.txt.)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r06final}
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
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/lds -- $B --steps 2 --warmup 1 > $O/lds.log 2>&1 < /dev/null
python profiles/summarize_pmc.py $(find $O/lds -name "*counter_collection.csv" | head -1) > $O/pmc_lds.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/lds2 -- $B --steps 2 --warmup 1 > $O/lds2.log 2>&1 < /dev/null
python profiles/summarize_pmc.py $(find $O/lds2 -name "*counter_collection.csv" | head -1) > $O/pmc_lds2.txt 2>&1
python profiles/make_util.py $O/pmc_lds.txt $O/pmc_lds2.txt > $O/util.json 2>> $O/lds.log
find $O -type f -size +1M -delete
if [ -s $O/traffic.json ]; then cp $O/traffic.json profiles/r06_traffic.json; fi          # (this run's own counters feed bench.py's `traffic` / `on_chip_roofs` fields below)
if [ -s $O/util.json ]; then cp $O/util.json profiles/r06_util.json; fi
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
$B --workload adv --steps 20 --warmup 4 > $O/bench_line_adv.json 2>> $O/bench.err < /dev/null
$B --workload adv --planner hardcode --steps 200 --warmup 4 > $O/bench_line_adv_hardcode.json 2>> $O/bench.err < /dev/null
$B --scenes 1 --agents 8 --steps 40 --warmup 5 > $O/bench_line_1x8.json 2>> $O/bench.err < /dev/null
$B --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 20 --warmup 6 > $O/bench_line_closed_loop_1x8.json 2>> $O/bench.err < /dev/null
python bench.py --no-cpu-baseline --workload train --steps 10 --warmup 3 > $O/bench_line_train.json 2>> $O/bench.err      # (with its roofline record)
$B --workload sample --steps 5 > $O/bench_line_sample.json 2>> $O/bench.err < /dev/null
$B --workload sharded4096 --steps 5 > $O/bench_line_sharded4096_1gpu.json 2>> $O/bench.err < /dev/null
$B --workload full --steps 1 --warmup 0 > $O/bench_line_full_pipeline.json 2>> $O/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt8 -- env STRIVE_HIP_GRAPH=0 $B --scenes 1 --agents 8 --steps 20 --warmup 3 > $O/kt8.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/kt8 -name "*.db" | head -1) | head -60 > $O/1x8_kernel_stats.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kta -- $B --workload adv --steps 5 --warmup 2 > $O/kta.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/kta -name "*.db" | head -1) | head -60 > $O/adv_kernel_stats.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktc -- $B --workload adv --planner hardcode --steps 5 --warmup 2 > $O/ktc.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/ktc -name "*.db" | head -1) | head -60 > $O/adv_hardcode_kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/ktt -- $B --workload train --steps 3 --warmup 1 > $O/ktt.log 2>&1 < /dev/null
python profiles/summarize_rocpd.py $(find $O/ktt -name "*.db" | head -1) | head -60 > $O/train_kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -8) > $O/gpu_tests.log
tail -c 400 $O/bench_line.json; tail -3 $O/gpu_tests.log; tail -5 $O/bench.err
