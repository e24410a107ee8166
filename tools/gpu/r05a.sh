# r05 call A: per-scene planner status + scene quarantine (tests, 200-iteration closed loop, full pipeline) and the LDS counter pass
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05a
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
(timeout 900 python -m pytest tests/test_planner.py tests/test_gpu_configs.py tests/test_loops.py -m gpu -q -x -k "planner or closed_loop or quarantine or scenario_json" 2>&1 < /dev/null | tail -25) > $O/gpu_tests_quarantine.log
$B --workload adv --planner hardcode --steps 200 --warmup 4 > $O/bench_line_adv_hardcode_200.json 2> $O/bench_adv_hardcode.err < /dev/null
$B --workload adv --planner hardcode --steps 10 --warmup 2 --on-planner-error raise > $O/bench_line_adv_hardcode_raise_10.json 2>> $O/bench_adv_hardcode.err < /dev/null
$B --workload adv --planner hardcode --steps 10 --warmup 2 > $O/bench_line_adv_hardcode_drop_10.json 2>> $O/bench_adv_hardcode.err < /dev/null
$B --workload full --steps 1 --warmup 0 --scenario-out $O/scenarios > $O/bench_line_full_pipeline.json 2> $O/bench_full.err < /dev/null
ls $O/scenarios | wc -l > $O/scenarios_count.txt; rm -rf $O/scenarios
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $O/counters_avail.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/lds -- $B --steps 2 --warmup 1 > $O/lds.log 2>&1 < /dev/null
python profiles/summarize_pmc.py $(find $O/lds -name "*counter_collection.csv" | head -1) > $O/pmc_lds.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/lds2 -- $B --steps 2 --warmup 1 > $O/lds2.log 2>&1 < /dev/null
python profiles/summarize_pmc.py $(find $O/lds2 -name "*counter_collection.csv" | head -1) > $O/pmc_lds2.txt 2>&1
find $O -type f -size +1M -delete
tail -5 $O/gpu_tests_quarantine.log
