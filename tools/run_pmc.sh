cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | grep -i "MFMA\|LDS\|WAIT\|ACTIVE\|BUSY\|WAVE_CYCLES\|VALU" > gpurun_out/sq_counters.txt
rm -rf gpurun_out/prof_sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/prof_sq -- python bench.py --steps 1 --warmup 1 > gpurun_out/prof_sq.log 2>&1
F=$(find gpurun_out/prof_sq -name "*counter_collection.csv" | head -1)
python profiles/summarize_pmc.py $F > gpurun_out/pmc_sq.txt 2>&1
rm -rf gpurun_out/prof_sq2
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/prof_sq2 -- python bench.py --steps 1 --warmup 1 > gpurun_out/prof_sq2.log 2>&1
F=$(find gpurun_out/prof_sq2 -name "*counter_collection.csv" | head -1)
python profiles/summarize_pmc.py $F > gpurun_out/pmc_sq2.txt 2>&1
find gpurun_out/prof_sq gpurun_out/prof_sq2 -type f -size +2M -delete
