# r03: PMC passes over the map-CNN training backward alone (tools/cnn_bwd_probe.py, 704 crops = 256 + 256 + 192)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03m2
rm -rf $O; mkdir -p $O
P="python tools/cnn_bwd_probe.py 704 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $P > $O/fetch.log 2>&1
python profiles/summarize_pmc.py $(find $O/fetch -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $P > $O/write.log 2>&1
python profiles/summarize_pmc.py $(find $O/write -name "*counter_collection.csv" | head -1) > $O/pmc_write_size.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq -- $P > $O/sq.log 2>&1
python profiles/summarize_pmc.py $(find $O/sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq_waits.txt 2>&1
find $O -type f -size +1M -delete
