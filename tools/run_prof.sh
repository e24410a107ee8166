set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/gpu_tests.log
timeout 300 python bench.py > gpurun_out/bench_final.log 2>&1
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -- python bench.py --steps 5 --warmup 2 > gpurun_out/prof_kt.log 2>&1
DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > gpurun_out/kernel_stats.txt 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 2 --warmup 1 > gpurun_out/prof_fetch.log 2>&1
F=$(find gpurun_out/prof_fetch -name "*counter_collection.csv" | head -1)
python profiles/summarize_pmc.py $F > gpurun_out/pmc_fetch.txt 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 2 --warmup 1 > gpurun_out/prof_write.log 2>&1
F=$(find gpurun_out/prof_write -name "*counter_collection.csv" | head -1)
python profiles/summarize_pmc.py $F > gpurun_out/pmc_write.txt 2>&1
find gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write -type f -size +2M -delete
