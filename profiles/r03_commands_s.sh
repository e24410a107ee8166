# r03: conv_ws_kernel with column-major tile order: kernel time and FETCH_SIZE
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03w2
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 300 python tools/conv_ws_probe.py 512 > $O/probe.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
python profiles/summarize_pmc.py $(find $O/fetch -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 7 --warmup 2 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB | head -12 > $O/kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
