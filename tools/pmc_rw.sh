cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/pmc2; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for c in WRITE_SIZE FETCH_SIZE; do
timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- $B --steps 2 --warmup 1 > $O/$c.log 2>&1
python profiles/summarize_pmc.py $(find $O/$c -name "*counter_collection.csv" | head -1) | grep -E "conv|tail"
done
find $O -type f -size +1M -delete
