cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/kt2; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --no-cpu-baseline --no-roofline --steps 7 --warmup 2 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
head -50 $O/kernel_stats.txt
