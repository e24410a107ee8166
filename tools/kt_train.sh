cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/kt_train; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --no-cpu-baseline --no-roofline --workload train --steps 4 --warmup 1 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt 2>&1
find $O -type f -size +1M -delete
head -25 $O/kernel_stats.txt | cut -c1-150
