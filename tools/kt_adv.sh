cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/kt_adv; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --workload adv --steps 7 --warmup 2"
$B 2>/dev/null | tail -1 | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- $B > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt 2>&1
python profiles/gap_report.py $DB 5 rollout_init_kernel 2 > $O/gaps.txt 2>&1
find $O -type f -size +1M -delete
head -45 $O/kernel_stats.txt | cut -c1-150; head -12 $O/gaps.txt
