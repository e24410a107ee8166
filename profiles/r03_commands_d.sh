# r03: kernel traces of the non-headline workloads at HEAD (train, adv 'ego', sharded4096 on one GPU, 1 x 8 agents)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03d
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for W in "train:--workload train --steps 3 --warmup 1" "adv:--workload adv --steps 6 --warmup 2" "sharded4096:--workload sharded4096 --steps 3 --warmup 1" "1x8:--scenes 1 --agents 8 --steps 20 --warmup 3"; do
  N=${W%%:*}; A=${W#*:}
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_$N -- $B $A > $O/kt_$N.log 2>&1
  DB=$(find $O/kt_$N -name "*.db" | head -1)
  python profiles/summarize_rocpd.py $DB | head -45 > $O/${N}_kernel_stats.txt 2>&1
  tail -1 $O/kt_$N.log | cut -c1-300 > $O/${N}_bench_line.txt
done
find $O -type f -size +1M -delete
