# r03: A/B of the trainer's gradient sink (HIP backward calls accumulate straight into the flat bucket), same box, alternating
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03x4
rm -rf $O; mkdir -p $O
nproc > $O/host.txt; uptime >> $O/host.txt
B="python bench.py --no-cpu-baseline --no-roofline --workload train --steps 20 --warmup 3"
for rep in 1 2 3; do
  STRIVE_NO_GRADSINK=1 $B > $O/bench_nosink_$rep.json 2>> $O/bench.err
  $B > $O/bench_sink_$rep.json 2>> $O/bench.err
done
uptime >> $O/host.txt
