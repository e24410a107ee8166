# r03: kernel trace of the default bench command with conv2 on specialised waves (HEAD)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03v2
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 7 --warmup 2 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt 2>&1
STRIVE_CONV_WS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt0 -- $B --steps 7 --warmup 2 > $O/kt0.log 2>&1
DB=$(find $O/kt0 -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB > $O/kernel_stats_ws0.txt 2>&1
find $O -type f -size +1M -delete
