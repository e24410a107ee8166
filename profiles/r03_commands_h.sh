# r03: map-CNN backward alone; STRIVE_WGRAD_DBG phase attribution of the matrix-core weight-gradient kernel
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03p
rm -rf $O; mkdir -p $O
for D in 0; do
  STRIVE_WGRAD_DBG=$D timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt$D -- python tools/cnn_bwd_probe.py 704 3 > $O/probe$D.log 2>&1
  DB=$(find $O/kt$D -name "*.db" | head -1)
  python profiles/summarize_rocpd.py $DB | head -40 > $O/stats$D.txt 2>&1
done
find $O -type f -size +1M -delete
