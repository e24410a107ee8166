#!/bin/bash
# round 4: small-batch chain of the map CNN (conv1 one tile per workgroup, conv3 / conv4 one 32-channel block per workgroup)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04q}
mkdir -p $O
for sb in 32 0; do
  for rep in 1 2; do
    STRIVE_CNN_SMALL_BATCH=$sb timeout 200 python bench.py --steps 200 --warmup 20 --scenes 1 --agents 8 --no-cpu-baseline --no-roofline > $O/bench_1x8_sb${sb}_$rep.json 2> $O/bench_1x8_sb${sb}_$rep.err
  done
  STRIVE_CNN_SMALL_BATCH=$sb timeout 200 python bench.py --steps 100 --warmup 10 --scenes 2 --agents 12 --no-cpu-baseline --no-roofline > $O/bench_2x12_sb$sb.json 2> $O/bench_2x12_sb$sb.err
  STRIVE_CNN_SMALL_BATCH=$sb timeout 200 python bench.py --workload adv --planner hardcode --total-agents 8 --scenes 1 --steps 20 --warmup 6 --no-cpu-baseline --no-roofline > $O/bench_cl1x8_sb$sb.json 2> $O/bench_cl1x8_sb$sb.err
done
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json
for f in $O/*.err; do if [ -s $f ]; then echo "== $f"; tail -n 3 $f; fi; done
cd /tmp && export TMPDIR=/tmp
for sb in 32 0; do
  STRIVE_HIP_GRAPH=0 STRIVE_CNN_SMALL_BATCH=$sb timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_sb$sb -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --scenes 1 --agents 8 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof_sb$sb.log 2>&1
done
cd $GRAFT_REPO_ROOT
for sb in 32 0; do
  f=$(find $O/prof_sb$sb -name '*kernel_stats.csv' 2>/dev/null | head -n 1)
  echo "== sb $sb $f"
  if [ -n "$f" ]; then cp "$f" $O/prof_1x8_sb${sb}_stats.csv; head -n 14 "$f" < /dev/null | cut -c1-150; fi
  rm -rf $O/prof_sb$sb          # keep only the summary: the traces are larger than what gpurun copies back
done
timeout 600 python -m pytest tests -m gpu -x -q -k "cnn or crop or rollout or full_size or training" > $O/gpu_cnn_tests.log 2>&1; tail -3 $O/gpu_cnn_tests.log
