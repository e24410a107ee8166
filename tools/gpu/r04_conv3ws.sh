#!/bin/bash
# round 4 (last GPU seconds): conv3 on specialised waves with the weight ring -- time per launch, bit identity, the closure with it
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r04c3}
rm -rf $O; mkdir -p $O
timeout 40 python tools/conv_ws_probe.py 512 > $O/conv_ws_probe.txt 2>&1 < /dev/null; grep "N=" $O/conv_ws_probe.txt; tail -n 2 $O/conv_ws_probe.txt | grep -v "N=" | cut -c1-300
B="python bench.py --no-cpu-baseline --no-roofline"
STRIVE_CONV3_WS=2 timeout 40 $B --steps 20 --warmup 4 > $O/bench_32x16_conv3ws2.json 2> $O/b2.err < /dev/null
timeout 40 $B --steps 20 --warmup 4 > $O/bench_32x16_conv3ws0.json 2> $O/b0.err < /dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_*.json < /dev/null | awk '!s[$0]++'
